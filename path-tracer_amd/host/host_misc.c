/*
 * host_misc.c - camera basis, seed blocks, row flips, PPM snapshot, captions and the adaptive
 * frame budget: the small host-side pieces around the device path.  Each function cites the
 * reference code it stands in for.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pthost.h"
#include "pthost_internal.h"

#define PI_F 3.14159265358979323846f
#define CAMERA_FOV_DEGREES 45.f /* HARD_CODED_CAMERA_FOV, reference camera.h:5 */

/* v *= 1/|v| with the reference's double-promoted reciprocal square root (vec.h:20) */
static void normalize_promoted(pt_vec3 *v) {
    float l = (float)(1.0 / sqrt((double)(v->x * v->x + v->y * v->y + v->z * v->z)));
    v->x = l * v->x;
    v->y = l * v->y;
    v->z = l * v->z;
}

static pt_vec3 cross3(pt_vec3 a, pt_vec3 b) {
    pt_vec3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}

/* reference camera.c:8-32.  The view direction comes from yaw/pitch only; `target` is unused.
 * fov (radians) is used as a linear film scale. */
void pthost_update_camera_basis(pt_camera *cam) {
    float dx = (float)(sin((double)cam->yaw) * cos((double)cam->pitch));
    float dy = (float)sin((double)cam->pitch);
    float dz = (float)(cos((double)cam->yaw) * cos((double)cam->pitch));
    const float minus_one = -1;
    cam->dir.x = minus_one * dx;
    cam->dir.y = minus_one * dy;
    cam->dir.z = minus_one * dz;
    normalize_promoted(&cam->dir);

    const pt_vec3 up = {0.f, 1.f, 0.f};
    const float fov = (PI_F / 180.f) * CAMERA_FOV_DEGREES;

    cam->x = cross3(cam->dir, up);
    normalize_promoted(&cam->x);
    float kx = cam->width * fov / cam->height;
    cam->x.x = kx * cam->x.x;
    cam->x.y = kx * cam->x.y;
    cam->x.z = kx * cam->x.z;

    cam->y = cross3(cam->x, cam->dir);
    normalize_promoted(&cam->y);
    cam->y.x = fov * cam->y.x;
    cam->y.y = fov * cam->y.y;
    cam->y.z = fov * cam->y.z;
}

/* glibc's rand() with its default state: TYPE_3 additive feedback generator, degree 31,
 * separation 3, seeded with 1 (srand is never called by the reference).  Output k is
 * (r[k+344] >> 1) with r[i] = r[i-31] + r[i-3] (mod 2^32). */
typedef struct {
    uint32_t r[34];
    int i; /* ring position */
} glibc_rand_state;

static void glibc_rand_init(glibc_rand_state *st, uint32_t seed) {
    int32_t r[344];
    r[0] = (int32_t)seed;
    for (int i = 1; i < 31; ++i) {
        r[i] = (int32_t)((16807LL * r[i - 1]) % 2147483647);
        if (r[i] < 0) r[i] += 2147483647;
    }
    for (int i = 31; i < 34; ++i) r[i] = r[i - 31];
    for (int i = 34; i < 344; ++i) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
    /* keep the last 34 values as a ring */
    for (int i = 0; i < 34; ++i) st->r[i] = (uint32_t)r[344 - 34 + i];
    st->i = 0;
}

static inline uint32_t glibc_rand_next(glibc_rand_state *st) {
    /* ring of 34: slot i holds r[n-34]; r[n] = r[n-31] + r[n-3] */
    uint32_t v = st->r[(st->i + 3) % 34] + st->r[(st->i + 31) % 34];
    st->r[st->i] = v;
    st->i = (st->i + 1) % 34;
    return v >> 1;
}

/* reference main-gpu.c:91-96 */
void pthost_seed_block(uint32_t *seeds, size_t count, uint64_t skip) {
    glibc_rand_state st;
    glibc_rand_init(&st, 1u);
    for (uint64_t k = 0; k < skip; ++k) (void)glibc_rand_next(&st);
    for (size_t k = 0; k < count; ++k) {
        uint32_t v = glibc_rand_next(&st);
        seeds[k] = v < 2 ? 2 : v;
    }
}

void pthost_flip_rows_u32x2(uint32_t *seeds, int width, int height) {
    size_t row = (size_t)width * 2;
    uint32_t *tmp = (uint32_t *)malloc(row * sizeof(uint32_t));
    if (!tmp) return;
    for (int y = 0; y < height / 2; ++y) {
        uint32_t *a = seeds + (size_t)y * row;
        uint32_t *b = seeds + (size_t)(height - 1 - y) * row;
        memcpy(tmp, a, row * sizeof(uint32_t));
        memcpy(a, b, row * sizeof(uint32_t));
        memcpy(b, tmp, row * sizeof(uint32_t));
    }
    free(tmp);
}

/* reference displayfunc.c:504-518 */
int pthost_write_ppm(const char *path, const uint32_t *pixels, int width, int height) {
    FILE *f = fopen(path, "w");
    if (!f) {
        pthost_set_error("Failed to open image file: %s", path);
        return -1;
    }
    fprintf(f, "P3\n%d %d\n%d\n", width, height, 255);
    for (int y = height - 1; y >= 0; --y) {
        const unsigned char *p = (const unsigned char *)(&pixels[(size_t)y * width]);
        for (int x = 0; x < width; ++x, p += 4) fprintf(f, "%d %d %d ", p[0], p[1], p[2]);
    }
    fclose(f);
    return 0;
}

/* reference time-utils.c:18-32 */
static void human_time(double seconds, char *out, size_t cap) {
    if (seconds < 1)
        snprintf(out, cap, "%03.0fms", roundf((float)(seconds * 1000.f)));
    else if (seconds < 60)
        snprintf(out, cap, "%.0fs", roundf((float)seconds));
    else if (seconds < 3600)
        snprintf(out, cap, "%.0fm%.0fs", floorf((float)(seconds / 60.f)), fmod(floorf((float)seconds), 60.f));
    else
        snprintf(out, cap, "%.0fh%02.0fm", floorf((float)(seconds / 60.f / 60.f)),
                 fmod(floorf((float)(seconds / 60.f)), 60.f));
}

void pthost_snapshot_name(char *out, size_t out_len, const char *title, int passes,
                          double elapsed_seconds) {
    char t[32];
    human_time(elapsed_seconds, t, sizeof t);
    snprintf(out, out_len, "%s-%i-%s.ppm", title, passes, t);
}

/* reference main-gpu.c:846-857 */
void pthost_captions(char *line1, char *line2, size_t len, double elapsed_total,
                     double elapsed_frame, int current_sample, int samples_this_frame, int width,
                     int height) {
    char t[32];
    human_time(elapsed_total, t, sizeof t);
    snprintf(line1, len, "Time:   %s  (%.3fs frames/s)", t, elapsed_frame);
    const double per_sec = (double)samples_this_frame * height * width / elapsed_frame;
    snprintf(line2, len, "Passes: %d  (%.1fK samples/s)", current_sample, per_sec / 1000.f);
}

/* reference main-gpu.c:774-781: one pass per frame for the first 20 samples, then as many as
 * fit in 0.5 s * min(currentSample - 20, 100) / 100 */
float pthost_frame_time_budget(int current_sample) {
    if (current_sample < 20) return 0.f;
    int d = current_sample - 20;
    const float k = (d < 100 ? d : 100) / 100.f;
    return 0.5f * k;
}

/* ---- checkpoint / resume (no reference counterpart; format in pthost.h) ---- */
static const char kCkptMagic[8] = {'P', 'T', 'C', 'K', 'P', 'T', '0', '1'};

int pthost_checkpoint_write(const char *path, int width, int height, int current_sample,
                            const float *colors, const uint32_t *seeds) {
    if (!path || !colors || !seeds || width <= 0 || height <= 0 || current_sample < 0) {
        pthost_set_error("checkpoint: bad argument");
        return -1;
    }
    FILE *f = fopen(path, "wb");
    if (!f) {
        pthost_set_error("checkpoint: cannot create '%s'", path);
        return -1;
    }
    const int32_t hdr[3] = {width, height, current_sample};
    const size_t n = (size_t)width * height;
    int ok = fwrite(kCkptMagic, 1, 8, f) == 8 && fwrite(hdr, sizeof hdr, 1, f) == 1 &&
             fwrite(colors, 12, n, f) == n && fwrite(seeds, 8, n, f) == n;
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        pthost_set_error("checkpoint: short write to '%s'", path);
        return -1;
    }
    return 0;
}

int pthost_checkpoint_read(const char *path, int *width, int *height, int *current_sample,
                           float **colors, uint32_t **seeds) {
    if (!path || !width || !height || !current_sample || !colors || !seeds) {
        pthost_set_error("checkpoint: bad argument");
        return -1;
    }
    FILE *f = fopen(path, "rb");
    if (!f) {
        pthost_set_error("checkpoint: cannot open '%s'", path);
        return -1;
    }
    char magic[8];
    int32_t hdr[3];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kCkptMagic, 8) != 0 || fread(hdr, sizeof hdr, 1, f) != 1 ||
        hdr[0] <= 0 || hdr[1] <= 0 || hdr[2] < 0 || (long long)hdr[0] * hdr[1] > (1ll << 28)) {
        fclose(f);
        pthost_set_error("checkpoint: '%s' is not a PTCKPT01 file", path);
        return -1;
    }
    const size_t n = (size_t)hdr[0] * hdr[1];
    float *c = (float *)malloc(12 * n);
    uint32_t *s = (uint32_t *)malloc(8 * n);
    if (!c || !s || fread(c, 12, n, f) != n || fread(s, 8, n, f) != n) {
        fclose(f);
        free(c);
        free(s);
        pthost_set_error("checkpoint: '%s' is truncated", path);
        return -1;
    }
    fclose(f);
    *width = hdr[0];
    *height = hdr[1];
    *current_sample = hdr[2];
    *colors = c;
    *seeds = s;
    return 0;
}
