/*
 * radiance_oracle.c - TEST INFRASTRUCTURE (see pt_oracle.h): CPU restatement of the reference's
 * per-pixel radiance path, bit-identical to gcc's build of the reference (parity PINNED, see
 * pt_oracle.h).  It is the checker for the HIP kernels and the timed CPU baseline in bench.py;
 * the product never calls it.
 *
 * Follows: simplernd.h:11-25 (RNG), main-cpu.c:54-110 (pixel loop, running mean, 8-bit pack),
 * geomfunc.h:10-193 (primitive tests, closest/any hit), :199-306 (light sampling), :424-454
 * (cosine direction), :487-492 (sky), :494-798 and :800-1014 (the two integrators), vec.h (the
 * arithmetic forms and host-side clamp/max/sign macros).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <string.h>

#include "pt_oracle.h"

#define PI_F 3.14159265358979323846f
#define TWO_PI_F 6.28318530717958647693f

typedef pt_vec3 v3;

/* host-side macro semantics of vec.h:29-32 (NaN falls through both comparisons) */
#define clamp_macro(x, a, b) ((x) < (a) ? (a) : ((x) > (b) ? (b) : (x)))
#define fmax_macro(x, y) ((x) > (y) ? (x) : (y))

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(float k, v3 a) { return v3_make(k * a.x, k * a.y, k * a.z); }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* vec.h:20: l = 1.f / sqrt(dot) evaluated in double, rounded to float, then three multiplies */
static inline v3 v3_normalize_promoted(v3 a) {
    const float l = (float)(1.0 / sqrt((double)v3_dot(a, a)));
    return v3_scale(l, a);
}
/* vec.h:23: the upstream "is zero" test looks at x twice and never at y */
static inline int v3_is_black(v3 e) { return (e.x == 0.f) && (e.x == 0.f) && (e.z == 0.f); }

/* simplernd.h:11-25: two 16-bit multiply-with-carry streams, 23 mantissa bits -> [0,1) */
static inline float mwc_draw(uint32_t *s0, uint32_t *s1) {
    *s0 = 36969u * ((*s0) & 65535u) + ((*s0) >> 16);
    *s1 = 18000u * ((*s1) & 65535u) + ((*s1) >> 16);
    const uint32_t bits = ((((*s0) << 16) + (*s1)) & 0x007fffffu) | 0x40000000u;
    float f;
    memcpy(&f, &bits, sizeof f);
    return (f - 2.f) / 2.f;
}

/* main-cpu.c:100-103 with vec.h:36: (int)(pow(clamp(x,0,1), 1/g) * 255 + .5) per channel, R in
 * the low byte.  A NaN channel converts to INT_MIN on x86 (cvttsd2si), and the three shifted
 * terms OR to 0x80000000; written out so the result does not rest on undefined behaviour. */
static inline uint32_t pack_channel(float x, float gamma) {
    const float c = clamp_macro(x, 0.f, 1.f);
    const double v = pow((double)c, (double)(1.f / gamma)) * 255.f + 0.5f;
    if (v != v) return 0x80000000u;
    return (uint32_t)(int)v;
}

uint32_t oracle_accumulate(pt_vec3 *color, const pt_vec3 *sample, int current_sample, float gamma) {
    if (current_sample == 0) {
        *color = *sample;
    } else {
        const float k1 = current_sample;
        const float k2 = 1.f / (k1 + 1.f);
        color->x = (color->x * k1 + sample->x) * k2;
        color->y = (color->y * k1 + sample->y) * k2;
        color->z = (color->z * k1 + sample->z) * k2;
    }
    return pack_channel(color->x, gamma) | (pack_channel(color->y, gamma) << 8) |
           (pack_channel(color->z, gamma) << 16);
}

/* Optional replay of the product's shadow-ray culling (NULL = the reference's full scan):
 * keep[l * n_objects + i], l = light in sampling order.  Set between renders, read-only during. */
static const uint8_t *g_shadow_keep = NULL;
void oracle_set_shadow_candidates(const uint8_t *keep) { g_shadow_keep = keep; }
/* ... and, for the ENCLOSED family of those sets, which objects are spheres the rule dropped for some light
 * (flags[i] != 0, n_objects bytes; NULL = none): closest_hit then reports whether every such sphere answered with
 * its far root (the path vertex is inside them all) */
static const uint8_t *g_enclosing = NULL;
void oracle_set_enclosing(const uint8_t *flags) { g_enclosing = flags; }

/* Diagnostic of the replay (serial renders only): every shadow ray whose culled scan answers differently
 * from the reference's full scan is logged - origin, direction, maxt, light (sampling order), the object the
 * ray starts on, the first object the full scan reports, its distance: 12 floats a record. */
static float *g_mismatch_log = NULL;
static int g_mismatch_capacity = 0, g_mismatch_count = 0;
void oracle_set_mismatch_log(float *records, int capacity) {
    g_mismatch_log = records;
    g_mismatch_capacity = capacity;
    g_mismatch_count = 0;
}
int oracle_mismatch_count(void) { return g_mismatch_count; }

/* Diagnostic (serial renders, plain instance): every ray the render traces, 12 floats a record - kind (0 extension,
 * 1 shadow), origin, direction, maxt (1e20 for an extension ray), answer (extension: distance, 0 for a miss; shadow:
 * 1 + the first occluder in index order, 0 if free), extension: object hit or -1 / shadow: the occluder's distance,
 * the slot y * width + x of the pixel, the segment of the path.  Used by
 * scripts/flipped_rays.py to put the same rays to the GPU builds through the ray probe. */
static float *g_ray_log = NULL;
static long g_ray_capacity = 0, g_ray_count = 0;
static int g_ray_pixel = 0, g_ray_depth = 0;
void oracle_set_ray_log(float *records, long capacity) {
    g_ray_log = records;
    g_ray_capacity = capacity;
    g_ray_count = 0;
}
long oracle_ray_count(void) { return g_ray_count; }
static void log_ray(int kind, const pt_ray *ray, float maxt, float answer, float id) {
    if (!g_ray_log || g_ray_count >= g_ray_capacity) return;
    float *r = g_ray_log + 12 * g_ray_count++;
    r[0] = (float)kind;
    r[1] = ray->o.x; r[2] = ray->o.y; r[3] = ray->o.z;
    r[4] = ray->d.x; r[5] = ray->d.y; r[6] = ray->d.z;
    r[7] = maxt; r[8] = answer; r[9] = id; r[10] = (float)g_ray_pixel; r[11] = (float)g_ray_depth;
}

/* ---- instance 1: plain ---- */
#define FN(name) plain_##name
#define TALLY(field) ((void)0)
#define CTR_PARAM
#define CTR_ARG
static inline float plain_draw(uint32_t *s0, uint32_t *s1) { return mwc_draw(s0, s1); }
static inline void plain_camera_ray(const pt_camera *cam, int x, int y, int width, int height,
                                    uint32_t *s0, uint32_t *s1, pt_ray *out);
#include "camera_ray_impl.h"
#include "radiance_oracle_impl.h"
#undef FN
#undef TALLY
#undef CTR_PARAM
#undef CTR_ARG

/* ---- instance 2: counting ---- */
#define ORACLE_COUNTING 1
#define FN(name) counted_##name
#define TALLY(field) (ctr->field++)
#define CTR_PARAM , pt_oracle_counters *ctr
#define CTR_ARG , ctr
static inline float counted_draw(uint32_t *s0, uint32_t *s1, pt_oracle_counters *ctr) {
    ctr->rng_draws++;
    return mwc_draw(s0, s1);
}
static inline void counted_camera_ray(const pt_camera *cam, int x, int y, int width, int height,
                                      uint32_t *s0, uint32_t *s1, pt_ray *out,
                                      pt_oracle_counters *ctr);
#include "camera_ray_impl.h"
#include "radiance_oracle_impl.h"
#undef FN
#undef TALLY
#undef CTR_PARAM
#undef CTR_ARG

/* ---- public entry points ---- */

float oracle_get_random(uint32_t *seed0, uint32_t *seed1) { return mwc_draw(seed0, seed1); }

void oracle_camera_ray(const pt_camera *cam, int x, int y, int width, int height, uint32_t *seed0,
                       uint32_t *seed1, pt_ray *out) {
    plain_camera_ray(cam, x, y, width, height, seed0, seed1, out);
}

int oracle_intersect(const pt_object *objects, uint32_t n, const pt_ray *ray, float *t, uint32_t *id) {
    return plain_closest_hit(objects, n, ray, t, id, NULL);
}

int oracle_intersect_p(const pt_object *objects, uint32_t n, const pt_ray *ray, float maxt) {
    return plain_any_hit(objects, n, ray, maxt, NULL, UINT32_MAX, 0);
}

void oracle_sample_lights(const pt_object *objects, uint32_t n, uint32_t lights, uint32_t *seed0,
                          uint32_t *seed1, const pt_vec3 *hit, const pt_vec3 *normal,
                          pt_vec3 *result) {
    plain_direct_light(objects, n, lights, seed0, seed1, hit, normal, UINT32_MAX, 0, result);
}

void oracle_radiance(const pt_object *objects, uint32_t n, uint32_t lights, const pt_ray *ray,
                     uint32_t *seed0, uint32_t *seed1, const pt_scene *scene,
                     const pt_material *materials, int integrator, pt_vec3 *radiance) {
    plain_radiance(objects, n, lights, ray, seed0, seed1, scene->sky1, scene->sky2, materials,
                   integrator, radiance);
}

void oracle_render_pass(const pt_object *objects, uint32_t n, uint32_t lights,
                        const pt_material *materials, const pt_camera *cam, const pt_scene *scene,
                        int width, int height, uint32_t *seeds, pt_vec3 *colors, uint32_t *pixels,
                        int current_sample, int integrator, int seed_layout, int threads) {
    if (threads <= 1) {
        plain_render_rows(objects, n, lights, materials, cam, scene, width, height, seeds, colors,
                          pixels, current_sample, integrator, seed_layout, 0, height);
        return;
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int y = 0; y < height; y++)
        plain_render_rows(objects, n, lights, materials, cam, scene, width, height, seeds, colors,
                          pixels, current_sample, integrator, seed_layout, y, y + 1);
}

void oracle_render_pass_counted(const pt_object *objects, uint32_t n, uint32_t lights,
                                const pt_material *materials, const pt_camera *cam,
                                const pt_scene *scene, int width, int height, uint32_t *seeds,
                                pt_vec3 *colors, uint32_t *pixels, int current_sample,
                                int integrator, int seed_layout, pt_oracle_counters *counters) {
    counted_render_rows(objects, n, lights, materials, cam, scene, width, height, seeds, colors,
                        pixels, current_sample, integrator, seed_layout, 0, height, counters);
}

int oracle_max_threads(void) { return omp_get_max_threads(); }
