/*
 * ptcli.c - headless C host for the MI355X radiance path: what main() and updateRendering() of the
 * reference's main-gpu.c do (main-gpu.c:919-962, 727-839), with the OpenCL plumbing replaced by
 * the ptgpu C-ABI and the GLUT window replaced by a pass budget and a PPM snapshot.
 *
 *   ptcli [<useGPU> <forceWorkSize> <kernel file> <width> <height> <scene file>]
 *         [--spp N] [--out image.ppm] [--integrator pt|dl] [--math fast|strict]
 *         [--kernel wavefront|mega] [--models DIR] [--device N] [--quiet]
 *
 * The six positional arguments are the reference's (main-gpu.c:924-930); with none, its defaults
 * are used (1 0 path-tracing.cl 800 600 scenes/stacked-spheres.txt).  <useGPU> and
 * <forceWorkSize> are accepted and ignored (there is one backend and it picks its own launch
 * shape); a <kernel file> named ray-tracing.cl selects the direct-lighting integrator, anything
 * else the path tracer - the reference selects the integrator by which .cl file it builds.
 * Errors print to stderr and exit(-1) like the reference.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "ptgpu.h"
#include "pthost.h"

static double wall_clock(void) { /* reference time-utils.c:7-12 */
    struct timeval t;
    gettimeofday(&t, NULL);
    return t.tv_sec + t.tv_usec / 1000000.0;
}

static void die_gpu(const char *what) {
    fprintf(stderr, "%s: %s\n", what, ptgpu_last_error());
    exit(-1);
}

int main(int argc, char *argv[]) {
    const char *scene_name = "scenes/stacked-spheres.txt", *kernel_file = "path-tracing.cl";
    const char *out_path = NULL, *model_dir = NULL;
    int width = 800, height = 600, spp = 64, device = 0, quiet = 0;
    int integrator = -1, math = PTGPU_MATH_FAST, kernel = PTGPU_KERNEL_AUTO, cull = PTGPU_CULL_AUTO;

    int npos = 0;
    const char *pos[6];
    for (int i = 1; i < argc; ++i) {
        const char *a = argv[i];
        const char *v = (i + 1 < argc) ? argv[i + 1] : NULL;
        if (!strcmp(a, "--spp") && v) { spp = atoi(v); ++i; }
        else if (!strcmp(a, "--out") && v) { out_path = v; ++i; }
        else if (!strcmp(a, "--models") && v) { model_dir = v; ++i; }
        else if (!strcmp(a, "--device") && v) { device = atoi(v); ++i; }
        else if (!strcmp(a, "--integrator") && v) { integrator = !strcmp(v, "dl") ? PT_INTEGRATOR_DL : PT_INTEGRATOR_PT; ++i; }
        else if (!strcmp(a, "--math") && v) { math = !strcmp(v, "strict") ? PTGPU_MATH_STRICT : PTGPU_MATH_FAST; ++i; }
        else if (!strcmp(a, "--kernel") && v) {
            kernel = !strcmp(v, "mega") ? PTGPU_KERNEL_MEGA : !strcmp(v, "wavefront") ? PTGPU_KERNEL_WAVEFRONT : PTGPU_KERNEL_AUTO;
            ++i;
        } else if (!strcmp(a, "--cull") && v) {
            cull = !strcmp(v, "off") ? PTGPU_CULL_OFF : !strcmp(v, "safe") ? PTGPU_CULL_SAFE
                   : !strcmp(v, "tight") ? PTGPU_CULL_TIGHT : !strcmp(v, "enclosed") ? PTGPU_CULL_ENCLOSED
                   : !strcmp(v, "ball") ? PTGPU_CULL_BALL : PTGPU_CULL_AUTO;
            ++i;
        }
        else if (!strcmp(a, "--quiet")) quiet = 1;
        else if (a[0] == '-' && a[1] == '-') {
            fprintf(stderr, "Usage: %s [<use CPU/GPU device> <workgroup size> <kernel file name> <window width> "
                            "<window height> <scene file>] [--spp N] [--out image.ppm] [--integrator pt|dl] "
                            "[--math fast|strict] [--kernel auto|wavefront|mega] [--cull auto|off|safe|tight|enclosed|ball] [--models DIR] "
                            "[--device N]\n", argv[0]);
            exit(-1);
        } else if (npos < 6) pos[npos++] = a;
    }
    if (npos == 6) {
        kernel_file = pos[2];
        width = atoi(pos[3]);
        height = atoi(pos[4]);
        scene_name = pos[5];
    } else if (npos != 0) {
        exit(-1); /* main-gpu.c:944-946 */
    }
    if (integrator < 0) integrator = strstr(kernel_file, "ray-tracing") ? PT_INTEGRATOR_DL : PT_INTEGRATOR_PT;

    pthost_scene sc;
    if (pthost_load_scene(scene_name, model_dir, width, height, &sc) != 0) {
        fprintf(stderr, "%s\n", pthost_last_error());
        exit(-1);
    }
    pthost_update_camera_basis(&sc.camera);
    if (!quiet) fprintf(stderr, "FegemoPT: %s (%u objects, %u lights)\n", scene_name, sc.object_count, sc.light_count);

    const size_t n = (size_t)width * height;
    uint32_t *seeds = malloc(sizeof(uint32_t) * 2 * n);
    uint32_t *pixels = malloc(sizeof(uint32_t) * n);
    if (!seeds || !pixels) {
        fprintf(stderr, "out of memory\n");
        exit(-1);
    }
    pthost_seed_block(seeds, 2 * n, 0);

    ptgpu_ctx *ctx = NULL;
    if (ptgpu_create(&ctx, device, integrator)) die_gpu("Failed to create the device context");
    if (ptgpu_set_option(ctx, PTGPU_OPT_MATH, math) || ptgpu_set_option(ctx, PTGPU_OPT_KERNEL, kernel) ||
        ptgpu_set_option(ctx, PTGPU_OPT_SHADOW_CULL, cull))
        die_gpu("Failed to set an option");
    if (ptgpu_set_scene(ctx, sc.objects, sc.object_count, sc.light_count, sc.materials,
                        sc.scene.material_count, &sc.scene))
        die_gpu("Failed to upload the scene");
    if (ptgpu_set_camera(ctx, &sc.camera)) die_gpu("Failed to upload the camera");
    if (ptgpu_resize(ctx, width, height, seeds)) die_gpu("Failed to create the image buffers");

    /* the progressive loop of updateRendering(), main-gpu.c:774-794: one pass per frame for the
     * first 20 samples, then as many passes as fit the frame's time budget; every frame ends with
     * the blocking read of the packed pixels (main-gpu.c:803-816). */
    const double start = wall_clock();
    int current_sample = 0;
    char line1[256], line2[256];
    while (current_sample < spp) {
        const double frame_start = wall_clock();
        const int first = current_sample;
        const float budget = pthost_frame_time_budget(current_sample);
        if (current_sample < 20) {
            if (ptgpu_render_pass(ctx, current_sample)) die_gpu("Failed to enqueue work");
            current_sample++;
        } else {
            for (;;) {
                /* the reference runs one pass then clFinish; we hand the device up to 8 at once */
                int batch = spp - current_sample < 8 ? spp - current_sample : 8;
                if (ptgpu_render(ctx, current_sample, batch)) die_gpu("Failed to enqueue work");
                if (ptgpu_finish(ctx)) die_gpu("Failed to finish");
                current_sample += batch;
                if (current_sample >= spp || wall_clock() - frame_start > budget) break;
            }
        }
        if (ptgpu_read_pixels(ctx, pixels)) die_gpu("Failed to read the pixel buffer");
        if (!quiet) {
            pthost_captions(line1, line2, sizeof line1, wall_clock() - start, wall_clock() - frame_start,
                            current_sample, current_sample - first, width, height);
            fprintf(stderr, "%s | %s\n", line1, line2);
        }
    }
    const double elapsed = wall_clock() - start;
    printf("%s: %d passes, %dx%d, %.3f s, %.2f Mpaths/s\n", sc.title, current_sample, width, height, elapsed,
           (double)n * current_sample / elapsed / 1e6);

    char name[300];
    if (!out_path) { /* the 'p' key's file name, displayfunc.c:500-503 */
        pthost_snapshot_name(name, sizeof name, sc.title, current_sample, elapsed);
        out_path = name;
    }
    if (pthost_write_ppm(out_path, pixels, width, height)) {
        fprintf(stderr, "%s\n", pthost_last_error());
        exit(-1);
    }
    if (!quiet) fprintf(stderr, "wrote %s\n", out_path);
    ptgpu_destroy(ctx);
    pthost_free_scene(&sc);
    free(seeds);
    free(pixels);
    return 0;
}
