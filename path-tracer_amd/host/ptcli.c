/*
 * ptcli.c - headless C host for the MI355X radiance path: what main() and updateRendering() of the
 * reference's main-gpu.c do (main-gpu.c:919-962, 727-839), with the OpenCL plumbing replaced by
 * the ptgpu C-ABI and the GLUT window replaced by a pass budget and a PPM snapshot.
 *
 *   ptcli [<useGPU> <forceWorkSize> <kernel file> <width> <height> <scene file>]
 *         [--spp N] [--out image.ppm] [--integrator pt|dl] [--math fast|strict]
 *         [--kernel wavefront|mega] [--models DIR] [--device N] [--gpus N] [--script FILE] [--quiet]
 *
 * The six positional arguments are the reference's (main-gpu.c:924-930); with none, its defaults
 * are used (1 0 path-tracing.cl 800 600 scenes/stacked-spheres.txt).  <useGPU> and
 * <forceWorkSize> are accepted and ignored (there is one backend and it picks its own launch
 * shape); a <kernel file> named ray-tracing.cl selects the direct-lighting integrator, anything
 * else the path tracer - the reference selects the integrator by which .cl file it builds.
 * Errors print to stderr and exit(-1) like the reference.
 *
 * --gpus N (no counterpart in the reference, which drives one device: main-gpu.c:398-404): ONE process, N
 * contexts on devices <device> .. <device>+N-1.  The passes are split by disjoint seed streams - GPU g takes
 * the g-th block of 2*W*H rand() outputs and renders ceil(spp / N) passes into its own sample buffer - and
 * the buffers are summed in place with one RCCL all-reduce over xGMI (ncclCommInitAll + ncclAllReduce on
 * ptgpu_colors_device_ptr / ptgpu_stream), scaled by 1/N and re-packed on the first device.  A NaN sample
 * (the reference's black firefly) poisons its pixel's sum: the combined image shows the union
 * of every GPU's NaN pixels, as one GPU rendering all the passes from one seed stream would show its own.
 * --gpus 1 goes through the same code with a communicator of one (rehearsal on a one-GPU box).
 *
 * --script FILE: the reference's interactive session without its window - the events GLUT would deliver, one per
 * line, driving the same state changes (libpthost's pthost_input_*: displayfunc.c:408-447, 482-493, 496-671;
 * main-gpu.c:885-913) and, per `idle` line, one frame of updateRendering():
 *     idle <ms> [passes]    update() at GLUT_ELAPSED_TIME = ms: held WASD keys move the camera, then one frame is
 *                           rendered - `passes` passes if given, else the reference's schedule (one pass below 20
 *                           samples, then as many as fit 0.5 s * min(sample - 20, 100) / 100, main-gpu.c:774-794)
 *     key <c> | keyup <c>   c = a character, `space`, `esc` or a decimal code ('p' writes the snapshot PPM)
 *     shift down|up         mouse <button> <x> <y>         motion <x> <y>         reshape <w> <h>
 *     # comment
 * A camera change uploads the camera and restarts the running mean (sample 0) on the seeds as they stand; a reshape
 * re-allocates the image at the new size with the next block of rand() seeds.  GLUT fires one reshape when the
 * window appears: a script that wants the interactive run's seeds starts with `reshape <w> <h>`.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ctype.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "ptgpu.h"
#include "pthost.h"

#define MAX_GPUS 64

static double wall_clock(void) { /* reference time-utils.c:7-12 */
    struct timeval t;
    gettimeofday(&t, NULL);
    return t.tv_sec + t.tv_usec / 1000000.0;
}

static void die_gpu(const char *what) {
    fprintf(stderr, "%s: %s\n", what, ptgpu_last_error());
    exit(-1);
}

static void die_hip(const char *what, hipError_t e) {
    fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
    exit(-1);
}

static void die_rccl(const char *what, ncclResult_t r) {
    fprintf(stderr, "%s: %s\n", what, ncclGetErrorString(r));
    exit(-1);
}

/* The whole multi-GPU render (see the header comment).  Fills `pixels` with the combined image and returns
 * the passes rendered in total (N * ceil(spp / N)). */
static int render_on_gpus(int n_gpus, int first_device, int integrator, int math, int kernel, int cull,
                          const pthost_scene *sc, int width, int height, int spp, uint32_t *pixels, int quiet) {
    const size_t n = (size_t)width * height;
    ptgpu_ctx *ctx[MAX_GPUS];
    ncclComm_t comm[MAX_GPUS];
    int devs[MAX_GPUS];
    const int per_gpu = (spp + n_gpus - 1) / n_gpus;
    uint32_t *seeds = malloc(sizeof(uint32_t) * 2 * n);
    if (!seeds) {
        fprintf(stderr, "out of memory\n");
        exit(-1);
    }
    for (int g = 0; g < n_gpus; ++g) {
        devs[g] = first_device + g;
        if (ptgpu_create(&ctx[g], devs[g], integrator)) die_gpu("Failed to create the device context");
        if (ptgpu_set_option(ctx[g], PTGPU_OPT_MATH, math) || ptgpu_set_option(ctx[g], PTGPU_OPT_KERNEL, kernel) ||
            ptgpu_set_option(ctx[g], PTGPU_OPT_SHADOW_CULL, cull))
            die_gpu("Failed to set an option");
        if (ptgpu_set_scene(ctx[g], sc->objects, sc->object_count, sc->light_count, sc->materials,
                            sc->scene.material_count, &sc->scene))
            die_gpu("Failed to upload the scene");
        if (ptgpu_set_camera(ctx[g], &sc->camera)) die_gpu("Failed to upload the camera");
        pthost_seed_block(seeds, 2 * n, (uint64_t)g * 2 * n); /* disjoint seed streams */
        if (ptgpu_resize(ctx[g], width, height, seeds)) die_gpu("Failed to create the image buffers");
    }
    free(seeds);
    /* contexts (and their streams) first, the communicators after them: INTEGRATION.md, "Multi-GPU" */
    ncclResult_t r = ncclCommInitAll(comm, n_gpus, devs);
    if (r != ncclSuccess) die_rccl("ncclCommInitAll", r);

    /* every GPU's passes are enqueued before any of them is waited for */
    for (int done = 0; done < per_gpu; done += 64) {
        const int batch = per_gpu - done < 64 ? per_gpu - done : 64;
        for (int g = 0; g < n_gpus; ++g)
            if (ptgpu_render(ctx[g], done, batch)) die_gpu("Failed to enqueue work");
    }
    /* the one collective of the path: sum of the float4 sample buffers, in place, each on its context's stream */
    r = ncclGroupStart();
    if (r != ncclSuccess) die_rccl("ncclGroupStart", r);
    for (int g = 0; g < n_gpus; ++g) {
        void *colors = NULL, *stream = NULL;
        if (ptgpu_colors_device_ptr(ctx[g], &colors) || ptgpu_stream(ctx[g], &stream)) die_gpu("Failed to get the sample buffer");
        hipError_t e = hipSetDevice(devs[g]);
        if (e != hipSuccess) die_hip("hipSetDevice", e);
        r = ncclAllReduce(colors, colors, (size_t)PTGPU_COLOR_STRIDE_FLOATS * n, ncclFloat, ncclSum, comm[g], (hipStream_t)stream);
        if (r != ncclSuccess) die_rccl("ncclAllReduce", r);
    }
    r = ncclGroupEnd();
    if (r != ncclSuccess) die_rccl("ncclGroupEnd", r);
    for (int g = 0; g < n_gpus; ++g)
        if (ptgpu_finish(ctx[g])) die_gpu("Failed to finish");
    /* equal pass counts: the mean of the per-GPU means is the mean over all passes */
    if (ptgpu_scale_and_repack(ctx[0], 1.0f / (float)n_gpus)) die_gpu("Failed to re-pack the image");
    if (ptgpu_read_pixels(ctx[0], pixels)) die_gpu("Failed to read the pixel buffer");
    if (!quiet) fprintf(stderr, "%d GPU(s) x %d passes, one all-reduce of %zu floats\n", n_gpus, per_gpu, (size_t)PTGPU_COLOR_STRIDE_FLOATS * n);
    for (int g = 0; g < n_gpus; ++g) {
        ncclCommDestroy(comm[g]);
        ptgpu_destroy(ctx[g]);
    }
    return per_gpu * n_gpus;
}

/* passes handed to the device per call once past the 20-sample warm-up: 8, or 64 where the library runs the call
 * as ONE launch of the work-queue kernel (a launch per 8 passes would drain the chip eight times as often; the
 * reference's own loop is one pass and a clFinish at a time, main-gpu.c:782-793) */
static int passes_per_call(ptgpu_ctx *ctx) {
    ptgpu_launch_info li;
    if (ptgpu_get_launch_info(ctx, &li) == PTGPU_OK && li.work_queue) return 64;
    return 8;
}

/* one frame of updateRendering(), main-gpu.c:774-816: `fixed_passes` > 0 renders exactly that many passes,
 * otherwise the reference's schedule; ends with the blocking read of the packed pixels */
static int render_frame(ptgpu_ctx *ctx, int *current_sample, int fixed_passes, int spp_limit, uint32_t *pixels) {
    const double frame_start = wall_clock();
    const int first = *current_sample;
    if (fixed_passes > 0) {
        if (ptgpu_render(ctx, *current_sample, fixed_passes)) die_gpu("Failed to enqueue work");
        *current_sample += fixed_passes;
    } else if (*current_sample < 20) {
        if (ptgpu_render_pass(ctx, *current_sample)) die_gpu("Failed to enqueue work");
        (*current_sample)++;
    } else {
        const float budget = pthost_frame_time_budget(*current_sample);
        int per_call = 8;
        for (;;) {
            int batch = per_call;
            if (spp_limit > 0 && spp_limit - *current_sample < batch) batch = spp_limit - *current_sample;
            if (batch <= 0) break;
            if (ptgpu_render(ctx, *current_sample, batch)) die_gpu("Failed to enqueue work");
            if (ptgpu_finish(ctx)) die_gpu("Failed to finish");
            *current_sample += batch;
            per_call = passes_per_call(ctx);
            if ((spp_limit > 0 && *current_sample >= spp_limit) || wall_clock() - frame_start > budget) break;
        }
    }
    if (ptgpu_read_pixels(ctx, pixels)) die_gpu("Failed to read the pixel buffer");
    return *current_sample - first;
}

static int parse_key(const char *tok) {
    if (!strcmp(tok, "space")) return ' ';
    if (!strcmp(tok, "esc")) return 27;
    if (strlen(tok) == 1) return (unsigned char)tok[0];
    if (isdigit((unsigned char)tok[0])) return atoi(tok) & 255;
    return -1;
}

/* --script: see the header comment.  Returns the passes in the final image; *pixels_io may be re-allocated. */
static int run_script(const char *path, ptgpu_ctx *ctx, const pthost_scene *sc, int *width, int *height,
                      uint32_t **pixels_io, double start, int quiet) {
    FILE *f = fopen(path, "r");
    if (!f) {
        fprintf(stderr, "Failed to open script file: %s\n", path);
        exit(-1);
    }
    pthost_input in;
    pthost_input_init(&in, &sc->camera, *width, *height);
    if (ptgpu_set_camera(ctx, &in.camera)) die_gpu("Failed to upload the camera");
    int current_sample = 0, line_no = 0, frames = 0;
    char line[512];
    while (fgets(line, sizeof line, f)) {
        ++line_no;
        char cmd[32] = "", a[64] = "", b[64] = "", c[64] = "";
        const int n = sscanf(line, "%31s %63s %63s %63s", cmd, a, b, c);
        if (n < 1 || cmd[0] == '#') continue;
        int ev = PTHOST_EV_NONE, frame = 0, passes = 0;
        if (!strcmp(cmd, "idle") && n >= 2) { ev = pthost_input_idle(&in, atoi(a)); frame = 1; passes = n >= 3 ? atoi(b) : 0; }
        else if (!strcmp(cmd, "key") && n >= 2 && parse_key(a) >= 0) ev = pthost_input_keyboard(&in, (unsigned char)parse_key(a));
        else if (!strcmp(cmd, "keyup") && n >= 2 && parse_key(a) >= 0) ev = pthost_input_keyboard_up(&in, (unsigned char)parse_key(a));
        else if (!strcmp(cmd, "shift") && n >= 2) ev = pthost_input_shift(&in, !strcmp(a, "down"));
        else if (!strcmp(cmd, "mouse") && n >= 4) ev = pthost_input_mouse(&in, atoi(a), atoi(b), atoi(c));
        else if (!strcmp(cmd, "motion") && n >= 3) ev = pthost_input_motion(&in, atoi(a), atoi(b));
        else if (!strcmp(cmd, "reshape") && n >= 3 && atoi(a) > 0 && atoi(b) > 0) ev = pthost_input_reshape(&in, atoi(a), atoi(b));
        else {
            fprintf(stderr, "%s:%d: cannot read the event '%s'\n", path, line_no, cmd);
            exit(-1);
        }
        if (ev & PTHOST_EV_REALLOC) { /* reInitViewpointDependentBuffers(1), main-gpu.c:887-890 */
            *width = in.width;
            *height = in.height;
            const size_t px = (size_t)in.width * in.height;
            uint32_t *seeds = malloc(sizeof(uint32_t) * 2 * px);
            uint32_t *pixels = realloc(*pixels_io, sizeof(uint32_t) * px);
            if (!seeds || !pixels) {
                fprintf(stderr, "out of memory\n");
                exit(-1);
            }
            *pixels_io = pixels;
            for (size_t i = 0; i < px; ++i) pixels[i] = (uint32_t)i; /* main-gpu.c:99-102 */
            pthost_input_next_seeds(&in, seeds);
            if (ptgpu_resize(ctx, in.width, in.height, seeds)) die_gpu("Failed to create the image buffers");
            free(seeds);
        }
        if (ev & PTHOST_EV_CAMERA) { /* camera upload, currentSample = 0 (main-gpu.c:896-911) */
            if (ptgpu_set_camera(ctx, &in.camera)) die_gpu("Failed to upload the camera");
            current_sample = 0;
        }
        if (ev & PTHOST_EV_SNAPSHOT) { /* displayfunc.c:498-519: the pixels as last displayed */
            char name[300];
            pthost_snapshot_name(name, sizeof name, sc->title, current_sample, wall_clock() - start);
            if (pthost_write_ppm(name, *pixels_io, *width, *height)) fprintf(stderr, "%s\n", pthost_last_error());
            else if (!quiet) fprintf(stderr, "wrote %s\n", name);
        }
        if (ev & PTHOST_EV_QUIT) { /* displayfunc.c:527-530 */
            fprintf(stderr, "Done.\n");
            break;
        }
        if (frame) {
            const int done = render_frame(ctx, &current_sample, passes, 0, *pixels_io);
            ++frames;
            if (!quiet)
                fprintf(stderr, "frame %d: %d pass(es), sample %d, camera (%g %g %g) yaw %g pitch %g\n", frames, done, current_sample,
                        in.camera.orig.x, in.camera.orig.y, in.camera.orig.z, in.camera.yaw, in.camera.pitch);
        }
    }
    fclose(f);
    return current_sample;
}

int main(int argc, char *argv[]) {
    const char *scene_name = "scenes/stacked-spheres.txt", *kernel_file = "path-tracing.cl";
    const char *out_path = NULL, *model_dir = NULL, *script_path = NULL;
    int width = 800, height = 600, spp = 64, device = 0, quiet = 0, gpus = 0;
    int integrator = -1, math = PTGPU_MATH_FAST, kernel = PTGPU_KERNEL_AUTO, cull = PTGPU_CULL_AUTO;

    int npos = 0;
    const char *pos[6];
    for (int i = 1; i < argc; ++i) {
        const char *a = argv[i];
        const char *v = (i + 1 < argc) ? argv[i + 1] : NULL;
        if (!strcmp(a, "--spp") && v) { spp = atoi(v); ++i; }
        else if (!strcmp(a, "--out") && v) { out_path = v; ++i; }
        else if (!strcmp(a, "--models") && v) { model_dir = v; ++i; }
        else if (!strcmp(a, "--script") && v) { script_path = v; ++i; }
        else if (!strcmp(a, "--device") && v) { device = atoi(v); ++i; }
        else if (!strcmp(a, "--gpus") && v) { gpus = atoi(v); ++i; }
        else if (!strcmp(a, "--integrator") && v) { integrator = !strcmp(v, "dl") ? PT_INTEGRATOR_DL : PT_INTEGRATOR_PT; ++i; }
        else if (!strcmp(a, "--math") && v) { math = !strcmp(v, "strict") ? PTGPU_MATH_STRICT : PTGPU_MATH_FAST; ++i; }
        else if (!strcmp(a, "--kernel") && v) {
            kernel = !strcmp(v, "mega") ? PTGPU_KERNEL_MEGA : !strcmp(v, "wavefront") ? PTGPU_KERNEL_WAVEFRONT : PTGPU_KERNEL_AUTO;
            ++i;
        } else if (!strcmp(a, "--cull") && v) {
            cull = !strcmp(v, "off") ? PTGPU_CULL_OFF : !strcmp(v, "safe") ? PTGPU_CULL_SAFE
                   : !strcmp(v, "tight") ? PTGPU_CULL_TIGHT : !strcmp(v, "enclosed") ? PTGPU_CULL_ENCLOSED
                   : !strcmp(v, "ball") ? PTGPU_CULL_BALL : !strcmp(v, "self") ? PTGPU_CULL_ENCLOSED_SELF : PTGPU_CULL_AUTO;
            ++i;
        }
        else if (!strcmp(a, "--quiet")) quiet = 1;
        else if (a[0] == '-' && a[1] == '-') {
            fprintf(stderr, "Usage: %s [<use CPU/GPU device> <workgroup size> <kernel file name> <window width> "
                            "<window height> <scene file>] [--spp N] [--out image.ppm] [--integrator pt|dl] "
                            "[--math fast|strict] [--kernel auto|wavefront|mega] [--cull auto|off|safe|tight|enclosed|self|ball] [--models DIR] "
                            "[--device N] [--gpus N] [--script FILE]\n", argv[0]);
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
    if (gpus < 0 || gpus > MAX_GPUS) {
        fprintf(stderr, "--gpus must be 1..%d\n", MAX_GPUS);
        exit(-1);
    }
    if (gpus > 0) { /* passes split over GPUs by seed stream, one RCCL all-reduce (header comment) */
        const double t0 = wall_clock();
        const int rendered = render_on_gpus(gpus, device, integrator, math, kernel, cull, &sc, width, height, spp, pixels, quiet);
        const double dt = wall_clock() - t0;
        printf("%s: %d passes on %d GPU(s), %dx%d, %.3f s, %.2f Mpaths/s\n", sc.title, rendered, gpus, width, height, dt,
               (double)n * rendered / dt / 1e6);
        char multi_name[300];
        if (!out_path) {
            pthost_snapshot_name(multi_name, sizeof multi_name, sc.title, rendered, dt);
            out_path = multi_name;
        }
        if (pthost_write_ppm(out_path, pixels, width, height)) {
            fprintf(stderr, "%s\n", pthost_last_error());
            exit(-1);
        }
        if (!quiet) fprintf(stderr, "wrote %s\n", out_path);
        pthost_free_scene(&sc);
        free(seeds);
        free(pixels);
        return 0;
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
    if (script_path) {
        current_sample = run_script(script_path, ctx, &sc, &width, &height, &pixels, start, quiet);
    } else {
        while (current_sample < spp) {
            const double frame_start = wall_clock();
            const int done = render_frame(ctx, &current_sample, 0, spp, pixels);
            if (!quiet) {
                pthost_captions(line1, line2, sizeof line1, wall_clock() - start, wall_clock() - frame_start,
                                current_sample, done, width, height);
                fprintf(stderr, "%s | %s\n", line1, line2);
            }
        }
    }
    const double elapsed = wall_clock() - start;
    printf("%s: %d passes, %dx%d, %.3f s, %.2f Mpaths/s\n", sc.title, current_sample, width, height, elapsed,
           (double)width * height * current_sample / elapsed / 1e6);

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
