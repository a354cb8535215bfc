// ptgpu_api.hip - host side of libptgpu.so: the C-ABI of include/ptgpu.h.
//
// Stands where the OpenCL plumbing of the reference's main-gpu.c stands (buffer creation and
// blocking uploads :85-168, :516-586; kernel argument set-up and launch :682-794; read-back
// :803-827; release :58-80).  Kernels are ahead-of-time gfx950 code objects in this library; there
// is no JIT and no CPU fallback.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "device_params.h"
#include "ptgpu.h"

namespace {

thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(PTGPU_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

// LDS we are willing to spend on staged records per workgroup; beyond it the kernels read the
// records with wave-uniform loads from global memory instead (160 KiB per CU on gfx950).
constexpr unsigned kMaxStagedLdsBytes = 96 * 1024; // + 58 KiB of per-wave scratch in the wavefront kernel

// Regions of the image rendered on concurrent streams (the context's own stream is one of them):
// while one region's launch drains its last waves, the other regions' launches keep the CUs full.
// 4 = HIP's default number of hardware queues per process (GPU_MAX_HW_QUEUES).
constexpr int kMaxStreams = 4;

bool emits(const pt_vec3 &e) { return !((e.x == 0.f) && (e.x == 0.f) && (e.z == 0.f)); } // vec.h:23

} // namespace

struct ptgpu_ctx {
    int device = 0;
    int integrator = PT_INTEGRATOR_PT;
    int kernel = PTGPU_KERNEL_AUTO;
    int math = PTGPU_MATH_FAST;
    int want_lds = 1;
    int n_streams = kMaxStreams;      // regions rendered concurrently (PTGPU_OPT_STREAMS)
    int passes_per_launch = 0;        // 0 = automatic (PTGPU_OPT_PASSES_PER_LAUNCH)
    int waves_per_group = 0;          // 0 = default (PTGPU_OPT_WAVES_PER_GROUP)
    hipStream_t stream = nullptr;     // the context's stream: everything is ordered on it
    hipStream_t side[kMaxStreams - 1] = {};
    hipEvent_t fork_ev = nullptr, join_ev[kMaxStreams - 1] = {};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // scene
    float4 *d_prims = nullptr, *d_shade = nullptr, *d_lights = nullptr;
    int4 *d_segs = nullptr;
    bool have_scene = false, have_camera = false;
    ptgpu_scene_info info{};
    pt_scene scene{};
    pt_camera camera{};

    // image
    int width = 0, height = 0;
    float *d_colors = nullptr;
    uint32_t *d_pixels = nullptr, *d_seeds = nullptr;
};

namespace {

int bind(ptgpu_ctx *ctx) {
    if (!ctx) return fail(PTGPU_ERR_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return PTGPU_OK;
}

void free_scene(ptgpu_ctx *c) {
    if (c->d_prims) (void)hipFree(c->d_prims);
    if (c->d_shade) (void)hipFree(c->d_shade);
    if (c->d_lights) (void)hipFree(c->d_lights);
    if (c->d_segs) (void)hipFree(c->d_segs);
    c->d_prims = c->d_shade = c->d_lights = nullptr;
    c->d_segs = nullptr;
    c->have_scene = false;
}

void free_image(ptgpu_ctx *c) {
    if (c->d_colors) (void)hipFree(c->d_colors);
    if (c->d_pixels) (void)hipFree(c->d_pixels);
    if (c->d_seeds) (void)hipFree(c->d_seeds);
    c->d_colors = nullptr;
    c->d_pixels = c->d_seeds = nullptr;
    c->width = c->height = 0;
}

float as_float(int v) {
    float f;
    memcpy(&f, &v, sizeof f);
    return f;
}

PtDeviceParams make_params(const ptgpu_ctx *c, int first_sample, int n_passes) {
    PtDeviceParams p{};
    p.prims = c->d_prims;
    p.shade = c->d_shade;
    p.lights = c->d_lights;
    p.segs = c->d_segs;
    p.n_prims = (int)c->info.primitives;
    p.n_segs = (int)c->info.segments;
    p.n_lights = (int)c->info.lights;
    p.integrator = c->integrator;
    const pt_camera &k = c->camera;
    const float o[3] = {k.orig.x, k.orig.y, k.orig.z}, d[3] = {k.dir.x, k.dir.y, k.dir.z};
    const float x[3] = {k.x.x, k.x.y, k.x.z}, y[3] = {k.y.x, k.y.y, k.y.z};
    memcpy(p.cam_orig, o, sizeof o);
    memcpy(p.cam_dir, d, sizeof d);
    memcpy(p.cam_x, x, sizeof x);
    memcpy(p.cam_y, y, sizeof y);
    const float s1[3] = {c->scene.sky1.x, c->scene.sky1.y, c->scene.sky1.z};
    const float s2[3] = {c->scene.sky2.x, c->scene.sky2.y, c->scene.sky2.z};
    memcpy(p.sky1, s1, sizeof s1);
    memcpy(p.sky2, s2, sizeof s2);
    p.gamma = c->scene.gamma;
    p.width = c->width;
    p.height = c->height;
    p.current_sample = first_sample;
    p.n_passes = n_passes;
    p.colors = c->d_colors;
    p.pixels = c->d_pixels;
    p.seeds = c->d_seeds;
    p.tiles_x = (c->width + 7) / 8;
    p.n_tiles = p.tiles_x * ((c->height + 7) / 8);
    p.tile_first = 0;
    p.tile_stride = 1;
    p.n_launch_tiles = p.n_tiles;
    return p;
}

int enqueue(ptgpu_ctx *c, int first_sample, int n_passes) {
    if (!c->have_scene) return fail(PTGPU_ERR_STATE, "render before ptgpu_set_scene");
    if (!c->have_camera) return fail(PTGPU_ERR_STATE, "render before ptgpu_set_camera");
    if (!c->d_colors) return fail(PTGPU_ERR_STATE, "render before ptgpu_resize");
    if (first_sample < 0 || n_passes < 0) return fail(PTGPU_ERR_ARGUMENT, "negative sample index or pass count");
    if (n_passes == 0) return PTGPU_OK;
    PtLaunchConfig cfg{};
    cfg.use_lds = c->want_lds && c->info.lds_bytes > 0;
    cfg.lds_bytes = cfg.use_lds ? c->info.lds_bytes : 0;
    cfg.waves_per_group = c->waves_per_group;
    PtDeviceParams probe = make_params(c, first_sample, 1);
    // One region per stream when the image has enough tiles to keep every region's launch wide.
    // Cheap scenes (a pass is tens of microseconds) gain nothing from overlapping tails and lose to
    // the extra launches, so they stay on one stream with the passes looped inside the launch.
    const long long work = (long long)c->info.primitives * (1 + (long long)c->info.lights);
    const int regions = (c->n_streams > 1 && probe.n_tiles >= 2048 && work >= 128) ? c->n_streams : 1;
    // Passes done inside one launch: with several regions in flight a launch may be short (its
    // tail is covered by the other regions); a single region keeps 8 passes per launch.
    const int chunk = c->passes_per_launch > 0 ? c->passes_per_launch : (regions > 1 ? 1 : 8);

    const int kernel = c->kernel != PTGPU_KERNEL_AUTO ? c->kernel
                       : (c->info.lights >= 2 ? PTGPU_KERNEL_WAVEFRONT : PTGPU_KERNEL_MEGA);
    auto launch = [&](const PtDeviceParams &p, hipStream_t s) -> int {
        if (kernel == PTGPU_KERNEL_MEGA)
            return c->math == PTGPU_MATH_STRICT ? ptk_strict::launch_mega(p, cfg, s) : ptk_fast::launch_mega(p, cfg, s);
        return c->math == PTGPU_MATH_STRICT ? ptk_strict::launch_wavefront(p, cfg, s)
                                            : ptk_fast::launch_wavefront(p, cfg, s);
    };

    int launch_error = 0;
    if (regions > 1) { // fork: side streams start after everything already queued on the main one
        HIP_TRY(hipEventRecord(c->fork_ev, c->stream));
        for (int r = 1; r < regions; ++r) HIP_TRY(hipStreamWaitEvent(c->side[r - 1], c->fork_ev, 0));
    }
    for (int done = 0; done < n_passes; done += chunk) {
        const int n = (n_passes - done < chunk) ? n_passes - done : chunk;
        for (int r = 0; r < regions; ++r) {
            PtDeviceParams p = make_params(c, first_sample + done, n);
            p.tile_first = r;
            p.tile_stride = regions;
            p.n_launch_tiles = (p.n_tiles - r + regions - 1) / regions;
            const int rc = launch(p, r == 0 ? c->stream : c->side[r - 1]);
            if (rc != 0 && launch_error == 0) launch_error = rc;
        }
        if (launch_error) break;
    }
    if (regions > 1) { // join: the main stream continues only after every region is done
        for (int r = 1; r < regions; ++r) {
            HIP_TRY(hipEventRecord(c->join_ev[r - 1], c->side[r - 1]));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->join_ev[r - 1], 0));
        }
    }
    if (launch_error)
        return fail(PTGPU_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString((hipError_t)launch_error));
    return PTGPU_OK;
}

} // namespace

extern "C" {

const char *ptgpu_last_error(void) { return g_error; }

int ptgpu_create(ptgpu_ctx **out, int device_ordinal, int integrator) {
    if (!out) return fail(PTGPU_ERR_ARGUMENT, "ptgpu_create: null output pointer");
    *out = nullptr;
    if (integrator != PT_INTEGRATOR_PT && integrator != PT_INTEGRATOR_DL)
        return fail(PTGPU_ERR_ARGUMENT, "ptgpu_create: integrator must be 0 (path tracing) or 1 (direct lighting)");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(PTGPU_ERR_DEVICE, "no HIP device available (%s); libptgpu has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= count)
        return fail(PTGPU_ERR_ARGUMENT, "device ordinal %d out of range (0..%d)", device_ordinal, count - 1);
    HIP_TRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PTGPU_ERR_DEVICE, "device %d is %s; this library carries gfx950 code only",
                    device_ordinal, prop.gcnArchName);
    ptgpu_ctx *c = new (std::nothrow) ptgpu_ctx();
    if (!c) return fail(PTGPU_ERR_MEMORY, "out of host memory");
    c->device = device_ordinal;
    c->integrator = integrator;
    auto make_queues = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&c->ev0));
        HIP_TRY(hipEventCreate(&c->ev1));
        HIP_TRY(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
        for (int i = 0; i < kMaxStreams - 1; ++i) {
            HIP_TRY(hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&c->join_ev[i], hipEventDisableTiming));
        }
        return PTGPU_OK;
    };
    const int rc = make_queues();
    if (rc != PTGPU_OK) {
        char keep[sizeof g_error];
        memcpy(keep, g_error, sizeof keep);
        (void)ptgpu_destroy(c); // releases whatever was created
        memcpy(g_error, keep, sizeof keep);
        return rc;
    }
    *out = c;
    return PTGPU_OK;
}

int ptgpu_destroy(ptgpu_ctx *ctx) {
    if (!ctx) return PTGPU_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_scene(ctx);
    free_image(ctx);
    for (int i = 0; i < kMaxStreams - 1; ++i) {
        if (ctx->side[i]) (void)hipStreamSynchronize(ctx->side[i]);
        if (ctx->side[i]) (void)hipStreamDestroy(ctx->side[i]);
        if (ctx->join_ev[i]) (void)hipEventDestroy(ctx->join_ev[i]);
    }
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PTGPU_OK;
}

int ptgpu_set_option(ptgpu_ctx *ctx, int option, int value) {
    if (!ctx) return fail(PTGPU_ERR_ARGUMENT, "null context");
    switch (option) {
    case PTGPU_OPT_KERNEL:
        if (value != PTGPU_KERNEL_MEGA && value != PTGPU_KERNEL_WAVEFRONT && value != PTGPU_KERNEL_AUTO)
            return fail(PTGPU_ERR_ARGUMENT, "unknown kernel variant %d", value);
        ctx->kernel = value;
        return PTGPU_OK;
    case PTGPU_OPT_MATH:
        if (value != PTGPU_MATH_FAST && value != PTGPU_MATH_STRICT)
            return fail(PTGPU_ERR_ARGUMENT, "unknown math mode %d", value);
        ctx->math = value;
        return PTGPU_OK;
    case PTGPU_OPT_LDS:
        ctx->want_lds = value ? 1 : 0;
        return PTGPU_OK;
    case PTGPU_OPT_STREAMS:
        if (value < 1 || value > kMaxStreams)
            return fail(PTGPU_ERR_ARGUMENT, "streams must be 1..%d", kMaxStreams);
        ctx->n_streams = value;
        return PTGPU_OK;
    case PTGPU_OPT_WAVES_PER_GROUP:
        if (value < 0 || value > 16) return fail(PTGPU_ERR_ARGUMENT, "waves per group must be 0..16");
        ctx->waves_per_group = value;
        return PTGPU_OK;
    case PTGPU_OPT_PASSES_PER_LAUNCH:
        if (value < 0 || value > 64) return fail(PTGPU_ERR_ARGUMENT, "passes per launch must be 0..64");
        ctx->passes_per_launch = value;
        return PTGPU_OK;
    default:
        return fail(PTGPU_ERR_ARGUMENT, "unknown option %d", option);
    }
}

int ptgpu_set_scene(ptgpu_ctx *ctx, const pt_object *objects, uint32_t object_count,
                    uint32_t light_count, const pt_material *materials, uint32_t material_count,
                    const pt_scene *scene) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!objects || !scene || (!materials && material_count))
        return fail(PTGPU_ERR_ARGUMENT, "ptgpu_set_scene: null argument");
    if (object_count > (1u << 24)) return fail(PTGPU_ERR_ARGUMENT, "too many objects (%u)", object_count);

    std::vector<float4> prims, shade, lights;
    std::vector<int4> segs;
    ptgpu_scene_info info{};
    for (uint32_t i = 0; i < object_count; ++i) {
        const pt_object &o = objects[i];
        if (o.type != PT_SPHERE && o.type != PT_TRIANGLE) continue; // MODEL placeholders and junk are skipped
        const bool light = emits(o.emission);
        if (!light && (o.material_id < 0 || (uint32_t)o.material_id >= material_count))
            return fail(PTGPU_ERR_ARGUMENT,
                        "object %u uses material %d but only %u materials were given "
                        "(the reference would read past its material buffer)",
                        i, (int)o.material_id, material_count);
        const int p = (int)info.primitives++;
        if (o.type == PT_SPHERE) {
            info.spheres++;
            prims.push_back(make_float4(o.center.x, o.center.y, o.center.z, o.radius));
            prims.push_back(make_float4(o.radius * o.radius, 0.f, 0.f, 0.f));
            prims.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
        } else {
            info.triangles++;
            prims.push_back(make_float4(o.p1.x, o.p1.y, o.p1.z, 0.f));
            prims.push_back(make_float4(o.p2.x - o.p1.x, o.p2.y - o.p1.y, o.p2.z - o.p1.z, 0.f));
            prims.push_back(make_float4(o.p3.x - o.p1.x, o.p3.y - o.p1.y, o.p3.z - o.p1.z, 0.f));
        }
        if (segs.empty() || segs.back().x != (int)o.type)
            segs.push_back(make_int4((int)o.type, p, p + 1, 0));
        else
            segs.back().z = p + 1;

        pt_material m{};
        if (!light) m = materials[o.material_id];
        const int flags = ((int)m.type & 0xff) | ((light ? 1 : 0) << 8) | (((int)o.type & 0xff) << 16);
        shade.push_back(make_float4(o.emission.x, o.emission.y, o.emission.z, as_float(flags)));
        shade.push_back(make_float4(o.color.x, o.color.y, o.color.z, m.sigma));
        shade.push_back(make_float4(m.kd.x, m.kd.y, m.kd.z, m.A));
        shade.push_back(make_float4(m.B, 0.f, 0.f, 0.f));

        if (light && info.lights < light_count) { // geomfunc.h:220: stop after lightCount emitters
            info.lights++;
            const size_t q = prims.size() - 3;
            lights.push_back(make_float4(prims[q].x, prims[q].y, prims[q].z, as_float((int)o.type)));
            if (o.type == PT_SPHERE)
                lights.push_back(make_float4(o.radius, 0.f, 0.f, 0.f));
            else
                lights.push_back(prims[q + 1]);
            lights.push_back(prims[q + 2]);
            lights.push_back(make_float4(o.emission.x, o.emission.y, o.emission.z, o.area));
        }
    }
    info.segments = (uint32_t)segs.size();
    const size_t staged = (prims.size() + lights.size()) * sizeof(float4);
    info.lds_bytes = staged <= kMaxStagedLdsBytes ? (uint32_t)staged : 0;

    HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_scene(ctx);
    auto upload = [&](auto *&dst, const auto &src) -> hipError_t {
        const size_t bytes = (src.size() ? src.size() : 1) * sizeof(src[0]);
        hipError_t e = hipMalloc((void **)&dst, bytes);
        if (e != hipSuccess) return e;
        if (src.size()) e = hipMemcpy(dst, src.data(), src.size() * sizeof(src[0]), hipMemcpyHostToDevice);
        return e;
    };
    HIP_TRY(upload(ctx->d_prims, prims));
    HIP_TRY(upload(ctx->d_shade, shade));
    HIP_TRY(upload(ctx->d_lights, lights));
    HIP_TRY(upload(ctx->d_segs, segs));
    ctx->info = info;
    ctx->scene = *scene;
    ctx->have_scene = true;
    return PTGPU_OK;
}

int ptgpu_get_scene_info(ptgpu_ctx *ctx, ptgpu_scene_info *info) {
    if (!ctx || !info) return fail(PTGPU_ERR_ARGUMENT, "null argument");
    if (!ctx->have_scene) return fail(PTGPU_ERR_STATE, "no scene set");
    *info = ctx->info;
    return PTGPU_OK;
}

int ptgpu_set_camera(ptgpu_ctx *ctx, const pt_camera *camera) {
    if (!ctx || !camera) return fail(PTGPU_ERR_ARGUMENT, "null argument");
    // the camera travels as a kernel argument; work already queued keeps the one it was given
    ctx->camera = *camera;
    ctx->have_camera = true;
    return PTGPU_OK;
}

int ptgpu_resize(ptgpu_ctx *ctx, int width, int height, const uint32_t *seeds) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 28))
        return fail(PTGPU_ERR_ARGUMENT, "image size %dx%d out of range", width, height);
    if (!seeds) return fail(PTGPU_ERR_ARGUMENT, "ptgpu_resize: null seeds");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_image(ctx);
    const size_t n = (size_t)width * height;
    HIP_TRY(hipMalloc((void **)&ctx->d_colors, n * 3 * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&ctx->d_pixels, n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&ctx->d_seeds, n * 2 * sizeof(uint32_t)));
    HIP_TRY(hipMemset(ctx->d_colors, 0, n * 3 * sizeof(float)));
    HIP_TRY(hipMemset(ctx->d_pixels, 0, n * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(ctx->d_seeds, seeds, n * 2 * sizeof(uint32_t), hipMemcpyHostToDevice));
    ctx->width = width;
    ctx->height = height;
    return PTGPU_OK;
}

int ptgpu_render_pass(ptgpu_ctx *ctx, int current_sample) {
    int rc = bind(ctx);
    if (rc) return rc;
    return enqueue(ctx, current_sample, 1);
}

int ptgpu_render(ptgpu_ctx *ctx, int first_sample, int n_passes) {
    int rc = bind(ctx);
    if (rc) return rc;
    return enqueue(ctx, first_sample, n_passes);
}

int ptgpu_render_timed(ptgpu_ctx *ctx, int first_sample, int n_passes, float *elapsed_ms) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!elapsed_ms) return fail(PTGPU_ERR_ARGUMENT, "null elapsed_ms");
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    rc = enqueue(ctx, first_sample, n_passes);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return PTGPU_OK;
}

int ptgpu_finish(ptgpu_ctx *ctx) {
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PTGPU_OK;
}

static int read_back(ptgpu_ctx *ctx, void *dst, const void *src, size_t bytes, const char *what) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!dst) return fail(PTGPU_ERR_ARGUMENT, "%s: null destination", what);
    if (!src) return fail(PTGPU_ERR_STATE, "%s before ptgpu_resize", what);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PTGPU_OK;
}

int ptgpu_read_pixels(ptgpu_ctx *ctx, uint32_t *pixels) {
    if (!ctx) return fail(PTGPU_ERR_ARGUMENT, "null context");
    return read_back(ctx, pixels, ctx->d_pixels, (size_t)ctx->width * ctx->height * sizeof(uint32_t),
                     "ptgpu_read_pixels");
}

int ptgpu_read_colors(ptgpu_ctx *ctx, float *colors) {
    if (!ctx) return fail(PTGPU_ERR_ARGUMENT, "null context");
    return read_back(ctx, colors, ctx->d_colors, (size_t)ctx->width * ctx->height * 3 * sizeof(float),
                     "ptgpu_read_colors");
}

int ptgpu_read_seeds(ptgpu_ctx *ctx, uint32_t *seeds) {
    if (!ctx) return fail(PTGPU_ERR_ARGUMENT, "null context");
    return read_back(ctx, seeds, ctx->d_seeds, (size_t)ctx->width * ctx->height * 2 * sizeof(uint32_t),
                     "ptgpu_read_seeds");
}

int ptgpu_write_colors(ptgpu_ctx *ctx, const float *colors) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!colors) return fail(PTGPU_ERR_ARGUMENT, "ptgpu_write_colors: null source");
    if (!ctx->d_colors) return fail(PTGPU_ERR_STATE, "ptgpu_write_colors before ptgpu_resize");
    HIP_TRY(hipMemcpyAsync(ctx->d_colors, colors, (size_t)ctx->width * ctx->height * 3 * sizeof(float),
                           hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PTGPU_OK;
}

int ptgpu_colors_device_ptr(ptgpu_ctx *ctx, void **device_ptr) {
    if (!ctx || !device_ptr) return fail(PTGPU_ERR_ARGUMENT, "null argument");
    if (!ctx->d_colors) return fail(PTGPU_ERR_STATE, "no image buffers yet");
    *device_ptr = ctx->d_colors;
    return PTGPU_OK;
}

int ptgpu_stream(ptgpu_ctx *ctx, void **hip_stream) {
    if (!ctx || !hip_stream) return fail(PTGPU_ERR_ARGUMENT, "null argument");
    *hip_stream = (void *)ctx->stream;
    return PTGPU_OK;
}

int ptgpu_scale_and_repack(ptgpu_ctx *ctx, float scale) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!ctx->d_colors || !ctx->have_scene) return fail(PTGPU_ERR_STATE, "no image or scene yet");
    const PtDeviceParams p = make_params(ctx, 0, 0);
    rc = ptk_fast::launch_repack(p, scale, ctx->stream);
    if (rc != 0) return fail(PTGPU_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return PTGPU_OK;
}

} // extern "C"
