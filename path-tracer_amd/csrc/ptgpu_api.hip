// ptgpu_api.hip - host side of libptgpu.so: the C-ABI of include/ptgpu.h.
//
// Stands where the OpenCL plumbing of the reference's main-gpu.c stands (buffer creation and
// blocking uploads :85-168, :516-586; kernel argument set-up and launch :682-794; read-back
// :803-827; release :58-80).  Kernels are ahead-of-time gfx950 code objects in this library; there
// is no JIT and no CPU fallback.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <mutex>
#include <cstring>
#include <new>
#include <vector>

#include "bvh_build.h"
#include "device_params.h"
#include "ptgpu.h"

// ptgpu_render_static: the work queue of the wavefront kernel for callers that cannot allocate.
// One copy per GPU, held until the process ends; every call on a GPU shares that GPU's copy.
// In host code the name below is only a stand-in for the symbol: its device address is looked up once
// per device (static_queue_of) and only that address is ever handed to memsets, copies and kernels.
static constexpr size_t kStaticQueueWords = (size_t)PTGPU_STATIC_MAX_TILES + 2; // give-up word, counter, one word per tile
static __device__ __align__(256) int scratch[kStaticQueueWords];

static constexpr int kMaxDevices = 64;
static std::mutex g_static_mutex;
static int *g_static_queue[kMaxDevices] = {}; // per device: address of ITS copy of `scratch`
// the context whose ptgpu_render_static calls are in flight on the device's queue (null: free).  Calls of ONE context are
// ordered on its stream and may follow each other; another context's call would zero a queue in flight, so it is
// refused until the owner's ptgpu_finish (or a read-back, or its destruction) has seen the queue drained.
static const void *g_static_owner[kMaxDevices] = {};
static bool static_queue_claim(int device, const void *ctx) {
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (g_static_owner[device] && g_static_owner[device] != ctx) return false;
    g_static_owner[device] = ctx;
    return true;
}
static void static_queue_release(int device, const void *ctx) {
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (device >= 0 && device < kMaxDevices && g_static_owner[device] == ctx) g_static_owner[device] = nullptr;
}

// `device` must be the current device.  Allocates nothing: the array is part of the loaded code object.
static int static_queue_of(int device, int **out) {
    if (device < 0 || device >= kMaxDevices) return -1;
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (!g_static_queue[device]) {
        void *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return (int)(e != hipSuccess ? e : hipErrorInvalidSymbol);
        g_static_queue[device] = (int *)ptr;
    }
    *out = g_static_queue[device];
    return 0;
}

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

#ifndef PT_DEBUG_KNOBS
#define PT_DEBUG_KNOBS 0 // getenv switches of the measurement scripts: compiled into the variant builds only
#endif
#ifndef PT_COOP_STRAGGLERS
#define PT_COOP_STRAGGLERS 24 // measured 16 .. 48 on obj-model.txt, its bunny variant and large-object.txt
#endif
struct ptgpu_ctx {
    int device = 0;
    int integrator = PT_INTEGRATOR_PT;
    int kernel = PTGPU_KERNEL_AUTO;
    int math = PTGPU_MATH_FAST;
    int want_lds = 1;
    int n_streams = kMaxStreams;      // regions rendered concurrently (PTGPU_OPT_STREAMS)
    int passes_per_launch = 0;        // 0 = automatic (PTGPU_OPT_PASSES_PER_LAUNCH)
    int waves_per_group = 0;          // 0 = default (PTGPU_OPT_WAVES_PER_GROUP)
    int shadow_cull = PTGPU_CULL_AUTO; // PTGPU_OPT_SHADOW_CULL
    int ray_bins = 1;                  // PTGPU_OPT_RAY_BINS
    int bvh = PTGPU_BVH_AUTO;          // PTGPU_OPT_BVH
    hipStream_t stream = nullptr;     // the context's stream: everything is ordered on it
    hipStream_t side[kMaxStreams - 1] = {};
    hipEvent_t fork_ev = nullptr, join_ev[kMaxStreams - 1] = {};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // scene
    float4 *d_prims = nullptr, *d_shade = nullptr, *d_lights = nullptr;
    int4 *d_segs = nullptr;
    // shadow-ray tables (device_params.h); null when the scene gains nothing from them
    float4 *d_sprims = nullptr, *d_sprims_planes = nullptr, *d_sshade = nullptr;
    unsigned long long *d_smask_safe = nullptr, *d_smask_tight = nullptr, *d_smask_enclosed = nullptr, *d_smask_ball = nullptr, *d_cmask = nullptr;
    unsigned ball_groups = 0;   // sphere groups some light tests by the ray's end point (PTGPU_CULL_BALL)
    std::vector<float4> enclosing; // (centre, radius) of the spheres PTGPU_CULL_ENCLOSED took out of a set
    int *d_enc_index = nullptr, *d_enc_slot = nullptr; // those spheres in index order / as shadow-order slots (<= 32)
    int n_enc = 0;
    float4 *d_cbounds = nullptr;
    // BVH (device_params.h): one block + the id tables in each build's numbering
    float4 *d_bvh = nullptr, *d_bvh_planes = nullptr; // (the fast build's copy: leaf triangles in plane form)
    int *d_bvh_ids_index = nullptr, *d_bvh_ids_slot = nullptr, *d_bvh_rest_index = nullptr, *d_bvh_rest_slot = nullptr;
    int n_bnodes = 0, n_bslots = 0, n_brest = 0, bvh_type = 0;
    bool bvh_auto = false;             // AUTO turns it on for this scene
    bool bvh_auto_exact = false; // a hierarchy worth walking exists, candidate sets or not
    bool sets_tight_pay = false; // the SAFE / TIGHT sets cull enough to be worth their masks (else only ENCLOSED's do)
    float4 *d_hyb_nodes = nullptr;     // hierarchy over the shadow order (hybrid)
    int n_hnodes = 0;
    uint32_t lds_fast_hyb = 0;
    uint32_t lds_bvh = 0, lds_bvh_nodes = 0; // whole block + lights; nodes + lights (when the block does not fit)
    int sg_sph = 0, sg_tri = 0, mw_sph = 0, mw_tri = 0, n_clusters = 0, n_cquads = 0, smask_levels = 0;
    unsigned convex_counts = 0;
    // LDS for the staged records per build and cull state (0 = does not fit, read from global)
    uint32_t lds_strict = 0, lds_strict_cull = 0, lds_fast = 0;
    bool have_scene = false, have_camera = false;
    ptgpu_scene_info info{};
    pt_scene scene{};
    pt_camera camera{};

    // image
    int width = 0, height = 0;
    float4 *d_colors = nullptr;   // (r, g, b, 0) per pixel (PTGPU_COLOR_STRIDE_FLOATS)
    uint32_t *d_pixels = nullptr;
    uint2 *d_seeds = nullptr;
    std::vector<float> color_staging; // host side of read_colors / write_colors (12 <-> 16 bytes per pixel)
    int *d_work = nullptr;      // work queue of the wavefront kernel: 2 + tiles words (device_params.h)
    int *static_work = nullptr; // this device's copy of the file-scope `scratch` array (ptgpu_render_static)
    int work_queue = 2;         // PTGPU_OPT_WORK_QUEUE
    // a work-queue launch was issued since the last check of its give-up word: [0] own queue, [1] static
    bool queue_in_flight[2] = {false, false};
    int compute_units = 0;
    ptgpu_launch_info last_launch{}; // what the last render call issued (ptgpu_get_launch_info)
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
    if (c->d_sprims) (void)hipFree(c->d_sprims);
    if (c->d_sprims_planes) (void)hipFree(c->d_sprims_planes);
    if (c->d_sshade) (void)hipFree(c->d_sshade);
    c->d_sshade = nullptr;
    if (c->d_cmask) (void)hipFree(c->d_cmask);
    if (c->d_cbounds) (void)hipFree(c->d_cbounds);
    c->d_cmask = nullptr;
    c->d_cbounds = nullptr;
    if (c->d_bvh) (void)hipFree(c->d_bvh);
    if (c->d_bvh_planes) (void)hipFree(c->d_bvh_planes);
    for (int **q : {&c->d_bvh_ids_index, &c->d_bvh_ids_slot, &c->d_bvh_rest_index, &c->d_bvh_rest_slot}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    c->d_bvh = c->d_bvh_planes = nullptr;
    c->n_bnodes = c->n_bslots = c->n_brest = 0;
    if (c->d_hyb_nodes) (void)hipFree(c->d_hyb_nodes);
    c->d_hyb_nodes = nullptr;
    c->n_hnodes = 0;
    if (c->d_smask_safe) (void)hipFree(c->d_smask_safe);
    if (c->d_smask_tight) (void)hipFree(c->d_smask_tight);
    if (c->d_smask_enclosed) (void)hipFree(c->d_smask_enclosed);
    if (c->d_smask_ball) (void)hipFree(c->d_smask_ball);
    if (c->d_enc_index) (void)hipFree(c->d_enc_index);
    if (c->d_enc_slot) (void)hipFree(c->d_enc_slot);
    c->d_enc_index = c->d_enc_slot = nullptr;
    c->n_enc = 0;
    c->d_sprims = nullptr;
    c->d_sprims_planes = nullptr;
    c->d_smask_safe = c->d_smask_tight = c->d_smask_enclosed = c->d_smask_ball = nullptr;
    c->d_prims = c->d_shade = c->d_lights = nullptr;
    c->d_segs = nullptr;
    c->enclosing.clear(); // what the ENCLOSED / BALL camera check walks: never the previous scene's
    c->ball_groups = 0;
    c->have_scene = false;
}

void free_image(ptgpu_ctx *c) {
    if (c->d_colors) (void)hipFree(c->d_colors);
    if (c->d_pixels) (void)hipFree(c->d_pixels);
    if (c->d_seeds) (void)hipFree(c->d_seeds);
    if (c->d_work) (void)hipFree(c->d_work);
    c->d_work = nullptr;
    c->d_colors = nullptr;
    c->d_pixels = nullptr;
    c->d_seeds = nullptr;
    c->width = c->height = 0;
}

int as_int(float f) {
    int v;
    memcpy(&v, &f, sizeof v);
    return v;
}

float as_float(int v) {
    float f;
    memcpy(&f, &v, sizeof f);
    return f;
}

#ifndef PT_QUEUE_MAX_UNITS
#define PT_QUEUE_MAX_UNITS (1ll << 30)
#endif
constexpr long long kQueueMaxUnits = PT_QUEUE_MAX_UNITS; // units of one work-queue launch (an int in the kernel)
constexpr double kEnclosedMargin = 1e-5; // "inside a sphere" = closer to its centre than r (1 - margin)
// The image a launch works on: the context's own buffers (ptgpu_resize) or, for ptgpu_render_static,
// the caller's, with the work queue in the static array.
struct ImageTarget {
    int width, height;
    float4 *colors;
    uint32_t *pixels;
    uint2 *seeds;
    int *work;
    bool is_static; // always the wavefront kernel with the work queue, whatever the options say
};
ImageTarget own_image(const ptgpu_ctx *c) {
    return ImageTarget{c->width, c->height, c->d_colors, c->d_pixels, c->d_seeds, c->d_work, false};
}

PtDeviceParams make_params(const ptgpu_ctx *c, const ImageTarget &img, int first_sample, int n_passes) {
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
    p.width = img.width;
    p.height = img.height;
    p.current_sample = first_sample;
    p.n_passes = n_passes;
    p.colors = img.colors;
    p.pixels = img.pixels;
    p.seeds = img.seeds;
    p.tiles_x = (img.width + 7) / 8;
    p.n_tiles = p.tiles_x * ((img.height + 7) / 8);
    p.tile_first = 0;
    p.tile_stride = 1;
    p.n_launch_tiles = p.n_tiles;
    int cull = c->shadow_cull;
    if (cull == PTGPU_CULL_AUTO) cull = c->math == PTGPU_MATH_STRICT ? PTGPU_CULL_SAFE : PTGPU_CULL_ENCLOSED;
    if (cull == PTGPU_CULL_ENCLOSED || cull == PTGPU_CULL_BALL || cull == PTGPU_CULL_ENCLOSED_SELF) {
        // the rule holds for paths that start inside every sphere it dropped: camera rays leave from
        // orig + 0.1 rdir (main-cpu.c:76), |rdir| < 3 for any sane field of view, so one unit of margin
        for (const float4 &s : c->enclosing) {
            const double dx = c->camera.orig.x - s.x, dy = c->camera.orig.y - s.y, dz = c->camera.orig.z - s.z;
            if (!(std::sqrt(dx * dx + dy * dy + dz * dz) <= (double)s.w * (1 - kEnclosedMargin) - 1.0)) cull = PTGPU_CULL_TIGHT;
        }
    }
    if ((cull == PTGPU_CULL_SAFE || cull == PTGPU_CULL_TIGHT) && !c->sets_tight_pay) cull = PTGPU_CULL_OFF;
    p.sprims = c->d_sprims;
    p.sshade = c->d_sshade;
    p.smask = cull == PTGPU_CULL_SAFE       ? c->d_smask_safe
              : cull == PTGPU_CULL_TIGHT    ? c->d_smask_tight
              : cull == PTGPU_CULL_ENCLOSED ? c->d_smask_enclosed
              : cull == PTGPU_CULL_ENCLOSED_SELF ? c->d_smask_enclosed
              : cull == PTGPU_CULL_BALL     ? c->d_smask_ball
                                            : nullptr;
    p.self_sphere = cull == PTGPU_CULL_ENCLOSED_SELF && p.smask ? 1 : 0;
    // the spheres the ENCLOSED family dropped: the extension rays watch their answers (radiance_device.inc, sphere_t_odd)
    const bool family = cull == PTGPU_CULL_ENCLOSED || cull == PTGPU_CULL_BALL || cull == PTGPU_CULL_ENCLOSED_SELF;
    p.n_enc = family && p.smask ? c->n_enc : 0;
    p.enc_list = c->math == PTGPU_MATH_STRICT ? c->d_enc_index : c->d_enc_slot;
    p.ball_groups = cull == PTGPU_CULL_BALL && p.smask ? c->ball_groups : 0u;
    p.n_sph_real = (int)c->info.spheres;
    p.sg_sph = c->sg_sph;
    p.sg_tri = c->sg_tri;
    p.mw_sph = c->mw_sph;
    p.mw_tri = c->mw_tri;
    p.smask_levels = p.smask ? c->smask_levels : 0;
    // ray binning on triangle clusters: fast build, TIGHT sets only (the bins are built from them)
    const bool bins = c->ray_bins && (cull == PTGPU_CULL_TIGHT || cull == PTGPU_CULL_ENCLOSED || cull == PTGPU_CULL_BALL || cull == PTGPU_CULL_ENCLOSED_SELF) && p.smask &&
                      c->math == PTGPU_MATH_FAST;
    p.n_clusters = bins ? c->n_clusters : 0;
    p.n_cquads = bins ? c->n_cquads : 0;
    p.convex_counts = bins ? c->convex_counts : 0u;
    p.cmask = c->d_cmask;
    p.cbounds = c->d_cbounds;
    // AUTO: the hierarchy where the candidate sets do not apply - and, for the strict build left to its defaults
    // under path tracing, wherever a hierarchy of some size exists: SAFE sets (what that build would scan otherwise)
    // cull little, and any answer through the hierarchy is the linear scan's bit for bit (cornell.txt strict:
    // 34 -> 65 Mpaths/s, scripts/strict_choice.py).  A caller who names a cull mode gets that mode's sets.
    const bool exact_by_tree = c->math == PTGPU_MATH_STRICT && c->shadow_cull == PTGPU_CULL_AUTO &&
                               c->integrator == PT_INTEGRATOR_PT && c->bvh_auto_exact;
    const bool use_bvh = c->n_bnodes > 0 && (c->bvh == PTGPU_BVH_ON || (c->bvh == PTGPU_BVH_AUTO && (c->bvh_auto || exact_by_tree)));
    // the fast build's own translation units read triangles in plane form (radiance_device.inc)
#if !(defined(PT_EXACT) && (PT_EXACT & 16)) // (measurement build with Moeller-Trumbore in the fast kernels: scripts/bias_bisect.sh)
    if (c->math == PTGPU_MATH_FAST) p.sprims = c->d_sprims_planes;
#endif
    if (use_bvh) { // the hierarchy replaces the candidate sets
        p.smask = nullptr;
        p.self_sphere = 0;
        p.n_enc = 0;
        p.ball_groups = 0;
        p.n_clusters = 0;
        p.n_cquads = 0;
        p.convex_counts = 0;
        p.bvh = c->d_bvh;
#if !(defined(PT_EXACT) && (PT_EXACT & 16))
        if (c->math == PTGPU_MATH_FAST) p.bvh = c->d_bvh_planes;
#endif
        const bool strict = c->math == PTGPU_MATH_STRICT;
        p.bvh_ids = strict ? c->d_bvh_ids_index : c->d_bvh_ids_slot;
        p.bvh_rest_ids = strict ? c->d_bvh_rest_index : c->d_bvh_rest_slot;
        p.n_bnodes = c->n_bnodes;
        p.coop_stragglers = PT_COOP_STRAGGLERS;
        p.n_bslots = c->n_bslots;
        p.n_brest = c->n_brest;
        p.bvh_type = c->bvh_type;
    } else if (c->bvh != PTGPU_BVH_OFF && c->math == PTGPU_MATH_FAST && p.smask && c->n_hnodes > 0) {
        p.hyb_nodes = c->d_hyb_nodes; // candidate sets stay on; extension rays take the hierarchy
        p.n_hnodes = c->n_hnodes;
    }
    // the fast build's point test lives in the hybrid instance (kernels_fast_ball.hip): without the
    // hierarchy (PTGPU_OPT_BVH OFF) BALL renders as ENCLOSED - the same sets, the sphere ray-tested
    if (c->math == PTGPU_MATH_FAST && !p.n_hnodes) p.ball_groups = 0;
    return p;
}

// ---- shadow-ray candidate sets (PTGPU_OPT_SHADOW_CULL, ptgpu.h) ---------------------------------
struct ShadowTables {
    std::vector<float4> sprims;
    std::vector<float4> sprims_planes; // the same array with the triangles in plane form (fast build)
    std::vector<int> prim_of_slot;   // shadow-order slot -> primitive index, -1 for a filler
    std::vector<int> cluster_of;     // primitive -> triangle cluster or -1
    std::vector<float4> cluster_bounds;              // (centre, padded radius) per cluster
    std::vector<float4> cluster_table;               // the bounds, then per cluster its convex planes (if any)
    unsigned convex_counts = 0;                      // planes per cluster, a nibble each
    std::vector<unsigned long long> cluster_masks;   // (1 + clusters) x mw_tri words
    std::vector<unsigned> light_clusters;            // per light: clusters present in its TIGHT set
    std::vector<float4> hyb_nodes;                   // hierarchy whose leaves are the triangle groups
    std::vector<unsigned long long> safe, tight, enclosed, ball;
    std::vector<unsigned> light_ball;                // per light: sphere groups tested by the ray's end point (BALL)
    std::vector<float4> enclosing;                   // spheres ENCLOSED dropped from at least one set
    std::vector<uint8_t> self_flag;                  // per primitive: such a sphere (its record says so in the second
                                                     // quad's .y: 1, 0 elsewhere)
    std::vector<int> enc_index, enc_slot;            // those spheres (<= 32) in index order / as shadow-order slots
    bool tables_built = false;
    int levels = 0;                                  // of the range table appended to safe / tight / enclosed / ball
    int sg_sph = 0, sg_tri = 0, mw_sph = 0, mw_tri = 0;
    double mean_safe = 0, mean_tight = 0, mean_enclosed = 0; // primitives tested per shadow ray, light-averaged
};

struct D3 { double x, y, z; };
D3 d3(const float4 &v) { return {v.x, v.y, v.z}; }
D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
D3 add(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
double len(D3 a) { return std::sqrt(dot(a, a)); }

// (p1, e1, e2) -> the plane form of a triangle record (radiance_device.inc, triangle_planes_t), in place
void to_plane_form(float4 *q) {
    const D3 p1 = d3(q[0]), e1 = d3(q[1]), e2 = d3(q[2]);
    const D3 n = cross(e1, e2);
    const double nn = dot(n, n);
    q[0] = q[1] = q[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(nn > 0.0) || !std::isfinite(nn) || !std::isfinite(dot(p1, p1))) return;
    const D3 c1 = cross(e2, n), c2 = cross(n, e1);
    const D3 n1 = {c1.x / nn, c1.y / nn, c1.z / nn}, n2 = {c2.x / nn, c2.y / nn, c2.z / nn};
    q[0] = make_float4((float)n.x, (float)n.y, (float)n.z, (float)dot(n, p1));
    q[1] = make_float4((float)n1.x, (float)n1.y, (float)n1.z, (float)-dot(n1, p1));
    q[2] = make_float4((float)n2.x, (float)n2.y, (float)n2.z, (float)-dot(n2, p1));
}

// keep[l * n_prims + i] = primitive i may occlude a shadow ray towards light l under `mode`
// (PTGPU_CULL_SAFE / TIGHT / ENCLOSED / BALL, ptgpu.h; anything else keeps everything); under BALL
// 2 = it may, and the test is whether the ray's point at maxt lies outside sphere i.
// prims: 3 quads per primitive in index order; types[i]: PT_SPHERE / PT_TRIANGLE; lights: 4 quads each;
// dielectric[i]: primitive i refracts (paths pass through it: it encloses nothing).
std::vector<uint8_t> shadow_candidates(const std::vector<float4> &prims, const std::vector<int> &types,
                                       const std::vector<float4> &lights, const std::vector<uint8_t> &dielectric,
                                       int mode) {
    const size_t n_prims = types.size(), n_lights = lights.size() / 4;
    std::vector<uint8_t> keep(n_lights * n_prims, 1);
    if (mode != PTGPU_CULL_SAFE && mode != PTGPU_CULL_TIGHT && mode != PTGPU_CULL_ENCLOSED && mode != PTGPU_CULL_BALL &&
        mode != PTGPU_CULL_ENCLOSED_SELF)
        return keep;
    const bool ball_rule = mode == PTGPU_CULL_BALL, self_rule = mode == PTGPU_CULL_ENCLOSED_SELF;
    const bool enclosed_rule = mode == PTGPU_CULL_ENCLOSED || ball_rule || self_rule;
    if (enclosed_rule) mode = PTGPU_CULL_TIGHT;
    std::vector<uint8_t> encloses(n_prims, 0), ruled(n_lights, 0); // sphere i encloses some light; light l has sets
    // the spheres the ENCLOSED rule may drop: opaque, the first 32 of them (BALL keeps a bit per such sphere, and the
    // kernels test every one of them for the rare vertex that may be outside one: a short list)
    std::vector<uint8_t> droppable(n_prims, 0);
    if (enclosed_rule) {
        int n = 0;
        for (size_t i = 0; i < n_prims && n < 32; ++i)
            if (types[i] == PT_SPHERE && !dielectric[i]) { droppable[i] = 1; ++n; }
    }
    const double eps = (double)PT_EPSILON;
    for (size_t l = 0; l < n_lights; ++l) {
        int light_type;
        memcpy(&light_type, &lights[4 * l].w, sizeof light_type);
        const D3 p0 = d3(lights[4 * l]), e1 = d3(lights[4 * l + 1]), e2 = d3(lights[4 * l + 2]);
        D3 n = cross(e1, e2);
        const double nn = len(n), size = len(e1) + len(e2);
        // one-sided triangle lights with a usable normal; every other light keeps every primitive - except, under
        // ENCLOSED, the opaque spheres a SPHERE light lies strictly inside (the wall spheres of the sphere-lit
        // rooms: cornell-large.txt, cornell-original.txt, materials.txt ...): the same rule, the same reason
        if (light_type != PT_TRIANGLE || !std::isfinite(nn) || !(nn > 1e-12 * size * size)) {
            if (enclosed_rule && light_type == PT_SPHERE) {
                const double rl = std::fabs((double)lights[4 * l + 1].x);
                for (size_t i = 0; i < n_prims; ++i) {
                    if (!droppable[i]) continue;
                    const double r = std::fabs((double)prims[3 * i].w);
                    if (len(sub(d3(prims[3 * i]), p0)) + rl <= r * (1 - kEnclosedMargin)) { // (a NaN keeps it)
                        encloses[i] = 1;
                        keep[l * n_prims + i] = 0;
                    }
                }
            }
            continue;
        }
        ruled[l] = 1;
        n = {n.x / nn, n.y / nn, n.z / nn};
        // `h` = height of a point of the primitive above the light's plane, `reach` = its distance
        // from the light.  A NaN makes the comparison false: kept.
        auto behind = [&](double h, double reach) {
            return mode == PTGPU_CULL_SAFE ? h < -(10 * eps + 1e-4 * reach) : h <= 1e-5 * (reach + size);
        };
        for (size_t i = 0; i < n_prims; ++i) {
            bool all_behind;
            if (types[i] == PT_SPHERE) {
                const D3 c = sub(d3(prims[3 * i]), p0);
                const double r = std::fabs((double)prims[3 * i].w);
                all_behind = behind(dot(c, n) + r, len(c) + r);
                // ENCLOSED: the light's three corners strictly inside an opaque sphere (a NaN keeps it)
                if (droppable[i]) {
                    const double inside = r * (1 - kEnclosedMargin);
                    if (len(c) <= inside && len(sub(c, e1)) <= inside && len(sub(c, e2)) <= inside) {
                        if (!all_behind) encloses[i] = 1; // what make_params checks the camera against
                        all_behind = true;
                    }
                }
            } else {
                const D3 a = sub(d3(prims[3 * i]), p0);
                const D3 b = add(a, d3(prims[3 * i + 1])), c = add(a, d3(prims[3 * i + 2]));
                all_behind = behind(dot(a, n), len(a)) && behind(dot(b, n), len(b)) && behind(dot(c, n), len(c));
            }
            if (all_behind) keep[l * n_prims + i] = 0;
        }
    }
    // BALL: a shadow ray of a path that lives inside an opaque sphere (the spheres ENCLOSED relies
    // on) crosses the sphere before maxt exactly when its point at maxt is outside it: where such a
    // sphere is still in a light's set, 2 = "test that point, not the ray".
    // (The device keeps a light's point-tested spheres in a 32-bit word, and one scene-wide word of
    // spheres that leave the ray scans: scenes of up to 32 spheres whose lights all have sets.)
    size_t n_spheres = 0;
    for (size_t i = 0; i < n_prims; ++i) n_spheres += types[i] == PT_SPHERE;
    bool all_ruled = true; // a set bit of such a sphere then means the point test for every light
    for (size_t l = 0; l < n_lights; ++l) all_ruled = all_ruled && ruled[l];
    if (ball_rule && n_spheres <= 32 && all_ruled)
        for (size_t l = 0; l < n_lights; ++l)
            for (size_t i = 0; i < n_prims; ++i)
                if (ruled[l] && encloses[i] && keep[l * n_prims + i]) keep[l * n_prims + i] = 2;
    // A sphere the rule took out of some light's set is still tested, whatever the light, by the rays of a path vertex
    // that may lie outside a dropped sphere (4; ptgpu.h: upstream's paths do leave their room through its corners) -
    // under ENCLOSED_SELF also by the rays that start ON it (3)
    if (enclosed_rule)
        for (size_t l = 0; l < n_lights; ++l)
            for (size_t i = 0; i < n_prims; ++i)
                if (encloses[i] && !keep[l * n_prims + i]) keep[l * n_prims + i] = self_rule ? 3 : 4;
    return keep;
}

// Triangle clusters for the fast build's ray binning (device_params.h): connected components of
// triangles that share a vertex position, the kMaxClusters largest with at least kMinClusterTris
// triangles.  cluster_of[i] = cluster of primitive i or -1.
constexpr int kMaxClusters = 8, kMinClusterTris = 8;
std::vector<int> find_clusters(const std::vector<float4> &prims, const std::vector<int> &types,
                               std::vector<float4> &bounds) {
    const size_t n = types.size();
    std::vector<int> parent(n);
    for (size_t i = 0; i < n; ++i) parent[i] = (int)i;
    auto find = [&](int x) {
        while (parent[x] != x) x = parent[x] = parent[parent[x]];
        return x;
    };
    struct Key {
        uint32_t x, y, z;
        bool operator<(const Key &o) const { return x != o.x ? x < o.x : y != o.y ? y < o.y : z < o.z; }
    };
    std::map<Key, int> first_user;
    auto corners = [&](size_t i, float4 v[3]) {
        const float4 a = prims[3 * i], e1 = prims[3 * i + 1], e2 = prims[3 * i + 2];
        v[0] = a;
        v[1] = make_float4(a.x + e1.x, a.y + e1.y, a.z + e1.z, 0.f); // = p2, p3 up to one rounding: enough to
        v[2] = make_float4(a.x + e2.x, a.y + e2.y, a.z + e2.z, 0.f); // bound; sharing is tested on p1 + e below
    };
    for (size_t i = 0; i < n; ++i) {
        if (types[i] != PT_TRIANGLE) continue;
        float4 v[3];
        corners(i, v);
        for (const float4 &c : v) {
            // quantise to ~1e-5 relative so that p1 + (p2 - p1) still meets the neighbour's p2
            auto q = [](float f) { return (uint32_t)(as_int(f) + 0x40) & ~0x7fu; };
            const Key k{q(c.x), q(c.y), q(c.z)};
            auto it = first_user.find(k);
            if (it == first_user.end()) first_user.emplace(k, (int)i);
            else parent[find((int)i)] = find(it->second);
        }
    }
    std::map<int, int> size_of;
    for (size_t i = 0; i < n; ++i)
        if (types[i] == PT_TRIANGLE) size_of[find((int)i)]++;
    std::vector<std::pair<int, int>> big; // (-size, root)
    for (auto &kv : size_of)
        if (kv.second >= kMinClusterTris) big.push_back({-kv.second, kv.first});
    std::sort(big.begin(), big.end());
    if ((int)big.size() > kMaxClusters) big.resize(kMaxClusters);
    std::map<int, int> id_of;
    for (size_t k = 0; k < big.size(); ++k) id_of[big[k].second] = (int)k;
    std::vector<int> cluster_of(n, -1);
    std::vector<D3> lo(big.size(), D3{1e300, 1e300, 1e300}), hi(big.size(), D3{-1e300, -1e300, -1e300});
    for (size_t i = 0; i < n; ++i) {
        if (types[i] != PT_TRIANGLE) continue;
        auto it = id_of.find(find((int)i));
        if (it == id_of.end()) continue;
        const int c = cluster_of[i] = it->second;
        float4 v[3];
        corners(i, v);
        for (const float4 &p : v) {
            lo[c] = {std::min(lo[c].x, (double)p.x), std::min(lo[c].y, (double)p.y), std::min(lo[c].z, (double)p.z)};
            hi[c] = {std::max(hi[c].x, (double)p.x), std::max(hi[c].y, (double)p.y), std::max(hi[c].z, (double)p.z)};
        }
    }
    bounds.clear();
    for (size_t c = 0; c < big.size(); ++c) {
        const D3 mid{(lo[c].x + hi[c].x) / 2, (lo[c].y + hi[c].y) / 2, (lo[c].z + hi[c].z) / 2};
        double r = 0;
        for (size_t i = 0; i < n; ++i) {
            if (cluster_of[i] != (int)c) continue;
            float4 v[3];
            corners(i, v);
            for (const float4 &p : v) r = std::max(r, len(sub(d3(p), mid)));
        }
        // padded far beyond any rounding of the triangle test or of the segment test below; a NaN
        // or infinite bound makes the device test pass for every ray (comparisons written that way)
        r = r * 1.001 + 10 * (double)PT_EPSILON;
        bounds.push_back(make_float4((float)mid.x, (float)mid.y, (float)mid.z, (float)r));
    }
    return cluster_of;
}

// A cluster that is a CLOSED CONVEX mesh (cornell.txt's cube) need not be tested triangle by triangle by a shadow
// ray: the ray is occluded by it exactly when the segment crosses its boundary between EPSILON and maxt, which the
// planes of its faces decide (device: segment_crosses_convex).  Returns the outward planes (unit normal, offset;
// coplanar faces merged) or nothing when the cluster is not such a mesh or has more than kMaxConvexPlanes faces.
constexpr int kMaxConvexPlanes = 15; // a nibble per cluster (8 clusters: one 32-bit kernel argument)
std::vector<float4> convex_planes(const std::vector<float4> &prims, const std::vector<int> &cluster_of, int cluster) {
    std::vector<D3> verts;
    std::vector<std::array<D3, 3>> tris;
    for (size_t i = 0; i < cluster_of.size(); ++i) {
        if (cluster_of[i] != cluster) continue;
        const D3 a = d3(prims[3 * i]), b = add(a, d3(prims[3 * i + 1])), c = add(a, d3(prims[3 * i + 2]));
        tris.push_back({a, b, c});
        verts.push_back(a); verts.push_back(b); verts.push_back(c);
    }
    if (tris.size() < 4) return {};
    D3 mid{0, 0, 0};
    double scale = 0;
    for (const D3 &v : verts) mid = add(mid, v);
    mid = {mid.x / verts.size(), mid.y / verts.size(), mid.z / verts.size()};
    for (const D3 &v : verts) scale = std::max(scale, len(sub(v, mid)));
    if (!(scale > 0) || !std::isfinite(scale)) return {};
    const double tol = 1e-5 * scale;
    // closed: every edge (by quantised end points) is used by exactly two triangles
    auto key = [&](const D3 &v) {
        return std::array<long long, 3>{llround((v.x - mid.x) / tol / 4), llround((v.y - mid.y) / tol / 4), llround((v.z - mid.z) / tol / 4)};
    };
    std::map<std::pair<std::array<long long, 3>, std::array<long long, 3>>, int> edges;
    for (const auto &t : tris)
        for (int e = 0; e < 3; ++e) {
            auto a = key(t[e]), b = key(t[(e + 1) % 3]);
            if (a == b) return {}; // degenerate triangle
            if (b < a) std::swap(a, b);
            edges[{a, b}]++;
        }
    for (const auto &kv : edges)
        if (kv.second != 2) return {};
    // convex: every vertex on or behind every face's plane, the plane oriented away from the centroid
    std::vector<float4> planes;
    std::vector<D3> normals;
    std::vector<double> offsets;
    for (const auto &t : tris) {
        D3 n = cross(sub(t[1], t[0]), sub(t[2], t[0]));
        const double nn = len(n);
        if (!(nn > 1e-12 * scale * scale)) return {};
        n = {n.x / nn, n.y / nn, n.z / nn};
        double w = dot(n, t[0]);
        if (dot(n, mid) > w) { n = {-n.x, -n.y, -n.z}; w = -w; }
        // the centroid strictly inside: a flat, double-sided polygon (closed by the edge count, "convex" by the vertex
        // test) has its centroid IN every face plane, the flip above is then decided by rounding, and what would come
        // out bounds an infinite slab - a false occluder far outside the polygon
        if (!(dot(n, mid) < w - tol)) return {};
        for (const D3 &v : verts)
            if (dot(n, v) > w + tol) return {};
        bool seen = false;
        for (size_t k = 0; k < normals.size() && !seen; ++k)
            seen = len(sub(normals[k], n)) < 1e-6 && std::fabs(offsets[k] - w) < tol;
        if (seen) continue;
        normals.push_back(n);
        offsets.push_back(w);
        planes.push_back(make_float4((float)n.x, (float)n.y, (float)n.z, (float)w));
        if ((int)planes.size() > kMaxConvexPlanes) return {};
    }
    if (planes.size() < 4) return {}; // fewer planes bound no volume
    return planes;
}

void build_shadow_tables(const std::vector<float4> &prims, const std::vector<int> &types,
                         const std::vector<float4> &lights, const std::vector<uint8_t> &dielectric, ShadowTables &t) {
    t.cluster_of = find_clusters(prims, types, t.cluster_bounds);
    t.self_flag.assign(types.size(), 0);
    const int n_clusters = (int)t.cluster_bounds.size();
    // cluster table (device_params.h, cbounds): [bounds][planes of the convex clusters, cluster by cluster]; the plane
    // counts travel as a kernel argument, a nibble per cluster (LDS is short: cornell.txt's workgroup fits three to
    // a CU with 112 bytes to spare - its six cube planes are 96)
    {
        std::vector<float4> planes;
        t.convex_counts = 0;
        for (int c = 0; c < n_clusters; ++c) {
            const std::vector<float4> pc = convex_planes(prims, t.cluster_of, c);
            t.convex_counts |= (unsigned)pc.size() << (4 * c);
            planes.insert(planes.end(), pc.begin(), pc.end());
        }
        t.cluster_table = t.cluster_bounds;
        t.cluster_table.insert(t.cluster_table.end(), planes.begin(), planes.end());
    }
    // shadow order: spheres; then the triangles of cluster 0, 1, ... each padded to whole groups;
    // then the unclustered triangles.  slots[] = primitive index or -1 for a never-hit filler.
    std::vector<int> sph_slots, tri_slots;
    auto pad = [](std::vector<int> &v, int g) { while (v.size() % g) v.push_back(-1); };
    for (size_t i = 0; i < types.size(); ++i)
        if (types[i] == PT_SPHERE) sph_slots.push_back((int)i);
    const size_t n_lights = lights.size() / 4, n_prims = types.size();
    // lights x primitives decisions: beyond 2^26 of them the tables are not built and every shadow
    // ray scans everything, as upstream
    t.tables_built = (double)n_lights * (double)n_prims <= (double)(1u << 26);
    std::vector<uint8_t> keep_enclosed;
    if (t.tables_built) {
        // the spheres the ENCLOSED rule drops for some light come first (slot k = bit k of the per-vertex masks and of
        // BALL's words: the kernels walk slots [0, n_enc) without an index table), then the others; within each
        // class those that most lights still have to test first
        keep_enclosed = shadow_candidates(prims, types, lights, dielectric, PTGPU_CULL_ENCLOSED);
        std::vector<size_t> users(n_prims, 0);
        std::vector<uint8_t> dropped_somewhere(n_prims, 0);
        for (size_t l = 0; l < n_lights; ++l)
            for (int i : sph_slots) {
                users[i] += keep_enclosed[l * n_prims + i] == 1;
                dropped_somewhere[i] |= keep_enclosed[l * n_prims + i] >= 3;
            }
        std::stable_sort(sph_slots.begin(), sph_slots.end(), [&](int a, int b) {
            return dropped_somewhere[a] != dropped_somewhere[b] ? dropped_somewhere[a] > dropped_somewhere[b] : users[a] > users[b];
        });
    }
    pad(sph_slots, kSphereGroup);
    // Triangle order = leaf order of a hierarchy whose leaves never mix clusters (bvh_build.h,
    // keyed build): the groups of 4 below are its leaves, cluster by cluster, unclustered last.
    static_assert(kTriGroup == ptbvh::kLeaf, "a shadow group is a leaf");
    const ptbvh::Built over = ptbvh::build(prims, types, &t.cluster_of, PT_TRIANGLE);
    tri_slots = over.prim_of_slot;
    t.hyb_nodes = over.nodes;
    auto cluster_of_group = [&](int g) {
        for (int k = 0; k < kTriGroup; ++k)
            if (tri_slots[g * kTriGroup + k] >= 0) return t.cluster_of[tri_slots[g * kTriGroup + k]];
        return -1;
    };
    t.sg_sph = (int)sph_slots.size() / kSphereGroup;
    t.sg_tri = (int)tri_slots.size() / kTriGroup;
    t.mw_sph = (t.sg_sph + 63) / 64;
    t.mw_tri = (t.sg_tri + 63) / 64;
    const float ninf = -__builtin_inff();
    // never-hit fillers: a sphere with radius^2 = -inf (NaN discriminant fails every compare) and a
    // triangle with zero edges (0 * inf = NaN barycentrics and distance, "t > EPSILON" fails)
    auto put = [&](const std::vector<int> &slots, bool sphere) {
        for (int id : slots) {
            t.prim_of_slot.push_back(id);
            if (id >= 0) {
                for (int q = 0; q < 3; ++q) t.sprims.push_back(prims[3 * id + q]);
            } else {
                t.sprims.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
                t.sprims.push_back(sphere ? make_float4(ninf, 0.f, 0.f, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f));
                t.sprims.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
            }
        }
    };
    put(sph_slots, true);
    put(tri_slots, false);
    // plane form of the triangles (radiance_device.inc, triangle_planes_t), derived in double; a
    // triangle without area or with a non-finite coordinate becomes the never-hit filler
    t.sprims_planes = t.sprims;
    for (size_t slot = sph_slots.size(); slot < t.prim_of_slot.size(); ++slot) to_plane_form(&t.sprims_planes[3 * slot]);

    // cluster_masks: row 0 = triangle groups outside every cluster, row 1 + c = groups of cluster c
    t.cluster_masks.assign((size_t)(n_clusters + 1) * t.mw_tri, 0ull);
    for (int g = 0; g < t.sg_tri; ++g)
        t.cluster_masks[(size_t)(1 + cluster_of_group(g)) * t.mw_tri + g / 64] |= 1ull << (g % 64);

    const int words = t.mw_sph + t.mw_tri;
    auto in_set = [](uint8_t k) { return k == 1 || k == 2; }; // (3, 4: only for the rays of a vertex not well inside)
    auto masks = [&](int mode, std::vector<unsigned long long> &out, double &mean) {
        const std::vector<uint8_t> keep = mode == PTGPU_CULL_ENCLOSED
                                              ? keep_enclosed
                                              : shadow_candidates(prims, types, lights, dielectric, mode);
        out.assign(n_lights * words, 0ull);
        if (mode == PTGPU_CULL_BALL) t.light_ball.assign(n_lights, 0u);
        double total = 0;
        for (size_t l = 0; l < n_lights; ++l) {
            for (size_t k = 0; k < sph_slots.size(); ++k)
                if (sph_slots[k] >= 0 && in_set(keep[l * n_prims + sph_slots[k]])) {
                    out[l * words + (k / kSphereGroup) / 64] |= 1ull << ((k / kSphereGroup) % 64);
                    if (keep[l * n_prims + sph_slots[k]] == 2) // at most 32 spheres (shadow_candidates)
                        t.light_ball[l] |= 1u << (k / kSphereGroup);
                }
            for (size_t k = 0; k < tri_slots.size(); ++k)
                if (tri_slots[k] >= 0 && in_set(keep[l * n_prims + tri_slots[k]]))
                    out[l * words + t.mw_sph + (k / kTriGroup) / 64] |= 1ull << ((k / kTriGroup) % 64);
            for (int w = 0; w < words; ++w)
                total += (w < t.mw_sph ? kSphereGroup : kTriGroup) * __builtin_popcountll(out[l * words + w]);
        }
        mean = n_lights ? total / n_lights : 0;
    };
    // ... nor beyond 8 MiB of masks
    t.tables_built = t.tables_built && (double)n_lights * words * sizeof(unsigned long long) <= (double)(8u << 20);
    if (!t.tables_built) {
        t.mean_safe = t.mean_tight = t.mean_enclosed = (double)n_prims;
        t.light_clusters.assign(n_lights, 0u);
        t.light_ball.assign(n_lights, 0u);
        return;
    }
    masks(PTGPU_CULL_SAFE, t.safe, t.mean_safe);
    masks(PTGPU_CULL_TIGHT, t.tight, t.mean_tight);
    masks(PTGPU_CULL_ENCLOSED, t.enclosed, t.mean_enclosed);
    double mean_ball;
    masks(PTGPU_CULL_BALL, t.ball, mean_ball);
    // Range table (device_params.h, smask_levels): a batch of shadow rays aims at a RANGE of lights and scans
    // the union of their sets; level k holds the union over 2^k consecutive lights, so any range is the OR
    // of two entries - two independent loads instead of a dependent load per light.  Level 0 is the table itself.
    t.levels = 0;
    if (words == 2 && n_lights >= 2 && n_lights <= (1u << 16)) {
        int levels = 1;
        while ((2u << (levels - 1)) <= n_lights) ++levels; // 2^(levels-1) <= n_lights
        for (std::vector<unsigned long long> *m : {&t.safe, &t.tight, &t.enclosed, &t.ball}) {
            m->resize((size_t)levels * n_lights * words, 0ull);
            for (int k = 1; k < levels; ++k)
                for (size_t i = 0; i + (1u << k) <= n_lights; ++i)
                    for (int w = 0; w < words; ++w)
                        (*m)[((size_t)k * n_lights + i) * words + w] =
                            (*m)[((size_t)(k - 1) * n_lights + i) * words + w] |
                            (*m)[((size_t)(k - 1) * n_lights + i + (1u << (k - 1))) * words + w];
        }
        t.levels = levels;
    }
    {
        const std::vector<uint8_t> tight = shadow_candidates(prims, types, lights, dielectric, PTGPU_CULL_TIGHT);
        for (size_t i = 0; i < n_prims; ++i) {
            bool dropped = false;
            for (size_t l = 0; l < n_lights && !dropped; ++l) dropped = tight[l * n_prims + i] && keep_enclosed[l * n_prims + i] >= 3; // (4: kept for the rays not well inside)
            if (dropped) t.enclosing.push_back(prims[3 * i]); // (centre, radius)
            t.self_flag[i] = dropped && types[i] == PT_SPHERE;
        }
        // the dropped spheres, in each build's numbering, flagged in their records (second quad, .y)
        for (size_t i = 0; i < n_prims; ++i)
            if (t.self_flag[i]) t.enc_index.push_back((int)i);
        for (size_t slot = 0; slot < t.prim_of_slot.size(); ++slot) {
            const int i = t.prim_of_slot[slot];
            if (i < 0 || !t.self_flag[i]) continue;
            t.enc_slot.push_back((int)slot);
            t.sprims[3 * slot + 1].y = t.sprims_planes[3 * slot + 1].y = 1.f;
        }
    }

    // per light: the clusters with a group in its TIGHT set (the ones a ray towards it is binned on)
    t.light_clusters.assign(n_lights, 0u);
    for (size_t l = 0; l < n_lights; ++l)
        for (int c = 0; c < n_clusters; ++c)
            for (int w = 0; w < t.mw_tri; ++w)
                if (t.tight[l * words + t.mw_sph + w] & t.cluster_masks[(size_t)(1 + c) * t.mw_tri + w])
                    t.light_clusters[l] |= 1u << c;
}

// The device-side records of ptgpu_set_scene (device_params.h); shared with ptgpu_shadow_candidates.
struct PackedScene {
    std::vector<float4> prims, lights;
    std::vector<int> types, object_of_prim;
    std::vector<uint8_t> dielectric; // per primitive: material slot 2, upstream's built-in glass (displayfunc.c:139-150)
};
void pack_geometry(const pt_object *objects, uint32_t object_count, uint32_t light_count, PackedScene &g) {
    for (uint32_t i = 0; i < object_count; ++i) {
        const pt_object &o = objects[i];
        if (o.type != PT_SPHERE && o.type != PT_TRIANGLE) continue; // MODEL placeholders and junk are skipped
        g.types.push_back((int)o.type);
        g.object_of_prim.push_back((int)i);
        g.dielectric.push_back(o.material_id == 2);
        if (o.type == PT_SPHERE) {
            g.prims.push_back(make_float4(o.center.x, o.center.y, o.center.z, o.radius));
            g.prims.push_back(make_float4(o.radius * o.radius, 0.f, 0.f, 0.f));
            g.prims.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
        } else {
            g.prims.push_back(make_float4(o.p1.x, o.p1.y, o.p1.z, 0.f));
            g.prims.push_back(make_float4(o.p2.x - o.p1.x, o.p2.y - o.p1.y, o.p2.z - o.p1.z, 0.f));
            g.prims.push_back(make_float4(o.p3.x - o.p1.x, o.p3.y - o.p1.y, o.p3.z - o.p1.z, 0.f));
        }
        if (emits(o.emission) && g.lights.size() / 4 < light_count) { // geomfunc.h:220: stop after lightCount emitters
            const size_t q = g.prims.size() - 3;
            g.lights.push_back(make_float4(g.prims[q].x, g.prims[q].y, g.prims[q].z, as_float((int)o.type)));
            g.lights.push_back(o.type == PT_SPHERE ? make_float4(o.radius, 0.f, 0.f, 0.f) : g.prims[q + 1]);
            g.lights.push_back(g.prims[q + 2]);
            g.lights.push_back(make_float4(o.emission.x, o.emission.y, o.emission.z, o.area));
        }
    }
}

int enqueue(ptgpu_ctx *c, const ImageTarget &img, int first_sample, int n_passes) {
    if (!c->have_scene) return fail(PTGPU_ERR_STATE, "render before ptgpu_set_scene");
    if (!c->have_camera) return fail(PTGPU_ERR_STATE, "render before ptgpu_set_camera");
    if (!img.colors) return fail(PTGPU_ERR_STATE, "render before ptgpu_resize");
    if (first_sample < 0 || n_passes < 0) return fail(PTGPU_ERR_ARGUMENT, "negative sample index or pass count");
    if (n_passes == 0) return PTGPU_OK;
    PtLaunchConfig cfg{};
    PtDeviceParams probe = make_params(c, img, first_sample, 1);
    cfg.waves_per_group = c->waves_per_group;
    // One region per stream when the image has enough tiles to keep every region's launch wide.
    // Cheap scenes (a pass is tens of microseconds) gain nothing from overlapping tails and lose to
    // the extra launches, so they stay on one stream with the passes looped inside the launch.
    const long long work = (long long)c->info.primitives * (1 + (long long)c->info.lights);
    const int regions = (c->n_streams > 1 && probe.n_tiles >= 2048 && work >= 128) ? c->n_streams : 1;
    // Passes done inside one launch (the image state stays in registers across them, and a launch costs what ten
    // passes of caustic.txt do): 32 on one stream, 8 with several regions in flight - measured at 1024x768
    // (scripts/passes_per_launch.py): caustic.txt 47.9 (8) -> 53.9 (32) Gpaths/s, its direct-lighting run 65.6 -> 74.4,
    // cornell-large dl 28.2 -> 29.5; four regions with 1 / 8 passes per launch: materials dl 19.9 / 31.3, complex dl
    // 13.9 / 17.7, table-and-mirror dl 12.1 / 14.7, cornell dl 3.62 / 3.72.
    const int chunk = c->passes_per_launch > 0 ? c->passes_per_launch : (regions > 1 ? 8 : 32);

    // AUTO: the shadow-ray queues pay with two or more lights and incoherent hit points; the direct
    // lighting integrator stops at the (tile-coherent) first hit, where the inline loop is ahead.
    // A hierarchy is the other case for the wavefront kernel: its rays' walks are shared out over the wave
    // (radiance_kernels.inc, coop_walk) - mega / wavefront Mpaths/s: large-object 617 / 1 766, obj-model 1 156 /
    // 1 629, the bunny scene 162 / 519, table-and-mirror 1 740 / 2 547, complex (a sphere hierarchy) 2 161 / 2 403
    // (scripts/kernel_choice.py).
    // With the direct-lighting integrator only a large triangle mesh is (mega / wavefront: the bunny scene 813 /
    // 1 973, large-object 6 981 / 8 483, but obj-model 8 321 / 8 117, table-and-mirror 10 339 / 8 756).
    const bool hierarchy = probe.n_bnodes > 0;
    const bool big_mesh = hierarchy && probe.bvh_type == PT_TRIANGLE && probe.n_bslots >= 512;
    const int kernel = img.is_static                  ? PTGPU_KERNEL_WAVEFRONT
                       : c->kernel != PTGPU_KERNEL_AUTO ? c->kernel
                       : c->integrator == PT_INTEGRATOR_PT ? (c->info.lights >= 2 || hierarchy ? PTGPU_KERNEL_WAVEFRONT : PTGPU_KERNEL_MEGA)
                                                           : (big_mesh ? PTGPU_KERNEL_WAVEFRONT : PTGPU_KERNEL_MEGA);
    // a hierarchy too large for LDS is staged nodes only, where that pays: the wavefront kernel
    // (one large workgroup per CU shares the nodes; the mega kernel's 4-wave groups would each
    // hold a copy)
    const bool nodes_only = probe.n_bnodes && c->lds_bvh == 0 && c->lds_bvh_nodes > 0 && kernel == PTGPU_KERNEL_WAVEFRONT;
    const uint32_t stage_bytes = nodes_only                  ? c->lds_bvh_nodes
                                 : probe.n_bnodes            ? c->lds_bvh
                                 : probe.n_hnodes           ? c->lds_fast_hyb
                                 : c->math == PTGPU_MATH_FAST ? c->lds_fast
                                 : probe.smask               ? c->lds_strict_cull
                                                             : c->lds_strict;
    cfg.use_lds = c->want_lds && stage_bytes > 0;
    cfg.lds_bytes = cfg.use_lds ? stage_bytes : 0;
    auto launch = [&](const PtDeviceParams &p, hipStream_t s) -> int {
        const bool strict = c->math == PTGPU_MATH_STRICT;
        if (p.n_bnodes) { // the BVH builds of the same kernels
            if (kernel == PTGPU_KERNEL_MEGA)
                return strict ? ptk_strict_bvh::launch_mega(p, cfg, s) : ptk_fast_bvh::launch_mega(p, cfg, s);
            return strict ? ptk_strict_bvh::launch_wavefront(p, cfg, s) : ptk_fast_bvh::launch_wavefront(p, cfg, s);
        }
        if (p.ball_groups) { // PTGPU_CULL_BALL in effect: the instances with the point test (make_params)
            if (kernel == PTGPU_KERNEL_MEGA)
                return strict ? ptk_strict_ball::launch_mega(p, cfg, s) : ptk_fast_ball::launch_mega(p, cfg, s);
            return strict ? ptk_strict_ball::launch_wavefront(p, cfg, s) : ptk_fast_ball::launch_wavefront(p, cfg, s);
        }
        if (p.n_hnodes)
            return kernel == PTGPU_KERNEL_MEGA ? ptk_fast_hyb::launch_mega(p, cfg, s) : ptk_fast_hyb::launch_wavefront(p, cfg, s);
        if (kernel == PTGPU_KERNEL_MEGA) return strict ? ptk_strict::launch_mega(p, cfg, s) : ptk_fast::launch_mega(p, cfg, s);
        return strict ? ptk_strict::launch_wavefront(p, cfg, s) : ptk_fast::launch_wavefront(p, cfg, s);
    };

    auto record = [&](const PtDeviceParams &p, bool queued, int unit, int per_launch, int n_regions) {
        ptgpu_launch_info &li = c->last_launch;
        li = ptgpu_launch_info{};
        li.kernel = kernel;
        li.math = c->math;
        li.instance = p.n_bnodes ? PTGPU_INSTANCE_BVH : p.ball_groups ? PTGPU_INSTANCE_BALL : p.n_hnodes ? PTGPU_INSTANCE_HYBRID : PTGPU_INSTANCE_PLAIN;
        li.work_queue = queued ? 1 : 0;
        li.unit_passes = unit;
        li.passes_per_launch = per_launch;
        li.regions = n_regions;
        li.lds_bytes = cfg.lds_bytes;
        li.bvh_nodes_only = p.bvh_nodes_only;
        li.shadow_cull = !p.smask                            ? PTGPU_CULL_OFF
                         : p.smask == c->d_smask_safe        ? PTGPU_CULL_SAFE
                         : p.smask == c->d_smask_tight       ? PTGPU_CULL_TIGHT
                         : p.smask == c->d_smask_enclosed    ? (p.self_sphere ? PTGPU_CULL_ENCLOSED_SELF : PTGPU_CULL_ENCLOSED)
                                                             : PTGPU_CULL_BALL;
        li.ray_bins = p.n_clusters > 0;
    };
    cfg.compute_units = c->compute_units;
    // Work queue (PTGPU_OPT_WORK_QUEUE): one launch for the whole call.  Each unit costs a shared
    // counter increment and an agent-scope release / acquire (microseconds): AUTO takes it where a
    // unit can be made to run for a good fraction of a millisecond (cornell.txt 398 -> 468,
    // large-object 947 -> 1 314 Mpaths/s).
    // A unit is one tile x `unit_passes` consecutive passes, sized so that it outweighs its queue
    // traffic (one counter takes ~3e7 increments per second: every scene is capped at 1 920 Mpaths/s
    // with one-pass units): one pass for cornell.txt, 4 for obj-model, 16 for materials.txt
    // (6 843 -> 7 526 Mpaths/s with the queue, scripts/queue_units.py).
    // (with a hierarchy a ray costs a walk, not a scan: the primitives outside it + 16 per level stand for the scene -
    // complex.txt 3 -> 13 passes per unit: 2 583 -> 3 400 Mpaths/s, obj-model, table-and-mirror and large-object
    // unchanged at 10, 8 and 12, scripts/queue_choice.py)
    long long unit_work = work;
    if (probe.n_bnodes > 0)
        unit_work = (long long)(probe.n_brest + 16.0 * std::log2((double)probe.n_bslots + 1.0)) * (1 + (long long)c->info.lights);
    long long unit_passes = unit_work > 0 ? (4096 + unit_work - 1) / unit_work : 16;
    unit_passes = unit_passes < 1 ? 1 : unit_passes > 16 ? 16 : unit_passes;
#if PT_DEBUG_KNOBS // (measurement builds only, scripts/build_variants.sh: scripts/queue_units.py)
    if (const char *dbg = getenv("PTGPU_DEBUG_UNIT_PASSES")) unit_passes = atoi(dbg) > 0 ? atoi(dbg) : unit_passes;
#endif
    // the rule below looks at the unit the SCENE wants, not at what this call's pass count leaves of it: a cheap scene
    // rendered one pass per call (ptgpu_render_pass) must not fall back into the queue with one-pass units
    const long long unit_passes_wanted = unit_passes;
    if (unit_passes > n_passes) unit_passes = n_passes;
    // (Wavefront kernel only; AUTO: under the path-tracing integrator, where the rule above was
    // measured.  The mega kernel's scenes and the direct-lighting integrator are an order of magnitude
    // cheaper per pass and run into the counter's rate: a mega kernel with the queue ran caustic at 25
    // instead of 49 Gpaths/s and cornell dl at 1.8 instead of 3.6.)
    const bool queue = kernel == PTGPU_KERNEL_WAVEFRONT &&
                       (img.is_static || c->work_queue == 1 ||
                        (c->work_queue == 2 && c->integrator == PT_INTEGRATOR_PT && probe.n_tiles >= 2048 &&
                         (probe.n_bnodes > 0 || unit_passes_wanted < 16)));
    // (without a hierarchy, a scene whose unit would need the cap of 16 passes and more is too cheap for the queue now
    // that launches without it carry 8 passes on four region streams: materials.txt 7 126 with / 7 807 Mpaths/s
    // without, simple-spheres.txt 11 267 / 12 065; stacked-spheres.txt, open to the sky and uneven from tile to tile,
    // is the one scene that loses by it, 10 931 / 10 178 - nothing the library knows at launch tells it apart)
    if (queue) {
        // The kernel numbers its units with an int: a call of more than kQueueMaxUnits units (4K frame,
        // some 8 000 one-pass chunks) is issued as several launches, in order on the stream.
        const long long chunks_per_launch = kQueueMaxUnits / probe.n_tiles > 0 ? kQueueMaxUnits / probe.n_tiles : 1;
        const long long passes_per_launch = chunks_per_launch * unit_passes;
        for (long long done = 0; done < n_passes; done += passes_per_launch) {
            const int n = (int)(n_passes - done < passes_per_launch ? n_passes - done : passes_per_launch);
            PtDeviceParams p = make_params(c, img, first_sample + (int)done, n);
            p.bvh_nodes_only = nodes_only && cfg.use_lds;
            p.work = img.work;
            p.unit_passes = (int)unit_passes;
            // the counter and the per-tile words; the give-up word in front of them is sticky until it is read
            HIP_TRY(hipMemsetAsync(img.work + 1, 0, (1 + (size_t)p.n_tiles) * sizeof(int), c->stream));
            const int rc = launch(p, c->stream);
            if (rc != 0) return fail(PTGPU_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
            c->queue_in_flight[img.is_static ? 1 : 0] = true;
            record(p, true, (int)unit_passes, n, 1);
        }
        return PTGPU_OK;
    }
    int launch_error = 0;
    if (regions > 1) { // fork: side streams start after everything already queued on the main one
        HIP_TRY(hipEventRecord(c->fork_ev, c->stream));
        for (int r = 1; r < regions; ++r) HIP_TRY(hipStreamWaitEvent(c->side[r - 1], c->fork_ev, 0));
    }
    for (int done = 0; done < n_passes; done += chunk) {
        const int n = (n_passes - done < chunk) ? n_passes - done : chunk;
        for (int r = 0; r < regions; ++r) {
            PtDeviceParams p = make_params(c, img, first_sample + done, n);
            p.bvh_nodes_only = nodes_only && cfg.use_lds;
            p.tile_first = r;
            p.tile_stride = regions;
            p.n_launch_tiles = (p.n_tiles - r + regions - 1) / regions;
            const int rc = launch(p, r == 0 ? c->stream : c->side[r - 1]);
            if (rc != 0 && launch_error == 0) launch_error = rc;
            if (r == 0 && done == 0) record(p, false, 0, chunk, regions);
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
    c->compute_units = prop.multiProcessorCount;
    {   // this device's copy of the static work queue (ptgpu_render_static), with the device current
        const int e = static_queue_of(device_ordinal, &c->static_work);
        if (e != 0) {
            delete c;
            return fail(PTGPU_ERR_DEVICE, "hipGetSymbolAddress(scratch) failed on device %d: %s", device_ordinal,
                        e > 0 ? hipGetErrorString((hipError_t)e) : "device ordinal beyond the table");
        }
    }
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
    static_queue_release(ctx->device, ctx);
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
    case PTGPU_OPT_SHADOW_CULL:
        if (value < PTGPU_CULL_OFF || value > PTGPU_CULL_ENCLOSED_SELF)
            return fail(PTGPU_ERR_ARGUMENT, "unknown shadow cull mode %d", value);
        ctx->shadow_cull = value;
        return PTGPU_OK;
    case PTGPU_OPT_BVH:
        if (value < PTGPU_BVH_OFF || value > PTGPU_BVH_AUTO) return fail(PTGPU_ERR_ARGUMENT, "unknown BVH mode %d", value);
        ctx->bvh = value;
        return PTGPU_OK;
    case PTGPU_OPT_RAY_BINS:
        ctx->ray_bins = value ? 1 : 0;
        return PTGPU_OK;
    case PTGPU_OPT_WORK_QUEUE:
        if (value < 0 || value > 2) return fail(PTGPU_ERR_ARGUMENT, "work queue must be 0, 1 or 2 (automatic)");
        ctx->work_queue = value;
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

    PackedScene geo;
    pack_geometry(objects, object_count, light_count, geo);
    const std::vector<float4> &prims = geo.prims, &lights = geo.lights;
    const std::vector<int> &types = geo.types;
    std::vector<float4> shade;
    std::vector<int4> segs;
    ptgpu_scene_info info{};
    info.primitives = (uint32_t)types.size();
    info.lights = (uint32_t)(lights.size() / 4);
    for (int p = 0; p < (int)types.size(); ++p) {
        const pt_object &o = objects[geo.object_of_prim[p]];
        const bool light = emits(o.emission);
        if (!light && (o.material_id < 0 || (uint32_t)o.material_id >= material_count))
            return fail(PTGPU_ERR_ARGUMENT,
                        "object %d uses material %d but only %u materials were given "
                        "(the reference would read past its material buffer)",
                        geo.object_of_prim[p], (int)o.material_id, material_count);
        (o.type == PT_SPHERE ? info.spheres : info.triangles)++;
        if (segs.empty() || segs.back().x != (int)o.type)
            segs.push_back(make_int4((int)o.type, p, p + 1, 0));
        else
            segs.back().z = p + 1;

        pt_material m{};
        if (!light) m = materials[o.material_id];
        if ((int)m.type == 2) geo.dielectric[p] = 1; // whatever slot the caller's table keeps its glass in
        const int flags = ((int)m.type & 0xff) | ((light ? 1 : 0) << 8) | (((int)o.type & 0xff) << 16);
        shade.push_back(make_float4(o.emission.x, o.emission.y, o.emission.z, as_float(flags)));
        shade.push_back(make_float4(o.color.x, o.color.y, o.color.z, m.sigma));
        shade.push_back(make_float4(m.kd.x, m.kd.y, m.kd.z, m.A));
        shade.push_back(make_float4(m.B, 0.f, 0.f, 0.f));
    }
    info.segments = (uint32_t)segs.size();
    // device light records: the four geometry/emission quads + the plane quad (device_params.h)
    std::vector<float4> dlights;
    for (size_t l = 0; l < lights.size() / 4; ++l) {
        for (int q = 0; q < 4; ++q) dlights.push_back(lights[4 * l + q]);
        float4 plane = make_float4(0.f, 0.f, 0.f, -1.f);
        if (as_int(lights[4 * l].w) == PT_TRIANGLE) {
            const D3 p1 = d3(lights[4 * l]), e1 = d3(lights[4 * l + 1]), e2 = d3(lights[4 * l + 2]);
            const D3 n = cross(e1, e2);
            const double nn = len(n), size = len(e1) + len(e2);
            if (std::isfinite(nn) && nn > 1e-12 * size * size) {
                const float nx = (float)(n.x / nn), ny = (float)(n.y / nn), nz = (float)(n.z / nn);
                const double tol = 1e-4 * (len(p1) + size + 1.0);
                plane = make_float4(nx, ny, nz, (float)(dot(p1, {nx, ny, nz}) - tol));
            }
        }
        dlights.push_back(plane);
    }

    // Shadow order (always: the fast build traces in it) and the per-light candidate sets, which
    // are kept only when they pay (TIGHT removes at least a tenth of the tests, light-averaged)
    // and stay small.
    ShadowTables st;
    build_shadow_tables(prims, types, lights, geo.dielectric, st);
    for (size_t k = 0; k < st.enc_slot.size(); ++k) // (the fast build's kernels walk slots [0, n_enc): build_shadow_tables sorts them first)
        if (st.enc_slot[k] != (int)k) return fail(PTGPU_ERR_STATE, "internal: the dropped spheres are not the first slots of the shadow order");
    for (size_t i = 0; i < st.self_flag.size(); ++i) // (index-order records: the strict build's primary array)
        if (st.self_flag[i]) geo.prims[3 * i + 1].y = 1.f;
    if (info.lights >= (1u << 18)) { // the near-queue tag holds 18 bits of light index
        st.cluster_bounds.clear();
        st.cluster_table.clear();
    }
    for (size_t l = 0; l < st.light_clusters.size(); ++l)   // l1.w is free in both light kinds
        dlights[kLightQuads * l + 1].w = as_float(st.cluster_bounds.empty() ? 0 : (int)st.light_clusters[l]);
    unsigned ball_groups = 0;
    for (size_t l = 0; l < st.light_ball.size(); ++l) { // l2.w is free in both light kinds
        dlights[kLightQuads * l + 2].w = as_float((int)st.light_ball[l]);
        ball_groups |= st.light_ball[l];
    }
    std::vector<float4> sshade;
    for (int slot : st.prim_of_slot)
        for (int q = 0; q < 4; ++q) sshade.push_back(slot >= 0 ? shade[4 * slot + q] : make_float4(0.f, 0.f, 0.f, 0.f));
    // the tables are used when a mode culls a tenth of the primitives or more: SAFE / TIGHT by the TIGHT figure (the
    // gate of round 1), ENCLOSED also by its own (the sphere-lit rooms, where only ENCLOSED culls: make_params)
    const bool tight_pays = st.mean_tight <= 0.9 * info.primitives;
    const bool use_tables = st.tables_built && info.lights > 0 && (tight_pays || st.mean_enclosed <= 0.9 * info.primitives);
    info.shadow_groups = (uint32_t)(st.sg_sph + st.sg_tri);
    info.shadow_tests_safe = use_tables ? (float)st.mean_safe : (float)info.primitives;
    info.shadow_tests_tight = use_tables ? (float)st.mean_tight : (float)info.primitives;
    info.shadow_tests_enclosed = use_tables ? (float)st.mean_enclosed : (float)info.primitives;
    auto fits = [](size_t bytes) { return bytes <= kMaxStagedLdsBytes ? (uint32_t)bytes : 0u; };
    const size_t light_bytes = dlights.size() * sizeof(float4), shadow_bytes = st.sprims.size() * sizeof(float4);
    const uint32_t lds_strict = fits(prims.size() * sizeof(float4) + light_bytes);
    const uint32_t lds_strict_cull = fits(prims.size() * sizeof(float4) + light_bytes + shadow_bytes);
    const uint32_t lds_fast = fits(shadow_bytes + light_bytes + st.cluster_table.size() * sizeof(float4));
    const uint32_t lds_fast_hyb = fits(shadow_bytes + light_bytes + (st.cluster_table.size() + st.hyb_nodes.size()) * sizeof(float4));
    info.lds_bytes = lds_fast;

    // BVH: built for every scene with something to gain; AUTO uses it when the candidate sets do
    // not apply and the hierarchy's primitive type has at least 64 members
    ptbvh::Built bvh;
    std::vector<float4> bvh_block;
    std::vector<int> ids_index, ids_slot, rest_index, rest_slot;
    if (info.primitives >= 8 && info.primitives < (1u << 24)) {
        bvh = ptbvh::build(prims, types);
        {   // the other way to bin (bvh_build.h): kept if its tree is the cheaper one by the surface-area measure
            // (obj-model.txt: 12.41 against 12.95, the first stays; its bunny variant 14.63 / 14.57: 525 -> 650 Mpaths/s)
            ptbvh::Built other = ptbvh::build(prims, types, nullptr, -1, true);
#if PT_DEBUG_KNOBS
            if (getenv("PTGPU_DEBUG_BVH"))
                fprintf(stderr, "ptgpu: hierarchy cost %.3f (bins along the widest axis) / %.3f (along all three)\n", bvh.sah, other.sah);
#endif
            if (other.sah < bvh.sah) bvh = std::move(other);
        }
        // the cooperative walks pack (node | ray << 24) and (leaf slot | ray << 24) into one LDS word
        // (radiance_kernels.inc, coop_walk): a hierarchy beyond 2^24 nodes or slots is not used at all
        if (bvh.nodes.size() / 2 >= (1u << 24) || bvh.prim_of_slot.size() >= (1u << 24)) bvh = ptbvh::Built{};
        std::vector<int> slot_of_prim(types.size(), -1);
        for (size_t k = 0; k < st.prim_of_slot.size(); ++k)
            if (st.prim_of_slot[k] >= 0) slot_of_prim[st.prim_of_slot[k]] = (int)k;
        bvh_block = bvh.nodes;
        bvh_block.insert(bvh_block.end(), bvh.prims.begin(), bvh.prims.end());
        for (int id : bvh.prim_of_slot) {
            ids_index.push_back(id >= 0 ? id : 0x7fffffff);
            ids_slot.push_back(id >= 0 ? slot_of_prim[id] : 0x7fffffff);
        }
        for (int id : bvh.rest) {
            bvh_block.push_back(prims[3 * id]);
            bvh_block.push_back(prims[3 * id + 1]);
            float4 c = prims[3 * id + 2];
            c.w = as_float(types[id]);
            bvh_block.push_back(c);
            rest_index.push_back(id);
            rest_slot.push_back(slot_of_prim[id]);
        }
    }
    const size_t in_tree = bvh.prim_of_slot.size() ? types.size() - bvh.rest.size() : 0;
    const uint32_t lds_bvh = fits(bvh_block.size() * sizeof(float4) + light_bytes);
    const uint32_t lds_bvh_nodes = fits(bvh.nodes.size() * sizeof(float4) + light_bytes);
    info.bvh_nodes = (uint32_t)(bvh.nodes.size() / 2);
    info.bvh_primitives = (uint32_t)in_tree;

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
    HIP_TRY(upload(ctx->d_lights, dlights));
    HIP_TRY(upload(ctx->d_segs, segs));
    HIP_TRY(upload(ctx->d_sprims, st.sprims));
    HIP_TRY(upload(ctx->d_sprims_planes, st.sprims_planes));
    HIP_TRY(upload(ctx->d_sshade, sshade));
    if (use_tables) {
        ctx->ball_groups = ball_groups;
        HIP_TRY(upload(ctx->d_smask_safe, st.safe));
        HIP_TRY(upload(ctx->d_smask_tight, st.tight));
        HIP_TRY(upload(ctx->d_smask_enclosed, st.enclosed));
        HIP_TRY(upload(ctx->d_smask_ball, st.ball));
        ctx->enclosing = st.enclosing;
        HIP_TRY(upload(ctx->d_enc_index, st.enc_index));
        HIP_TRY(upload(ctx->d_enc_slot, st.enc_slot));

        ctx->n_enc = (int)st.enc_index.size();
        HIP_TRY(upload(ctx->d_cmask, st.cluster_masks));
        HIP_TRY(upload(ctx->d_cbounds, st.cluster_table));
        ctx->n_cquads = (int)st.cluster_table.size();
        ctx->convex_counts = st.convex_counts;
    }
    if (!bvh.nodes.empty()) {
        HIP_TRY(upload(ctx->d_bvh, bvh_block));
        // the fast build reads the triangles of the leaves in plane form, as it does everywhere else (the few
        // primitives outside the hierarchy keep their records: rest_prim_t)
        if (bvh.type == PT_TRIANGLE)
            for (size_t slot = 0; slot < bvh.prim_of_slot.size(); ++slot) to_plane_form(&bvh_block[bvh.nodes.size() + 3 * slot]);
        HIP_TRY(upload(ctx->d_bvh_planes, bvh_block));
        HIP_TRY(upload(ctx->d_bvh_ids_index, ids_index));
        HIP_TRY(upload(ctx->d_bvh_ids_slot, ids_slot));
        HIP_TRY(upload(ctx->d_bvh_rest_index, rest_index));
        HIP_TRY(upload(ctx->d_bvh_rest_slot, rest_slot));
        ctx->n_bnodes = (int)(bvh.nodes.size() / 2);
        ctx->n_bslots = (int)bvh.prim_of_slot.size();
        ctx->n_brest = (int)bvh.rest.size();
        ctx->bvh_type = bvh.type;
    }
    ctx->bvh_auto = !use_tables && in_tree >= 64 && bvh.rest.size() <= 64;
    ctx->bvh_auto_exact = in_tree >= 64 && bvh.rest.size() <= 64;
    if (use_tables && st.hyb_nodes.size() / 2 >= 8 && info.triangles >= 32) {
        HIP_TRY(upload(ctx->d_hyb_nodes, st.hyb_nodes));
        ctx->n_hnodes = (int)(st.hyb_nodes.size() / 2);
    }
    ctx->lds_fast_hyb = lds_fast_hyb;
    ctx->lds_bvh = lds_bvh;
    ctx->lds_bvh_nodes = lds_bvh_nodes;
    ctx->n_clusters = use_tables ? (int)st.cluster_bounds.size() : 0;
    if (!use_tables) ctx->n_cquads = 0;
    info.clusters = (uint32_t)ctx->n_clusters;
    for (int c = 0; c < ctx->n_clusters; ++c) info.convex_clusters += ((ctx->convex_counts >> (4 * c)) & 15u) ? 1u : 0u;
    for (int c : st.cluster_of) info.clustered_triangles += (ctx->n_clusters && c >= 0) ? 1u : 0u;
    ctx->sg_sph = st.sg_sph;
    ctx->sg_tri = st.sg_tri;
    ctx->mw_sph = st.mw_sph;
    ctx->mw_tri = st.mw_tri;
    ctx->smask_levels = use_tables ? st.levels : 0;
    ctx->sets_tight_pay = use_tables && tight_pays;
    ctx->lds_strict = lds_strict;
    ctx->lds_strict_cull = lds_strict_cull;
    ctx->lds_fast = lds_fast;
    ctx->info = info;
    ctx->scene = *scene;
    ctx->have_scene = true;
    return PTGPU_OK;
}

int ptgpu_shadow_candidates(const pt_object *objects, uint32_t object_count, uint32_t light_count, int mode,
                            uint8_t *keep, uint32_t *n_lights) {
    if (!objects || !n_lights) return fail(PTGPU_ERR_ARGUMENT, "ptgpu_shadow_candidates: null argument");
    if (mode != PTGPU_CULL_OFF && mode != PTGPU_CULL_SAFE && mode != PTGPU_CULL_TIGHT && mode != PTGPU_CULL_ENCLOSED &&
        mode != PTGPU_CULL_BALL && mode != PTGPU_CULL_ENCLOSED_SELF)
        return fail(PTGPU_ERR_ARGUMENT, "ptgpu_shadow_candidates: mode must be OFF, SAFE, TIGHT, ENCLOSED, BALL or ENCLOSED_SELF");
    PackedScene geo;
    pack_geometry(objects, object_count, light_count, geo);
    *n_lights = (uint32_t)(geo.lights.size() / 4);
    if (!keep) return PTGPU_OK;
    if ((double)*n_lights * (double)geo.types.size() > (double)(1u << 26))
        return fail(PTGPU_ERR_ARGUMENT, "ptgpu_shadow_candidates: %u lights x %zu primitives is beyond the size the "
                                        "tables are built for (set_scene scans everything for such a scene)",
                    *n_lights, geo.types.size());
    const std::vector<uint8_t> by_prim = shadow_candidates(geo.prims, geo.types, geo.lights, geo.dielectric, mode);
    const size_t n_prims = geo.types.size();
    memset(keep, 0, (size_t)*n_lights * object_count);
    for (size_t l = 0; l < *n_lights; ++l)
        for (size_t i = 0; i < n_prims; ++i)
            keep[l * object_count + geo.object_of_prim[i]] = by_prim[l * n_prims + i];
    return PTGPU_OK;
}

int ptgpu_cluster_info(const pt_object *objects, uint32_t object_count, uint32_t light_count, uint32_t *clusters,
                       uint32_t *convex_planes_per_cluster) {
    if (!objects || !clusters || !convex_planes_per_cluster)
        return fail(PTGPU_ERR_ARGUMENT, "ptgpu_cluster_info: null argument");
    PackedScene geo;
    pack_geometry(objects, object_count, light_count, geo);
    std::vector<float4> bounds;
    const std::vector<int> cluster_of = find_clusters(geo.prims, geo.types, bounds);
    *clusters = (uint32_t)bounds.size();
    for (int c = 0; c < kMaxClusters; ++c)
        convex_planes_per_cluster[c] = c < (int)bounds.size() ? (uint32_t)convex_planes(geo.prims, cluster_of, c).size() : 0u;
    return PTGPU_OK;
}

int ptgpu_get_launch_info(ptgpu_ctx *ctx, ptgpu_launch_info *info) {
    if (!ctx || !info) return fail(PTGPU_ERR_ARGUMENT, "null argument");
    *info = ctx->last_launch;
    return PTGPU_OK;
}

int ptgpu_probe_rays(ptgpu_ctx *ctx, int n, const float *origins, const float *directions, const float *maxt,
                     float *t, float *rgb, int *occluded, int cooperative) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!origins || !directions || !maxt || !t || !rgb || !occluded)))
        return fail(PTGPU_ERR_ARGUMENT, "null argument");
#if defined(PT_RAY_PROBE) && PT_RAY_PROBE
    if (!ctx->d_prims) return fail(PTGPU_ERR_STATE, "no scene set");
    if (n == 0) return PTGPU_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ImageTarget img = own_image(ctx); // (the launch parameters carry an image; the probe kernel does not touch it)
    const PtDeviceParams p = make_params(ctx, img, 0, 1);
    float *d_in = nullptr, *d_out = nullptr;
    const size_t in_floats = 7 * (size_t)n, out_floats = 5 * (size_t)n;
    HIP_TRY(hipMalloc((void **)&d_in, in_floats * sizeof(float)));
    hipError_t e = hipMalloc((void **)&d_out, out_floats * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_in, origins, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_in + 3 * (size_t)n, directions, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_in + 6 * (size_t)n, maxt, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    int launched = -1;
    if (e == hipSuccess) {
        PtRayProbe r;
        r.n = n;
        r.origin = d_in;
        r.direction = d_in + 3 * (size_t)n;
        r.maxt = d_in + 6 * (size_t)n;
        r.t = d_out;
        r.rgb = d_out + (size_t)n;
        r.occluded = (int *)(d_out + 4 * (size_t)n);
        r.cooperative = cooperative == 2 ? (p.n_clusters && p.convex_counts ? 2 : 0) : (cooperative && p.n_bnodes);
        const bool strict = ctx->math == PTGPU_MATH_STRICT;
        launched = p.n_bnodes      ? (strict ? ptk_strict_bvh::launch_probe(p, r, ctx->stream) : ptk_fast_bvh::launch_probe(p, r, ctx->stream))
                   : p.ball_groups ? (strict ? ptk_strict_ball::launch_probe(p, r, ctx->stream) : ptk_fast_ball::launch_probe(p, r, ctx->stream))
                   : p.n_hnodes    ? ptk_fast_hyb::launch_probe(p, r, ctx->stream)
                                   : (strict ? ptk_strict::launch_probe(p, r, ctx->stream) : ptk_fast::launch_probe(p, r, ctx->stream));
        if (launched == 0) e = hipStreamSynchronize(ctx->stream);
        if (launched == 0 && e == hipSuccess) e = hipMemcpy(t, d_out, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
        if (launched == 0 && e == hipSuccess) e = hipMemcpy(rgb, d_out + (size_t)n, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
        if (launched == 0 && e == hipSuccess) e = hipMemcpy(occluded, d_out + 4 * (size_t)n, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess || launched != 0) return fail(PTGPU_ERR_DEVICE, "the ray probe failed");
    return PTGPU_OK;
#else
    (void)cooperative;
    return fail(PTGPU_ERR_STATE, "this libptgpu.so is not the ray-probe build (scripts/build_variants.sh builds "
                                 "path-tracer_amd/variants/libptgpu_probe.so with -DPT_RAY_PROBE=1)");
#endif
}

int ptgpu_work_counters(ptgpu_ctx *ctx, unsigned long long *counters, int reset) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!counters) return fail(PTGPU_ERR_ARGUMENT, "null argument");
#if PT_COUNT_WORK
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < PTGPU_WORK_SLOTS; ++i) counters[i] = 0;
    int bad = 0;
    bad |= ptk_fast::work_counters(counters, reset);
    bad |= ptk_strict::work_counters(counters, reset);
    bad |= ptk_fast_bvh::work_counters(counters, reset);
    bad |= ptk_strict_bvh::work_counters(counters, reset);
    bad |= ptk_fast_hyb::work_counters(counters, reset);
    bad |= ptk_fast_ball::work_counters(counters, reset);
    bad |= ptk_strict_ball::work_counters(counters, reset);
    if (bad) return fail(PTGPU_ERR_DEVICE, "reading the work counters failed");
    return PTGPU_OK;
#else
    (void)reset;
    return fail(PTGPU_ERR_STATE, "this libptgpu.so is not the counting build (scripts/build_variants.sh builds "
                                 "path-tracer_amd/variants/libptgpu_count.so with -DPT_COUNT_WORK=1)");
#endif
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
    HIP_TRY(hipMalloc((void **)&ctx->d_colors, n * sizeof(float4)));
    HIP_TRY(hipMalloc((void **)&ctx->d_pixels, n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&ctx->d_seeds, n * 2 * sizeof(uint32_t)));
    const size_t work_words = 2 + (size_t)((width + 7) / 8) * ((height + 7) / 8);
    HIP_TRY(hipMalloc((void **)&ctx->d_work, work_words * sizeof(int)));
    HIP_TRY(hipMemset(ctx->d_work, 0, work_words * sizeof(int)));
    ctx->queue_in_flight[0] = false;
    HIP_TRY(hipMemset(ctx->d_colors, 0, n * sizeof(float4)));
    HIP_TRY(hipMemset(ctx->d_pixels, 0, n * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(ctx->d_seeds, seeds, n * 2 * sizeof(uint32_t), hipMemcpyHostToDevice));
    ctx->width = width;
    ctx->height = height;
    return PTGPU_OK;
}

int ptgpu_render_pass(ptgpu_ctx *ctx, int current_sample) {
    int rc = bind(ctx);
    if (rc) return rc;
    return enqueue(ctx, own_image(ctx), current_sample, 1);
}

int ptgpu_render(ptgpu_ctx *ctx, int first_sample, int n_passes) {
    int rc = bind(ctx);
    if (rc) return rc;
    return enqueue(ctx, own_image(ctx), first_sample, n_passes);
}

int ptgpu_render_static(ptgpu_ctx *ctx, int width, int height, int first_sample, int n_passes,
                        void *colors_dev, uint32_t *pixels_dev, uint32_t *seeds_dev) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!colors_dev || !pixels_dev || !seeds_dev) return fail(PTGPU_ERR_ARGUMENT, "ptgpu_render_static: null buffer");
    if (width <= 0 || height <= 0) return fail(PTGPU_ERR_ARGUMENT, "ptgpu_render_static: image size %dx%d out of range", width, height);
    const long long tiles = (long long)((width + 7) / 8) * ((height + 7) / 8);
    if (tiles > PTGPU_STATIC_MAX_TILES)
        return fail(PTGPU_ERR_ARGUMENT, "ptgpu_render_static: %dx%d is %lld tiles of 8x8 pixels, beyond the %d the static work "
                                        "queue holds (use ptgpu_resize + ptgpu_render for such a frame)",
                    width, height, tiles, PTGPU_STATIC_MAX_TILES);
    if (((uintptr_t)colors_dev & 15u) || ((uintptr_t)seeds_dev & 7u) || ((uintptr_t)pixels_dev & 3u))
        return fail(PTGPU_ERR_ARGUMENT, "ptgpu_render_static: colours must be 16-byte, seeds 8-byte, pixels 4-byte aligned");
    if (!ctx->static_work) return fail(PTGPU_ERR_STATE, "ptgpu_render_static: no static work queue on this device");
    if (n_passes > 0 && !static_queue_claim(ctx->device, ctx))
        return fail(PTGPU_ERR_STATE, "ptgpu_render_static: the static work queue of device %d is in use by another context "
                                     "(call ptgpu_finish on that context first)", ctx->device);
    const ImageTarget img{width, height, (float4 *)colors_dev, pixels_dev, (uint2 *)seeds_dev, ctx->static_work, true};
    return enqueue(ctx, img, first_sample, n_passes);
}

int ptgpu_render_timed(ptgpu_ctx *ctx, int first_sample, int n_passes, float *elapsed_ms) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!elapsed_ms) return fail(PTGPU_ERR_ARGUMENT, "null elapsed_ms");
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    rc = enqueue(ctx, own_image(ctx), first_sample, n_passes);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return PTGPU_OK;
}

// after the stream has drained: did a wave of a work-queue launch give up waiting for a tile's
// previous pass (device_params.h)?  It cannot, unless a wave died; the image would be incomplete.
static int check_work_queue(ptgpu_ctx *ctx) {
    int *const queues[2] = {ctx->d_work, ctx->static_work};
    int status = PTGPU_OK;
    for (int q = 0; q < 2; ++q) {
        if (!ctx->queue_in_flight[q] || !queues[q]) continue;
        ctx->queue_in_flight[q] = false;
        if (q == 1) static_queue_release(ctx->device, ctx); // (the stream has drained: the static queue is free again)
        int gave_up = 0;
        HIP_TRY(hipMemcpy(&gave_up, queues[q], sizeof gave_up, hipMemcpyDeviceToHost));
        if (gave_up) {
            (void)hipMemset(queues[q], 0, sizeof(int));
            status = fail(PTGPU_ERR_DEVICE, "work queue: a wave waited for a tile's previous pass beyond its limit; the frame is incomplete");
        }
    }
    return status;
}

int ptgpu_finish(ptgpu_ctx *ctx) {
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return check_work_queue(ctx);
}

static int read_back(ptgpu_ctx *ctx, void *dst, const void *src, size_t bytes, const char *what) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!dst) return fail(PTGPU_ERR_ARGUMENT, "%s: null destination", what);
    if (!src) return fail(PTGPU_ERR_STATE, "%s before ptgpu_resize", what);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return check_work_queue(ctx);
}

int ptgpu_read_pixels(ptgpu_ctx *ctx, uint32_t *pixels) {
    if (!ctx) return fail(PTGPU_ERR_ARGUMENT, "null context");
    return read_back(ctx, pixels, ctx->d_pixels, (size_t)ctx->width * ctx->height * sizeof(uint32_t),
                     "ptgpu_read_pixels");
}

// The device keeps one float4 per pixel (device_params.h); the ABI's array is the reference's vec
// layout, 3 floats per pixel (path-tracing.cl:87): the spare channel is dropped / added on the host.
int ptgpu_read_colors(ptgpu_ctx *ctx, float *colors) {
    if (!ctx) return fail(PTGPU_ERR_ARGUMENT, "null context");
    if (!colors) return fail(PTGPU_ERR_ARGUMENT, "ptgpu_read_colors: null destination");
    const size_t n = (size_t)ctx->width * ctx->height;
    try {
        ctx->color_staging.resize(n * PTGPU_COLOR_STRIDE_FLOATS);
    } catch (const std::bad_alloc &) {
        return fail(PTGPU_ERR_MEMORY, "out of host memory");
    }
    const int rc = read_back(ctx, ctx->color_staging.data(), ctx->d_colors, n * sizeof(float4), "ptgpu_read_colors");
    if (rc) return rc;
    const float *src = ctx->color_staging.data();
    for (size_t i = 0; i < n; ++i) memcpy(colors + 3 * i, src + PTGPU_COLOR_STRIDE_FLOATS * i, 3 * sizeof(float));
    return PTGPU_OK;
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
    const size_t n = (size_t)ctx->width * ctx->height;
    try {
        ctx->color_staging.assign(n * PTGPU_COLOR_STRIDE_FLOATS, 0.f);
    } catch (const std::bad_alloc &) {
        return fail(PTGPU_ERR_MEMORY, "out of host memory");
    }
    float *dst = ctx->color_staging.data();
    for (size_t i = 0; i < n; ++i) memcpy(dst + PTGPU_COLOR_STRIDE_FLOATS * i, colors + 3 * i, 3 * sizeof(float));
    HIP_TRY(hipMemcpyAsync(ctx->d_colors, dst, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
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
    const PtDeviceParams p = make_params(ctx, own_image(ctx), 0, 0);
    rc = ptk_fast::launch_repack(p, scale, ctx->stream);
    if (rc != 0) return fail(PTGPU_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return PTGPU_OK;
}

} // extern "C"
