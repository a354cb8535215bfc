/*
 * ptgpu.h - C-ABI of the MI355X radiance path (libptgpu.so, HIP for gfx950).
 *
 * Drop-in boundary: these entry points stand where the OpenCL host plumbing of the reference's
 * main-gpu.c stands, between the UI's three entry points (displayfunc.c:19-21) and the
 * `radianceGPU` kernel (path-tracing.cl:49-61 / ray-tracing.cl:32-45).  Records cross the boundary
 * in the reference's own byte layout (include/pt_records.h); all host arrays stay caller-owned and
 * are copied by the call (the reference uses blocking writes), device buffers are library-owned.
 *
 * Errors: the reference prints to stderr and exit(-1)s on any OpenCL failure (e.g.
 * main-gpu.c:709-712).  Here every call returns PTGPU_OK (0) or a negative status and
 * ptgpu_last_error() holds the message; nothing is printed and nothing exits.
 *
 * Threading: like the reference (one in-order queue, GLUT idle callback) a context is driven by
 * one host thread at a time; work is issued on one HIP stream owned by the context.
 *
 * There is no CPU path behind this interface: without a HIP device every call fails.
 */
#ifndef PTGPU_H
#define PTGPU_H

#include "pt_records.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ptgpu_ctx ptgpu_ctx;

enum {
    PTGPU_OK            = 0,
    PTGPU_ERR_ARGUMENT  = -1, /* null pointer, size out of range, material index out of range ... */
    PTGPU_ERR_STATE     = -2, /* call order: render before set_scene/resize ...                    */
    PTGPU_ERR_DEVICE    = -3, /* a HIP call failed; message carries hipGetErrorString()            */
    PTGPU_ERR_MEMORY    = -4
};

/* options for ptgpu_set_option */
enum {
    PTGPU_OPT_KERNEL = 0, /* PTGPU_KERNEL_*  (default AUTO)                             */
    PTGPU_OPT_MATH   = 1, /* PTGPU_MATH_*    (default FAST)                             */
    PTGPU_OPT_LDS    = 2, /* 1 = stage primitive/light records in LDS (default), 0 = read them
                             with wave-uniform (scalar) loads from global memory        */
    PTGPU_OPT_STREAMS = 3, /* 1..4 image regions rendered on concurrent HIP streams (default 4):
                              one region's launch tail overlaps the other regions' work.  All
                              work is still ordered on the context's stream (fork/join events) */
    PTGPU_OPT_PASSES_PER_LAUNCH = 4, /* passes looped inside one kernel launch; 0 = automatic */
    PTGPU_OPT_WAVES_PER_GROUP = 5,   /* wavefront kernel: waves per workgroup, 1..16; 0 = default */
    PTGPU_OPT_SHADOW_CULL = 6,       /* PTGPU_CULL_* (default AUTO): which primitives a shadow ray
                                        towards a one-sided triangle light is tested against     */
    PTGPU_OPT_RAY_BINS = 7,          /* 1 (default) = fast build with TIGHT sets bins shadow rays on
                                        triangle clusters (wavefront kernel): a ray that comes near no cluster's
                                        bounding sphere skips the clustered triangles; the others are traced in
                                        dense batches against the clusters they come near - a cluster that is a
                                        closed convex mesh (a box) by the planes of its faces: the segment crosses
                                        its boundary between EPSILON and maxt exactly when the any-hit scan over its
                                        triangles reports a hit, up to rays grazing an edge within rounding;
                                        0 = one queue, every triangle of the sets tested                 */
    PTGPU_OPT_BVH = 8,               /* PTGPU_BVH_* (default AUTO)                                  */
    PTGPU_OPT_WORK_QUEUE = 9         /* wavefront kernel: 1 = a render call is ONE launch that fills the
                                        chip once, its waves taking units (one tile x 1..16 consecutive
                                        passes, more where a pass is cheap) from a queue in device
                                        memory (no drain between passes; a tile's passes stay in
                                        order); 0 = a launch per pass and image region
                                        (PTGPU_OPT_STREAMS); 2 (default) = 1 where a unit is heavy
                                        enough to pay for its queue traffic (>= 2048 tiles and passes
                                        per unit x primitives x (1 + lights) >= 512: cheap scenes need
                                        several passes per call; path-tracing integrator).  The mega
                                        kernel has no queue.  Same image either way.                 */
};
/* Intersect / IntersectP (geomfunc.h:126-193) scan every object.  For a scene whose cost is that
 * scan over one large mesh the kernels can walk a bounding-volume hierarchy instead (each lane its
 * own way, nodes and leaves in LDS).  A primitive is skipped only when the ray misses a padded box
 * around it or enters the box beyond the closest hit so far, and the closest hit is chosen by
 * (distance, object index): the result is the linear scan's, whatever the visiting order.
 *   AUTO = on when the per-light candidate sets do not apply to the scene and the hierarchy holds
 *          at least 64 primitives (obj-model.txt, large-object.txt, complex.txt ...);
 *   ON   = on whenever a hierarchy could be built (>= 8 primitives); the candidate sets are then off. */
enum { PTGPU_BVH_OFF = 0, PTGPU_BVH_ON = 1, PTGPU_BVH_AUTO = 2 };
enum {
    PTGPU_KERNEL_MEGA      = 0, /* one thread = one pixel = one whole path; the direct counterpart of
                                   the reference kernel's shape, kept as the measured baseline  */
    PTGPU_KERNEL_WAVEFRONT = 1, /* wave-autonomous wavefront tracer: shadow rays compacted into
                                   per-wave LDS queues and traced by dense 64-lane batches       */
    PTGPU_KERNEL_AUTO      = 2  /* WAVEFRONT for the path-tracing integrator on scenes with two or
                                   more lights (the shadow-ray queues are what it is for) or with
                                   a hierarchy over 512 or more triangles, MEGA otherwise (measured
                                   5-15 % faster on one-light scenes and for the direct-lighting
                                   integrator); same results either way                          */
};
/* Shadow rays are only cast towards the FRONT of a triangle light (geomfunc.h:276-285 rejects a
 * sample whose direction has a positive cosine with the light's normal), so a primitive that lies
 * wholly behind the light's plane cannot be the occluder IntersectP (geomfunc.h:162-193) would
 * find.  ptgpu_set_scene builds, per light, the set of primitives that are NOT wholly behind it;
 * shadow rays scan that set instead of every object.  The answer of an any-hit scan does not depend
 * on order, so the image is the brute-force one whenever no culled primitive could have been
 * reported by the reference's floating-point tests:
 *   SAFE  culls only primitives behind the plane by a margin (10 EPSILON + 1e-4 of their distance
 *         to the light) far above any rounding of those tests: results stay bit-identical.
 *   TIGHT also culls primitives that touch the plane from behind (closed half-space, tolerance
 *         1e-5 relative): exact in real arithmetic; in floating point a ray grazing the light's
 *         plane within rounding distance may see a different occluder than upstream.
 *   ENCLOSED  TIGHT, and a sphere that is not a dielectric leaves the set of every light - triangle or sphere -
 *         that lies wholly inside it (the r = 1e4 wall spheres of the upstream scenes), FOR THE RAYS THAT START
 *         INSIDE IT: a segment between two points of a ball stays in the ball, so the sphere cannot occlude such a
 *         ray.  Whether a path vertex is inside is CHECKED: upstream's paths do leave their room - a bounce ray that
 *         starts within EPSILON of a second wall passes through it (t > EPSILON, geomfunc.h:36) and lands outside,
 *         where that wall occludes every light; round 2 assumed they could not, and that assumption was the whole
 *         one-signed deviation of this mode (+1.2e-4 of cornell.txt's radiance: 0.06 % of the vertices, fully lit
 *         instead of dark).  The check costs nothing: the extension ray that FINDS a vertex has tested every
 *         sphere, and a ray that starts inside a sphere and stays in it beyond EPSILON gets that sphere's far root
 *         from SphereIntersect (geomfunc.h:10-39); a vertex found by a ray for which some dropped sphere answered
 *         anything else (near root, no hit) may be outside it, and its shadow rays test every dropped sphere, as
 *         upstream does.  What remains of the rule's effect is the sphere a vertex STANDS ON (and, within rounding of
 *         a corner, its neighbour): upstream's float discriminant can report it for grazing rays ("acne"; their
 *         cosine, and with it their weight, is ~0: 1e-6 .. 1e-5 of the radiance, no sign).  At most 32 spheres are
 *         dropped per scene.  The library also checks at every launch that the camera is inside every sphere this
 *         rule dropped, and renders with the TIGHT sets when it is not.
 *   ENCLOSED_SELF  ENCLOSED, and a shadow ray is also tested against the dropped sphere its origin stands on
 *         (upstream's SphereIntersect on upstream's operands, per lane): the occlusion answers are TIGHT's - bit
 *         for bit in the strict build on cornell.txt, cornell-large.txt and cornell-original.txt at 1024x768x4 -
 *         for one more sphere test per shadow ray that starts on a wall.
 *   BALL  (opt-in) ENCLOSED, and where such a sphere (one that encloses some light) is still in the
 *         set of a triangle light that reaches out of it, its ray test is replaced by a test of
 *         one point: the ray starts inside the ball, so it crosses the sphere before maxt exactly
 *         when its point at maxt lies outside (|o + maxt d - centre|^2 > r^2).  cornell.txt: the
 *         floor, for the 64 light triangles of the cylinder that reach below it; no shadow ray
 *         tests a sphere any more and most batches of shadow rays have nothing left to scan
 *         (474 -> 494 Mpaths/s).  Exact in real arithmetic, but it also removes
 *         the acne of those grazing rays, which is larger than ENCLOSED's: +5e-4 of cornell.txt's
 *         radiance against the reference (tests/test_shadow_cull.py) - hence not the default.
 *         Same per-launch camera check.  Scenes of up to 32 spheres whose lights are all
 *         one-sided triangles; ENCLOSED otherwise.
 *   AUTO  = SAFE under PTGPU_MATH_STRICT, ENCLOSED under PTGPU_MATH_FAST. */
enum {
    PTGPU_CULL_OFF = 0, PTGPU_CULL_SAFE = 1, PTGPU_CULL_TIGHT = 2, PTGPU_CULL_AUTO = 3, PTGPU_CULL_ENCLOSED = 4,
    PTGPU_CULL_BALL = 5, PTGPU_CULL_ENCLOSED_SELF = 6
};
enum {
    PTGPU_MATH_FAST   = 0, /* FMA contraction on, v_rcp/v_rsq/v_sqrt/v_sin/v_cos f32 approximations:
                              statistical parity (tests state the tolerance)                        */
    PTGPU_MATH_STRICT = 1  /* no contraction, IEEE f32 divide/sqrt, and f64 exactly where the
                              reference's host build promotes to double: bit-level parity vehicle  */
};

/* replaces setUpOpenCL()'s platform/device/context/queue/program/kernel set-up,
 * reference main-gpu.c:216-506, 595-677.  Kernels are compiled ahead of time for gfx950.
 * integrator: PT_INTEGRATOR_PT (path-tracing.cl) or PT_INTEGRATOR_DL (ray-tracing.cl), the
 * reference's argv[3] (main-gpu.c:927). */
int ptgpu_create(ptgpu_ctx **out, int device_ordinal, int integrator);

/* replaces the release calls, reference main-gpu.c:58-80 */
int ptgpu_destroy(ptgpu_ctx *ctx);

int ptgpu_set_option(ptgpu_ctx *ctx, int option, int value);

/* replaces the object/material buffer creation + blocking writes, reference main-gpu.c:516-559,
 * and reInitSceneObjects(), :862-880.  `light_count` is the reference's lightCount kernel
 * argument (main-gpu.c:764): light sampling visits emissive objects in index order until that
 * many have been seen (geomfunc.h:220).  `scene` carries gamma and the two sky colours, passed
 * by value to the reference kernel (main-gpu.c:767).  Fails with PTGPU_ERR_ARGUMENT if an
 * object's material_id is >= material_count (the reference would read past the buffer). */
int ptgpu_set_scene(ptgpu_ctx *ctx, const pt_object *objects, uint32_t object_count,
                    uint32_t light_count, const pt_material *materials, uint32_t material_count,
                    const pt_scene *scene);

/* replaces the camera buffer write, reference main-gpu.c:563-586 and
 * reInitViewpointDependentBuffers(), :885-913.  The basis (dir, x, y) must already be computed
 * (pthost_update_camera_basis).  The caller resets its sample counter, as the reference does. */
int ptgpu_set_camera(ptgpu_ctx *ctx, const pt_camera *camera);

/* replaces allocateOutputBuffers()/freeBuffers(), reference main-gpu.c:85-168, 58-80:
 * (re)allocates colour (12*W*H), pixel (4*W*H) and seed (8*W*H) buffers and uploads the seeds.
 * seeds: 2*W*H words, pixel (x, y) uses seeds[2*(y*W+x)] and [..+1] (path-tracing.cl:64-75). */
int ptgpu_resize(ptgpu_ctx *ctx, int width, int height, const uint32_t *seeds);

/* replaces the 13 clSetKernelArg + executeKernel(), reference main-gpu.c:734-776, 682-713:
 * enqueues ONE pass (one path per pixel).  current_sample is the host's counter
 * (current_sample == 0 overwrites the running mean, path-tracing.cl:88-90).  Asynchronous. */
int ptgpu_render_pass(ptgpu_ctx *ctx, int current_sample);

/* n_passes consecutive passes starting at first_sample (the inner loop of updateRendering(),
 * reference main-gpu.c:782-793, without its per-pass clFinish).  Asynchronous. */
int ptgpu_render(ptgpu_ctx *ctx, int first_sample, int n_passes);

/* Same as ptgpu_render but bracketed by HIP events on the context's stream; blocks until done and
 * returns the elapsed device time in milliseconds.  (Measurement aid, no reference counterpart.) */
int ptgpu_render_timed(ptgpu_ctx *ctx, int first_sample, int n_passes, float *elapsed_ms);

/* No counterpart in the reference.  For callers that cannot allocate device memory - code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc: ptgpu_render into image
 * buffers the CALLER owns, with the wavefront kernel's work queue (counter + one word per 8x8 tile,
 * what ptgpu_resize otherwise allocates next to the image) in a static device array of the library.
 * The call allocates and frees nothing, on the device or through any library.
 *   colors_dev  W*H float4 (r, g, b, 0) at [(H-1-y)*W+x]: running mean, read unless first_sample == 0
 *   pixels_dev  W*H uint32 at [y*W+x], written
 *   seeds_dev   2*W*H uint32, pixel (x, y) at [2*(y*W+x)], read and written
 * (device pointers, 16 / 4 / 8-byte aligned).  Needs ptgpu_set_scene and ptgpu_set_camera, not
 * ptgpu_resize.  Always the wavefront kernel with the work queue, whatever PTGPU_OPT_KERNEL /
 * PTGPU_OPT_WORK_QUEUE say: the same image, bit for bit, as ptgpu_render with those two options set
 * to WAVEFRONT and 1.  Asynchronous on the context's stream; ptgpu_finish as usual.
 * Limit: PTGPU_STATIC_MAX_TILES tiles (a 4096x4096 frame; 1 MiB of device memory per GPU, held until
 * the process ends); a larger frame is refused with PTGPU_ERR_ARGUMENT.  Every call on one GPU shares
 * that GPU's array and its contents of the last call (the library clears what must start at zero).  One
 * context at a time owns a GPU's array - from its call until its ptgpu_finish or ptgpu_destroy; another
 * context's call on that GPU in between returns PTGPU_ERR_STATE instead of racing.  Calls of ONE context are
 * ordered by its stream; two threads driving one context are the caller's to serialise, as for every call here. */
#define PTGPU_STATIC_MAX_TILES 262142
int ptgpu_render_static(ptgpu_ctx *ctx, int width, int height, int first_sample, int n_passes,
                        void *colors_dev, uint32_t *pixels_dev, uint32_t *seeds_dev);

/* replaces clFinish(), reference main-gpu.c:786 */
int ptgpu_finish(ptgpu_ctx *ctx);

/* replaces the blocking read of pixelBuffer, reference main-gpu.c:803-816.
 * pixels[y*W+x] = R | G<<8 | B<<16 after gamma (path-tracing.cl:113-116); a NaN radiance packs
 * to 0x80000000 as on the reference's CPU path. */
int ptgpu_read_pixels(ptgpu_ctx *ctx, uint32_t *pixels);

/* Additions (the reference never reads colorBuffer or seedBuffer back, main-gpu.c:88, 803-827):
 * linear running-mean radiance, 3 floats per pixel at [(H-1-y)*W+x] (path-tracing.cl:87), and
 * the current RNG state.  Needed for parity checks, multi-GPU reduction and checkpointing. */
int ptgpu_read_colors(ptgpu_ctx *ctx, float *colors_xyz);
int ptgpu_read_seeds(ptgpu_ctx *ctx, uint32_t *seeds);
int ptgpu_write_colors(ptgpu_ctx *ctx, const float *colors_xyz);

/* Multi-GPU support (no reference counterpart): the device address of the colour
 * buffer so a caller can all-reduce it in place with RCCL, the stream the library works on, and a
 * re-pack of `pixels` from the (reduced) colour buffer scaled by `scale` (1/ranks after a sum
 * all-reduce of equal-pass running means).
 * On the device the sample buffer is one float4 per pixel - (r, g, b, 0) at [(H-1-y)*W+x] - so that
 * the kernels' accumulation is one 16-byte load and one 16-byte store per lane (the reference keeps
 * 3 floats per pixel, path-tracing.cl:87-111; ptgpu_read_colors / ptgpu_write_colors convert): the
 * buffer behind ptgpu_colors_device_ptr is PTGPU_COLOR_STRIDE_FLOATS * W * H floats, and a sum over
 * all of them leaves the spare channel 0. */
#define PTGPU_COLOR_STRIDE_FLOATS 4
int ptgpu_colors_device_ptr(ptgpu_ctx *ctx, void **device_ptr);
int ptgpu_stream(ptgpu_ctx *ctx, void **hip_stream);
int ptgpu_scale_and_repack(ptgpu_ctx *ctx, float scale);

/* Counters of the last set_scene (what was staged on the device). */
typedef struct ptgpu_scene_info {
    uint32_t primitives;   /* spheres + triangles (MODEL placeholders dropped) */
    uint32_t spheres;
    uint32_t triangles;
    uint32_t lights;       /* min(light_count, emissive primitives) */
    uint32_t segments;     /* runs of equal primitive type in index order */
    uint32_t lds_bytes;    /* LDS the staged records take per workgroup in the fast build (0 = not staged) */
    uint32_t shadow_groups;        /* primitive groups in the shadow-ray ordering (0 = no cull tables) */
    float    shadow_tests_safe;    /* mean primitives tested per shadow ray under SAFE, light-averaged */
    float    shadow_tests_tight;   /* same under TIGHT; equals `primitives` padded when nothing is culled */
    float    shadow_tests_enclosed; /* same under ENCLOSED */
    uint32_t clusters;             /* triangle clusters the fast build bins shadow rays on (0..8) */
    uint32_t clustered_triangles;  /* triangles inside them */
    uint32_t bvh_nodes;            /* nodes of the hierarchy built for PTGPU_OPT_BVH (0 = none) */
    uint32_t bvh_primitives;       /* primitives inside it; the others are scanned one by one */
    uint32_t convex_clusters;      /* clusters that are closed convex meshes of <= 15 planes: near shadow rays test
                                      their face planes instead of their triangles (PTGPU_OPT_RAY_BINS) */
} ptgpu_scene_info;
int ptgpu_get_scene_info(ptgpu_ctx *ctx, ptgpu_scene_info *info);

/* What the last render call on this context launched (measurement aid, no reference counterpart):
 * the library picks kernel, code instance, candidate sets and launch shape per scene, and bench.py
 * names them in its output instead of re-deriving the rules. */
enum { PTGPU_INSTANCE_PLAIN = 0, PTGPU_INSTANCE_BVH = 1, PTGPU_INSTANCE_HYBRID = 2, PTGPU_INSTANCE_BALL = 3 };
typedef struct ptgpu_launch_info {
    int kernel;            /* PTGPU_KERNEL_MEGA or _WAVEFRONT (AUTO resolved)                     */
    int math;              /* PTGPU_MATH_*                                                        */
    int instance;          /* PTGPU_INSTANCE_*: which build of the kernel source ran              */
    int work_queue;        /* 1 = one launch for the call, units from the queue                   */
    int unit_passes;       /* passes per queue unit (0 without the queue)                         */
    int passes_per_launch; /* passes inside one launch                                            */
    int regions;           /* concurrent region streams (1 with the queue)                        */
    unsigned lds_bytes;    /* LDS of the staged records per workgroup (0 = read from global)      */
    int bvh_nodes_only;    /* hierarchy staged nodes only                                         */
    int shadow_cull;       /* PTGPU_CULL_* in effect (OFF when the scene has no tables)           */
    int ray_bins;          /* shadow rays binned on triangle clusters                             */
} ptgpu_launch_info;
int ptgpu_get_launch_info(ptgpu_ctx *ctx, ptgpu_launch_info *info);

/* Work the kernels EXECUTED since the last reset, summed over every kernel instance, per active
 * lane (measurement aid, no reference counterpart; bench.py builds its roofline from it).  Only the
 * counting build of the library (-DPT_COUNT_WORK=1: path-tracer_amd/variants/libptgpu_count.so,
 * scripts/build_variants.sh) keeps them - it is slow and never the timed one; the regular library
 * returns PTGPU_ERR_STATE.  Waits for the context's stream first. */
enum {
    PTGPU_WORK_SPHERE_TESTS = 0,   /* geomfunc.h:10-39, ~20 flops                                   */
    PTGPU_WORK_TRIANGLE_TESTS_MT,  /* Moeller-Trumbore, geomfunc.h:45-78, ~52 flops                  */
    PTGPU_WORK_TRIANGLE_TESTS_PLANES, /* plane form (fast build), ~32 flops                         */
    PTGPU_WORK_BVH_NODES,          /* slab tests of the hierarchy walks, ~22 flops                   */
    PTGPU_WORK_LIGHT_SAMPLES,      /* geomfunc.h:227-300 evaluated (after the plane-side test), ~70  */
    PTGPU_WORK_LIGHT_POINTS,       /* sampled light point (part of a sample; once more for a binned ray) */
    PTGPU_WORK_PLANE_SIDE_TESTS,   /* hit point against a light's plane, 5 flops                     */
    PTGPU_WORK_SEGMENT_NEAR_TESTS, /* shadow segment against a cluster's bounding sphere, ~21        */
    PTGPU_WORK_BALL_POINT_TESTS,   /* PTGPU_CULL_BALL end-point tests, ~8 each                       */
    PTGPU_WORK_EXTENSION_RAYS,     /* closest-hit rays incl. their shading, ~120 beside the tests    */
    PTGPU_WORK_SHADOW_RAYS,        /* any-hit rays traced                                            */
    PTGPU_WORK_RNG_DRAWS,          /* GetRandom calls (integer work)                                 */
    PTGPU_WORK_PATHS,
    PTGPU_WORK_NODE_LANE_SLOTS,    /* 64 per wave and node step: with BVH_NODES, the lane utilisation of the walks */
    PTGPU_WORK_PRIM_LANE_SLOTS,    /* 64 per wave and primitive test: likewise, against the three test counters   */
    PTGPU_WORK_CONVEX_PLANE_TESTS, /* face planes of a closed convex cluster against a shadow segment, ~13 each    */
    PTGPU_WORK_SLOTS = 16
};
int ptgpu_work_counters(ptgpu_ctx *ctx, unsigned long long *counters /* [PTGPU_WORK_SLOTS] */, int reset);

/* Test aid without a reference counterpart (the reference traces rays only from inside its path loop): the closest
 * hit and the occlusion answer of rays the caller names, through the intersection routines of the build and code
 * instance the context's options select - the place where the tie rule (geomfunc.h:149) and the hierarchy walks can be
 * checked ray by ray (tests/test_gpu_rays.py).  origins / directions: 3 floats per ray (directions as the path
 * loop would pass them: normalised); maxt: 1 float per ray; t: distance of the closest hit or -1; rgb: colour +
 * emission of the surface hit; occluded: 1 if anything is hit between EPSILON and maxt.  cooperative != 0: where a
 * hierarchy is walked, through the wavefront kernel's wave-cooperative walks; cooperative == 2 (fast build with ray
 * bins): occluded = bit 0 the face-plane test of the scene's closed convex clusters, bit 1 the any-hit scan over the
 * same clusters' triangles (t and rgb untouched).  Needs a scene and a frame
 * (ptgpu_resize).  Only in the ray-probe build of the library (-DPT_RAY_PROBE=1, path-tracer_amd/variants/
 * libptgpu_probe.so); elsewhere it returns PTGPU_ERR_STATE. */
int ptgpu_probe_rays(ptgpu_ctx *ctx, int n, const float *origins, const float *directions, const float *maxt,
                     float *t, float *rgb, int *occluded, int cooperative);

/* The candidate sets behind PTGPU_OPT_SHADOW_CULL as plain data (host only, no device needed; no
 * reference counterpart - the reference scans every object, geomfunc.h:162-193).  For the scene
 * ptgpu_set_scene would stage from the same arguments and mode SAFE, TIGHT, ENCLOSED, BALL or ENCLOSED_SELF (OFF:
 * all ones; the ENCLOSED family as if the camera were inside every sphere the rule relies on),
 * keep[l * object_count + i] = 1 when object i is scanned for a shadow ray towards the l-th light
 * (sampling order), 0 when it is culled or is not a sphere/triangle, 2 (BALL) when sphere i is
 * tested by the ray's point at maxt instead of the ray, 4 (the ENCLOSED family) when sphere i is tested only by the
 * rays of a path vertex that may lie outside a dropped sphere (see ENCLOSED above), 3 (ENCLOSED_SELF) likewise but
 * including the rays that start on it.  *n_lights receives the number
 * of lights; call with keep = NULL to size the buffer (n_lights * object_count bytes).  Used by the
 * tests to replay the culled scan on the CPU and by bench.py to count the tests it leaves.
 * No material table is passed: an object counts as a dielectric (it encloses nothing) when its
 * material_id is 2, the slot of the reference's built-in glass (displayfunc.c:147-155);
 * ptgpu_set_scene also looks at materials[material_id].type, so for a caller-made table that keeps
 * glass in another slot the sets staged on the device can be larger than the ones returned here. */
int ptgpu_shadow_candidates(const pt_object *objects, uint32_t object_count, uint32_t light_count,
                            int mode, uint8_t *keep, uint32_t *n_lights);

/* The triangle clusters behind PTGPU_OPT_RAY_BINS as plain data (host only, no device needed; no reference
 * counterpart): *clusters = the clusters ptgpu_set_scene would bin shadow rays on for these objects (connected
 * components of the meshes, the 8 largest with >= 8 triangles), convex_planes_per_cluster[c] (8 entries) = the
 * face planes cluster c is tested by when it is a closed convex mesh WITH A VOLUME (4 .. 15 planes), 0 when it keeps
 * its triangles.  Used by the tests: a flat double-sided polygon is closed by the edge count and convex by the vertex
 * test, and must still keep its triangles. */
int ptgpu_cluster_info(const pt_object *objects, uint32_t object_count, uint32_t light_count, uint32_t *clusters,
                       uint32_t *convex_planes_per_cluster);

const char *ptgpu_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_H */
