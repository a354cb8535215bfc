// device_params.h - what the host side of libptgpu hands to the kernels, and the packed
// device-side scene layout.  Shared by ptgpu_api.hip and the two kernel translation units.
//
// Device scene layout (built once per ptgpu_set_scene, all arrays of float4 in HBM, read-only):
//
//   prims  [3 * n_prims]   primitive p = the p-th non-MODEL object in index order
//        sphere  : a = (center.xyz, radius)  b = (radius*radius, dropped, 0, 0)   c = 0   (dropped = 1: see enc_list below)
//        triangle: a = (p1.xyz, 0)           b = (e1.xyz, 0) = p2 - p1      c = (e2.xyz, 0) = p3 - p1
//     (e1, e2 and radius^2 are the same single float operations the reference performs on every
//      test, geomfunc.h:17, 48-49, so hoisting them is bit-neutral.)
//   segs   [n_segs]        runs of equal primitive type in index order: (type, begin, end, 0).
//     Index order is kept so closest-hit ties resolve to the lowest index as in the reference
//     (geomfunc.h:149); the run structure makes the inner loops type-pure.
//   shade  [4 * n_prims]   s0 = (emission.xyz, flags)  flags = material type | emissive << 8 | prim type << 16
//                          s1 = (color.xyz, sigma)  s2 = (kd.xyz, A)  s3 = (B, 0, 0, 0)
//   lights [5 * n_lights]  the first min(lightCount, #emissive) emissive primitives, index order:
//                          l0 = (p1 | center, type)  l1 = (e1 | radius,0,0; w = bit c set when cluster c has a
//                          group in the light's TIGHT set)  l2 = (e2; w = bit g set when sphere group g is tested by the
//                          ray's end point under PTGPU_CULL_BALL)  l3 = (emission.xyz, area)
//                          l4 = (unit normal of e1 x e2, plane offset n.p1 lowered by a tolerance) for
//                               triangle lights; (0,0,0,-1) for spheres and degenerate triangles.
//                               A point x with n.x < l4.w is behind the light: every sample of the
//                               light is rejected there (geomfunc.h:276-285).  Read by the fast build only.
//
// Shadow-ray tables (optional, see PTGPU_OPT_SHADOW_CULL in ptgpu.h):
//   sprims [3 * (sg_sph + 4 sg_tri)]  the same primitive records in shadow order: all spheres
//                          (those that most lights keep under ENCLOSED first), padded to a multiple
//                          of kSphereGroup with never-hit records, then all triangles, padded to a
//                          multiple of kTriGroup.  A group = that many consecutive primitives of
//                          one type, tested back to back.
//                          (An any-hit scan does not depend on the visiting order.)
//     The fast build (outside its BVH translation unit) reads its own copy of this array, with the
//     triangles in PLANE FORM (radiance_device.inc, triangle_planes_t):
//       q0 = (n, n.p1)  q1 = (n1, -n1.p1)  q2 = (n2, -n2.p1),  n = e1 x e2, n1 = e2 x n / n.n, n2 = n x e1 / n.n
//     Triangles are ordered cluster by cluster (connected components of the meshes, the 8 largest
//     with >= 8 triangles), each cluster padded to whole groups, unclustered triangles last.
//   smask  [n_lights * (mw_sph + mw_tri)]  per light, 64-bit words: bit g of the first mw_sph words
//                          = sphere group g may occlude a shadow ray towards this light; the next
//                          mw_tri words the same for triangle groups.
//
// BVH (optional, PTGPU_OPT_BVH; built by bvh_build.h): one array `bvh` of quads
//   [2 * n_bnodes]  nodes in depth-first order: (lo.xyz, skip) (hi.xyz, first) - a ray that misses
//                   the padded box continues at node `skip`; otherwise at the next node, or, when
//                   first >= 0, tests the 4 primitive slots first .. first + 3 and continues at `skip`
//   [3 * n_bslots]  primitive records of ONE type (bvh_type) in leaf order, never-hit fillers included
//   [3 * n_brest]   the other primitives, scanned one by one; the type is in the third quad's w
// bvh_ids[slot] / bvh_rest_ids[r] = the primitive's index in the build's own numbering (index order
// for the strict build, shadow order for the fast build): what load_surface takes and what breaks
// a tie between equal distances.  While the BVH is active the block is staged in LDS in place of
// the build's primary primitive array and the per-light candidate sets are off.  A block too large
// for LDS whose nodes alone fit (a mesh of ~1 500 to ~5 600 triangles: 16 B of nodes per triangle
// against 48 B of records) is staged nodes only, for the wavefront kernel: the walk - most of the
// accesses - stays in LDS and each lane reads the few leaves it reaches from global memory.
// The fast build's copy of the block holds the leaf triangles in plane form (the strict build's, and
// the primitives outside the hierarchy in both, keep p1 / e1 / e2).  The depth-first order with skip
// links is also what lets the wavefront kernel cut a walk anywhere: what is left of a walk is an index
// range of the node array (radiance_kernels.inc, coop_walk).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

constexpr int kLightQuads = 5;
constexpr int kSphereGroup = 1, kTriGroup = 4; // primitives per shadow group, tested back to back with no exit check
                                               // in between.  One sphere per group: with ENCLOSED most rays test one sphere
                                               // or none, and the device scan stays equal to the set sphere for sphere

struct PtDeviceParams {
    const float4 *prims;
    const float4 *shade;
    const float4 *lights;
    const int4   *segs;
    int n_prims, n_segs, n_lights, integrator;
    float cam_orig[3], cam_dir[3], cam_x[3], cam_y[3];
    float sky1[3], sky2[3];
    float gamma;
    int width, height;
    int current_sample;   // sample index of the first pass of this launch
    int n_passes;         // passes done inside this launch (wavefront kernel keeps state in registers)
    float4   *colors;     // (r, g, b, 0) per pixel at (H-1-y)*W+x: one 16-byte access per lane
    uint32_t *pixels;     // y*W+x
    uint2    *seeds;      // the two generator words of the pixel at y*W+x: one 8-byte access
    int tiles_x;          // image width in 8x8 pixel tiles (one wave = one tile)
    int n_tiles;
    // this launch covers tiles tile_first + k * tile_stride, k in [0, n_launch_tiles): the image is
    // dealt round-robin into regions so independent regions can run on concurrent streams
    int tile_first, tile_stride, n_launch_tiles;
    // Work queue of the wavefront kernel (null: one tile per wave, as laid out above).  work[1] = the
    // next unit, unit = chunk * n_tiles + tile, a chunk being unit_passes consecutive passes
    // (chunk-major, so a tile's passes are taken in order); work[2 + tile] = chunks of this launch
    // already stored for the tile; both zeroed by the host before the launch.  work[0] = set when a
    // wave gave up waiting (never, unless a wave died); sticky until the host has read it.
    int *work;
    int unit_passes;
    // PTGPU_CULL_BALL: the sphere groups (shadow order, first 32) that some light tests by the ray's
    // point at maxt instead of the ray - which lights, the light record says (l2.w); 0 = no such test
    unsigned ball_groups;
    // The ENCLOSED family of candidate sets (ptgpu.h) drops opaque spheres that hold a whole light for the rays that
    // start INSIDE them: n_enc (<= 32) such spheres - in the fast build slots [0, n_enc) of the shadow order, in the
    // strict build the records whose second quad says so in .y, named by enc_list[0 .. n_enc) (index order).
    // The extension rays watch those spheres' answers: a path vertex found by a ray that did not get the far root of
    // every one of them may lie outside one (upstream's paths do leave their room through its corners), and its
    // shadow rays test them all.  self_sphere (PTGPU_CULL_ENCLOSED_SELF): a shadow ray is also tested against the
    // dropped sphere its origin stands on.
    const int *enc_list;
    int n_enc;
    int self_sphere;
    // shadow order: always present (the fast build traces every ray in it); smask == nullptr
    // switches the per-light candidate sets off
    const float4 *sprims;
    const float4 *sshade;   // shade records in shadow order, zero for the fillers
    const unsigned long long *smask;
    int sg_sph, sg_tri;   // sphere / triangle groups in sprims
    int n_sph_real;       // spheres before the fillers of the sphere section (full scans stop there)
    int mw_sph, mw_tri;   // mask words per light for each
    // > 0 (only when mw_sph == mw_tri == 1): smask continues with a range table - level k, entry i, at
    // smask[(k * n_lights + i) * 2 ..] = union of the sets of lights i .. i + 2^k - 1 (level 0 = the sets
    // themselves); the union over any range of lights is the OR of two entries of one level
    int smask_levels;
    // triangle clusters (fast build, TIGHT sets): n_clusters == 0 switches the ray binning off
    const unsigned long long *cmask;  // (1 + n_clusters) x mw_tri words: row 0 = triangle groups in no
                                      // cluster, row 1 + c = groups of cluster c
    const float4 *cbounds;            // cluster table, n_cquads quads staged after the lights: n_clusters x (centre, padded
                                      // radius), then the outward face planes (unit normal, offset) of the clusters that
                                      // are closed convex meshes, cluster by cluster
    int n_clusters, n_cquads;
    unsigned convex_counts;           // planes of cluster c in bits 4c..4c+3 (0 = not a closed convex mesh)
    // BVH; n_bnodes == 0 switches it off
    const float4 *bvh;
    const int *bvh_ids, *bvh_rest_ids;
    int n_bnodes, n_bslots, n_brest, bvh_type;
    int bvh_nodes_only;   // 1: only the nodes (and the lights) are staged in LDS, leaves are read from global memory
    int coop_stragglers;  // cooperative walks (radiance_kernels.inc): the lanes still walking towards a leaf stop for
                          // now once fewer than this many are left, and what they have left is shared out (0: never)
    // hybrid (fast build with candidate sets): a hierarchy whose leaves are the triangle groups of
    // sprims (leaf `first` = slot within the triangle section); n_hnodes == 0 switches it off
    const float4 *hyb_nodes;
    int n_hnodes;
};

struct PtLaunchConfig {
    int use_lds;          // stage prims+lights in LDS
    unsigned lds_bytes;   // dynamic LDS for the staged records (see stage_records)
    int waves_per_group;  // wavefront kernel: waves per workgroup (0 = default)
    int compute_units;    // of the device (work-queue launches fill the chip once, no more)
};


// -DPT_RAY_PROBE=1 builds only (tests/test_gpu_rays.py): the closest hit and the occlusion answer of given rays
struct PtRayProbe {
    int n;
    const float *origin, *direction, *maxt; // 3, 3, 1 floats per ray
    float *t, *rgb;                         // distance of the closest hit (-1: none), colour + emission of its surface
    int *occluded;                          // any hit before maxt
    int cooperative;                        // hierarchy instances: through the wavefront kernel's cooperative walks
};
namespace ptk_fast {
int work_counters(unsigned long long *out, int reset); // -DPT_COUNT_WORK=1 builds only
int launch_mega(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_wavefront(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_probe(const PtDeviceParams &p, const PtRayProbe &r, hipStream_t s);
int launch_repack(const PtDeviceParams &p, float scale, hipStream_t s);
}
namespace ptk_strict {
int work_counters(unsigned long long *out, int reset); // -DPT_COUNT_WORK=1 builds only
int launch_mega(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_wavefront(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_probe(const PtDeviceParams &p, const PtRayProbe &r, hipStream_t s);
int launch_repack(const PtDeviceParams &p, float scale, hipStream_t s);
}
// the same kernels compiled with the BVH traversal in place of the linear scans (PT_BVH 1)
namespace ptk_fast_bvh {
int work_counters(unsigned long long *out, int reset); // -DPT_COUNT_WORK=1 builds only
int launch_mega(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_wavefront(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_probe(const PtDeviceParams &p, const PtRayProbe &r, hipStream_t s);
}
namespace ptk_fast_hyb {
int work_counters(unsigned long long *out, int reset); // -DPT_COUNT_WORK=1 builds only // PT_BVH 2: extension rays through the hierarchy over the shadow order
int launch_mega(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_wavefront(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_probe(const PtDeviceParams &p, const PtRayProbe &r, hipStream_t s);
}
namespace ptk_strict_bvh {
int work_counters(unsigned long long *out, int reset); // -DPT_COUNT_WORK=1 builds only
int launch_mega(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_wavefront(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_probe(const PtDeviceParams &p, const PtRayProbe &r, hipStream_t s);
}
// PTGPU_CULL_BALL: the hybrid fast build and the strict build with the point test compiled in
namespace ptk_fast_ball {
int work_counters(unsigned long long *out, int reset); // -DPT_COUNT_WORK=1 builds only
int launch_mega(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_wavefront(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_probe(const PtDeviceParams &p, const PtRayProbe &r, hipStream_t s);
}
namespace ptk_strict_ball {
int work_counters(unsigned long long *out, int reset); // -DPT_COUNT_WORK=1 builds only
int launch_mega(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_wavefront(const PtDeviceParams &p, const PtLaunchConfig &c, hipStream_t s);
int launch_probe(const PtDeviceParams &p, const PtRayProbe &r, hipStream_t s);
}
