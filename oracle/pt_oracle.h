/*
 * pt_oracle.h - TEST INFRASTRUCTURE: CPU restatement of the reference's radiance path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and
 * only as the checker / the timed CPU baseline.  The product (libptgpu.so) never links, loads or
 * falls back to anything in oracle/.
 *
 * Parity status: PINNED.  Every function here is bit-identical (colors, pixels, seeds, NaN
 * positions) to the reference's own geomfunc.h/simplernd.h/camera.c compiled by gcc from
 * /root/reference (oracle/_ref, see tests/test_oracle_vs_ref.py), and to the known answers the
 * survey recorded from the reference's main-cpu.c (tests/test_known_answers.py).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "pt_records.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Work counters of the reference algorithm:
 *   F = 20*sphere_tests + 52*triangle_tests + 70*light_samples + 120*extension_rays */
typedef struct pt_oracle_counters {
    uint64_t sphere_tests;
    uint64_t triangle_tests;
    uint64_t light_samples;
    uint64_t extension_rays;
    uint64_t shadow_rays;
    uint64_t rng_draws;
    uint64_t paths;
} pt_oracle_counters;

enum { ORACLE_SEEDS_CPU_LAYOUT = 0,   /* seeds[2*((H-1-y)*W+x)], reference main-cpu.c:63-66    */
       ORACLE_SEEDS_KERNEL_LAYOUT = 1 /* seeds[2*(y*W+x)],       reference path-tracing.cl:64-75 */ };

float oracle_get_random(uint32_t *seed0, uint32_t *seed1);                       /* simplernd.h:11-25 */
void  oracle_camera_ray(const pt_camera *cam, int x, int y, int width, int height,
                        uint32_t *seed0, uint32_t *seed1, pt_ray *out);           /* main-cpu.c:57-82 */
int   oracle_intersect(const pt_object *objects, uint32_t n, const pt_ray *ray,
                       float *t, uint32_t *id);                                  /* geomfunc.h:126-157 */
int   oracle_intersect_p(const pt_object *objects, uint32_t n, const pt_ray *ray,
                         float maxt);                                            /* geomfunc.h:162-193 */
void  oracle_sample_lights(const pt_object *objects, uint32_t n, uint32_t lights,
                           uint32_t *seed0, uint32_t *seed1, const pt_vec3 *hit,
                           const pt_vec3 *normal, pt_vec3 *result);              /* geomfunc.h:199-306 */
void  oracle_radiance(const pt_object *objects, uint32_t n, uint32_t lights, const pt_ray *ray,
                      uint32_t *seed0, uint32_t *seed1, const pt_scene *scene,
                      const pt_material *materials, int integrator,
                      pt_vec3 *radiance);                       /* geomfunc.h:494-798 / 800-1014 */
uint32_t oracle_accumulate(pt_vec3 *color, const pt_vec3 *sample, int current_sample,
                           float gamma);                                         /* main-cpu.c:89-103 */

/* One progressive pass, main-cpu.c:54-110.  colors[(H-1-y)*W+x], pixels[y*W+x].
 * threads <= 1 is the reference's serial loop; > 1 splits rows over OpenMP threads (pixels are
 * independent, results are identical). */
void oracle_render_pass(const pt_object *objects, uint32_t n, uint32_t lights,
                        const pt_material *materials, const pt_camera *cam, const pt_scene *scene,
                        int width, int height, uint32_t *seeds, pt_vec3 *colors, uint32_t *pixels,
                        int current_sample, int integrator, int seed_layout, int threads);

/* Same, serial, also counting the work done (counters are added to, not reset). */
void oracle_render_pass_counted(const pt_object *objects, uint32_t n, uint32_t lights,
                                const pt_material *materials, const pt_camera *cam,
                                const pt_scene *scene, int width, int height, uint32_t *seeds,
                                pt_vec3 *colors, uint32_t *pixels, int current_sample,
                                int integrator, int seed_layout, pt_oracle_counters *counters);

/* Replays the product's shadow-ray culling in the restatement (NOT part of the reference; NULL,
 * the default, is the reference's full scan): keep[l * n + i] = 0 makes the any-hit scan of a
 * shadow ray towards the l-th light skip object i; 2 (PTGPU_CULL_BALL, spheres only) replaces the
 * ray test of object i by "the ray's point at maxt lies outside the sphere"; 4 (the ENCLOSED family) keeps
 * the test of sphere i for the rays of a path vertex that may lie outside a dropped sphere (oracle_set_enclosing),
 * 3 (PTGPU_CULL_ENCLOSED_SELF) also for the rays that start on sphere i.  The matrix comes from
 * ptgpu_shadow_candidates (include/ptgpu.h) with the same objects and light count as the renders
 * that follow; the caller keeps it alive until it is reset.  Used to prove on the CPU that a cull
 * mode leaves every bit of the image alone, and to count the tests the culled algorithm performs. */
void oracle_set_shadow_candidates(const uint8_t *keep);
/* flags[i] != 0 (n objects): sphere i is one the ENCLOSED family dropped for some light = any byte >= 2 in column i of
 * the matrix above.  With it set, a path vertex counts as "may be outside" when the ray that found it did not get
 * the FAR root of every such sphere from SphereIntersect (geomfunc.h:10-39: what a ray that starts inside a sphere
 * and does not leave it within EPSILON gets) - the product's criterion, restated on the reference's own test. */
void oracle_set_enclosing(const uint8_t *flags);

/* Diagnostic of that replay (serial renders: threads <= 1): every shadow ray whose culled scan answers differently
 * from the full scan is logged, 12 floats a record - origin xyz, direction xyz, maxt, light (sampling order), the
 * object the ray starts on, the first object the full scan reports, its distance, the culled answer. */
void oracle_set_mismatch_log(float *records, int capacity);
int  oracle_mismatch_count(void);
/* every ray of a serial render, 12 floats a record (radiance_oracle.c) */
void oracle_set_ray_log(float *records, long capacity);
long oracle_ray_count(void);

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
