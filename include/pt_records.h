/*
 * pt_records.h - byte-exact host records shared by the C host, the CPU oracle and
 * the HIP library.  These are the structures the reference uploads to its OpenCL
 * kernel verbatim, so the C-ABI has to accept them byte for byte:
 *
 *   pt_vec3      12 B   reference vec.h:5-7
 *   pt_ray       24 B   reference geom.h:14-16
 *   pt_object    84 B   reference geom.h:26-43   (type 0, materialId 2, emission 4, color 16,
 *                                                 radius 28, center 32, p1 44, p2 56, p3 68, area 80)
 *   pt_material  44 B   reference material.h:16-32 (type 0 (+2 pad), kd 4, ks 16, sigma 28,
 *                                                 roughness 32, A 36, B 40)
 *   pt_camera    76 B   reference camera.h:7-18  (orig 0, target 12, pitch 24, yaw 28, width 32,
 *                                                 height 36, dir 40, x 52, y 64)
 *   pt_scene     32 B   reference scene.h:6-10   (gamma 0, sky1 4, sky2 16, materialCount 28)
 *
 * The sizes/offsets are checked at compile time below and again by tests/test_records.py.
 */
#ifndef PT_RECORDS_H
#define PT_RECORDS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_vec3 {
    float x, y, z;
} pt_vec3;

typedef struct pt_ray {
    pt_vec3 o, d;
} pt_ray;

/* object kinds, reference geom.h:22-24 */
enum { PT_SPHERE = 0, PT_TRIANGLE = 1, PT_MODEL = 2 };

/* material kinds, reference material.h:6-13 */
enum {
    PT_LAMBERTIAN = 0,
    PT_CONDUCTOR  = 1,
    PT_DIELECTRIC = 2,
    PT_MATTE      = 3,
    PT_PLASTIC    = 4,
    PT_METAL      = 5
};

typedef struct pt_object {
    int16_t type;        /* PT_SPHERE / PT_TRIANGLE / PT_MODEL (placeholder, never intersected) */
    int16_t material_id; /* index into the material table                                         */
    pt_vec3 emission;
    pt_vec3 color;
    float   radius;      /* spheres   */
    pt_vec3 center;      /* spheres   */
    pt_vec3 p1, p2, p3;  /* triangles */
    float   area;        /* 4*pi*r^2 or |e1 x e2|/2, used when the object is a light */
} pt_object;

typedef struct pt_material {
    int16_t type;
    int16_t pad_;
    pt_vec3 kd;
    pt_vec3 ks;
    float   sigma;
    float   roughness;
    float   A, B;        /* Oren-Nayar terms precomputed by the scene reader */
} pt_material;

typedef struct pt_camera {
    pt_vec3 orig, target;
    float   pitch, yaw;
    float   width, height;
    pt_vec3 dir, x, y;
} pt_camera;

typedef struct pt_scene {
    float    gamma;
    pt_vec3  sky1, sky2;
    uint32_t material_count;
} pt_scene;

/* integrators: 0 = path tracing (reference path-tracing.cl / geomfunc.h:494),
 *              1 = direct lighting (reference ray-tracing.cl / geomfunc.h:800) */
enum { PT_INTEGRATOR_PT = 0, PT_INTEGRATOR_DL = 1 };

#define PT_EPSILON 0.01f /* reference geom.h:6 */
#define PT_MAX_DEPTH 6   /* segments 0..6, reference geomfunc.h:520 */

#if defined(__cplusplus)
#define PT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define PT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

PT_STATIC_ASSERT(sizeof(pt_vec3) == 12, "pt_vec3 must be 12 bytes");
PT_STATIC_ASSERT(sizeof(pt_ray) == 24, "pt_ray must be 24 bytes");
PT_STATIC_ASSERT(sizeof(pt_object) == 84, "pt_object must be 84 bytes");
PT_STATIC_ASSERT(offsetof(pt_object, emission) == 4, "pt_object.emission");
PT_STATIC_ASSERT(offsetof(pt_object, color) == 16, "pt_object.color");
PT_STATIC_ASSERT(offsetof(pt_object, radius) == 28, "pt_object.radius");
PT_STATIC_ASSERT(offsetof(pt_object, center) == 32, "pt_object.center");
PT_STATIC_ASSERT(offsetof(pt_object, p1) == 44, "pt_object.p1");
PT_STATIC_ASSERT(offsetof(pt_object, p2) == 56, "pt_object.p2");
PT_STATIC_ASSERT(offsetof(pt_object, p3) == 68, "pt_object.p3");
PT_STATIC_ASSERT(offsetof(pt_object, area) == 80, "pt_object.area");
PT_STATIC_ASSERT(sizeof(pt_material) == 44, "pt_material must be 44 bytes");
PT_STATIC_ASSERT(offsetof(pt_material, kd) == 4, "pt_material.kd");
PT_STATIC_ASSERT(offsetof(pt_material, sigma) == 28, "pt_material.sigma");
PT_STATIC_ASSERT(offsetof(pt_material, A) == 36, "pt_material.A");
PT_STATIC_ASSERT(sizeof(pt_camera) == 76, "pt_camera must be 76 bytes");
PT_STATIC_ASSERT(offsetof(pt_camera, dir) == 40, "pt_camera.dir");
PT_STATIC_ASSERT(sizeof(pt_scene) == 32, "pt_scene must be 32 bytes");
PT_STATIC_ASSERT(offsetof(pt_scene, material_count) == 28, "pt_scene.material_count");

#ifdef __cplusplus
}
#endif
#endif /* PT_RECORDS_H */
