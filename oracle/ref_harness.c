/*
 * ref_harness.c - TEST INFRASTRUCTURE.  Builds the reference's OWN radiance code into
 * oracle/_ref/libptref.so so the CPU restatement (oracle/radiance_oracle.c) can be checked
 * against it bit for bit.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.
 *
 * What is the reference's and what is ours:
 *   - geomfunc.h / simplernd.h / vec.h / geom.h / material.h / camera.h / scene.h and camera.c are
 *     compiled from where they lie under /root/reference (see oracle/Makefile), unmodified, with the
 *     reference's flags (Makefile:5) plus `-include stdbool.h` (the CPU target does not compile
 *     without it under gcc 11: geomfunc.h:461,516 use `bool`).
 *   - the pixel loop below restates main-cpu.c:54-110 (it cannot be compiled from main-cpu.c, which
 *     pulls in displayfunc.h -> GL/freeglut.h, absent from this image), with the integrator
 *     selectable (main-cpu.c:84-87 has the direct-lighting call commented out).
 *   - readScene() (displayfunc.c) is NOT built: it needs GL/freeglut.h.  Scenes come from
 *     libpthost's reader; that reader is pinned end to end by the known answers in SURVEY.md A.2.
 *
 * The records are passed as raw bytes; their layout is the reference's own (checked by
 * ref_record_sizes against include/pt_records.h in the tests).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "camera.h"
#include "geom.h"
#include "geomfunc.h"
#include "scene.h"
#include "simplernd.h"

void ref_record_sizes(int out[6]) {
    out[0] = (int)sizeof(vec);
    out[1] = (int)sizeof(Ray);
    out[2] = (int)sizeof(Object);
    out[3] = (int)sizeof(Material);
    out[4] = (int)sizeof(Camera);
    out[5] = (int)sizeof(Scene);
}

float ref_get_random(unsigned int *seed0, unsigned int *seed1) { return GetRandom(seed0, seed1); }

void ref_update_camera_basis(Camera *camera) { updateCameraBasis(camera); }

int ref_intersect(const Object *objects, unsigned int n, const Ray *ray, float *t, unsigned int *id) {
    return Intersect(objects, n, ray, t, id);
}

int ref_intersect_p(const Object *objects, unsigned int n, const Ray *ray, float maxt) {
    return IntersectP(objects, n, ray, maxt);
}

void ref_sample_lights(const Object *objects, unsigned int n, unsigned int lights, unsigned int *seed0,
                       unsigned int *seed1, const vec *hit, const vec *normal, vec *result) {
    SampleLights(objects, n, lights, seed0, seed1, hit, normal, result);
}

void ref_radiance(const Object *objects, unsigned int n, unsigned int lights, const Ray *ray,
                  unsigned int *seed0, unsigned int *seed1, const Scene *scene,
                  const Material *materials, int integrator, vec *radiance) {
    if (integrator == 0)
        RadiancePathTracing(objects, n, lights, ray, seed0, seed1, scene->skyColor1, scene->skyColor2,
                            materials, radiance);
    else
        RadianceDirectLighting(objects, n, lights, ray, seed0, seed1, scene->skyColor1,
                               scene->skyColor2, materials, radiance);
}

/* Primary ray through pixel (x, y) built with the reference's own vec.h macros, following
 * main-cpu.c:66-82 (same math as GenerateCameraRay, path-tracing.cl:9-47). */
static Ray harness_camera_ray(const Camera *c, int x, int y, int w, int h, unsigned int *s0,
                              unsigned int *s1) {
    const float jx = GetRandom(s0, s1) - .5f;
    const float jy = GetRandom(s0, s1) - .5f;
    const float kcx = (x + jx) * (1.f / w) - .5f;
    const float kcy = (y + jy) * (1.f / h) - .5f;
    Ray r;
    vinit(r.d, c->x.x * kcx + c->y.x * kcy + c->dir.x, c->x.y * kcx + c->y.y * kcy + c->dir.y,
          c->x.z * kcx + c->y.z * kcy + c->dir.z);
    vsmul(r.o, 0.1f, r.d);
    vadd(r.o, r.o, c->orig);
    vnorm(r.d);
    return r;
}

/* Running mean + 8-bit pack, main-cpu.c:89-103, using the reference's toInt macro. */
static unsigned int harness_accumulate(vec *c, const vec *sample, int k, float gamma) {
    if (k == 0) {
        *c = *sample;
    } else {
        const float k1 = k;
        const float k2 = 1.f / (k1 + 1.f);
        c->x = (c->x * k1 + sample->x) * k2;
        c->y = (c->y * k1 + sample->y) * k2;
        c->z = (c->z * k1 + sample->z) * k2;
    }
    return (toInt(c->x, gamma)) | (toInt(c->y, gamma) << 8) | (toInt(c->z, gamma) << 16);
}

/* One progressive pass over the whole image in main-cpu.c's order (rows bottom-up in `colors`,
 * seeds CPU-style at 2*((H-1-y)*W+x), pixels at y*W+x), `currentSample` passed in. */
void ref_render_pass(const Object *objects, unsigned int objectCount, unsigned int lightCount,
                     const Material *materials, const Camera *cam, const Scene *sc, int width,
                     int height, unsigned int *seeds, vec *colors, unsigned int *pixels,
                     int currentSample, int integrator) {
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            const int slot = (height - y - 1) * width + x;
            unsigned int *s0 = &seeds[2 * slot], *s1 = &seeds[2 * slot + 1];
            const Ray ray = harness_camera_ray(cam, x, y, width, height, s0, s1);
            vec radiance;
            ref_radiance(objects, objectCount, lightCount, &ray, s0, s1, sc, materials, integrator,
                         &radiance);
            pixels[y * width + x] =
                harness_accumulate(&colors[slot], &radiance, currentSample, sc->gammaCorrection);
        }
    }
}
