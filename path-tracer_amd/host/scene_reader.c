/*
 * scene_reader.c - the text-scene grammar of the reference (readScene, displayfunc.c:103-406),
 * restated as a library call that fills a pthost_scene instead of globals.
 *
 * Grammar (whitespace is free-form exactly as with the reference's fscanf formats):
 *   [gamma G  sky1 r g b  sky2 r g b]             optional; default 2.2 / (.15,.3,.5) / (1,1,1)
 *   camera ex ey ez  tx ty tz                      target is parsed and never used (camera.c:9-15)
 *   materials N
 *     matte kr kg kb sigma | plastic kd(3) ks(3) roughness      N descriptors -> indices 3..3+N-1
 *   objects M
 *     sphere r  cx cy cz  er eg eb  r g b  mat
 *     triangle p1(3) p2(3) p3(3)  e(3)  c(3)  mat
 *     model path  pos(3) scale(3)  e(3) c(3)  mat
 *   (anything after the M-th object line is ignored)
 *
 * Quirks kept: three built-in materials occupy indices 0..2; an object's `mat`
 * integer 0..5 is stored as materialId and later used directly as the material index; an object is
 * a light iff !(e.x == 0 && e.x == 0 && e.z == 0) (the reference's viszero tests x twice, vec.h:23);
 * a `model` line leaves a PT_MODEL placeholder followed by one triangle per `f` line of the OBJ,
 * each inheriting emission/colour/material, so an emissive model makes every face a light.
 */
#include <libgen.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pthost.h"
#include "pthost_internal.h"

#define PI_F 3.14159265358979323846f
#define N_BUILTIN_MATERIALS 3
#define PTHOST_MAX_USER_MATERIALS (1u << 20)
#define PTHOST_MAX_OBJECTS (1u << 24)

static __thread char g_error[512];

void pthost_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}

const char *pthost_last_error(void) { return g_error; }

static int emits(const pt_vec3 *e) { return !((e->x == 0.f) && (e->x == 0.f) && (e->z == 0.f)); }

static int read_floats(FILE *f, float *dst, int n) {
    int got = 0;
    while (got < n && fscanf(f, "%f", &dst[got]) == 1) ++got;
    return got;
}

static int material_index_from_id(unsigned id, int16_t *out) {
    if (id > PT_METAL) return -1; /* getMaterialType's default branch, displayfunc.c:94-96 */
    *out = (int16_t)id;
    return 0;
}

static float triangle_area(const pt_object *t) {
    /* displayfunc.c:267-270: area = norm(e1 x e2) * 0.5f, norm() through double sqrt */
    float e1x = t->p2.x - t->p1.x, e1y = t->p2.y - t->p1.y, e1z = t->p2.z - t->p1.z;
    float e2x = t->p3.x - t->p1.x, e2y = t->p3.y - t->p1.y, e2z = t->p3.z - t->p1.z;
    float nx = e1y * e2z - e1z * e2y;
    float ny = e1z * e2x - e1x * e2z;
    float nz = e1x * e2y - e1y * e2x;
    return (float)(sqrt((double)(nx * nx + ny * ny + nz * nz)) * 0.5f);
}

static void title_from_path(char *title, size_t cap, const char *path) {
    char tmp[512];
    snprintf(tmp, sizeof tmp, "%s", path);
    const char *base = basename(tmp);
    snprintf(title, cap, "%s", base);
    char *end = title + strlen(title);
    while (end > title && *end != '.' && *end != '\\' && *end != '/') --end;
    if (end > title && *end == '.') *end = '\0';
}

void pthost_free_scene(pthost_scene *s) {
    if (!s) return;
    free(s->objects);
    free(s->materials);
    s->objects = NULL;
    s->materials = NULL;
    s->object_count = s->light_count = 0;
}

static int expand_model(pthost_scene *out, uint32_t declared, uint32_t slot, uint32_t *added,
                        const char *model_path, const char *model_dir, pt_vec3 pos, pt_vec3 scale) {
    char full[1024];
    if (model_dir && model_path[0] != '/')
        snprintf(full, sizeof full, "%s/%s", model_dir, model_path);
    else
        snprintf(full, sizeof full, "%s", model_path);

    float *verts = NULL;
    int32_t *corners = NULL;
    uint32_t nverts = 0, ncorners = 0, nfaces = 0;
    if (pthost_obj_load(full, &verts, &nverts, &corners, &ncorners, &nfaces) != 0) return -1;

    /* displayfunc.c:329: grow to declared + already added + this model's faces */
    size_t new_cap = (size_t)declared + *added + nfaces;
    pt_object *grown = (pt_object *)realloc(out->objects, sizeof(pt_object) * (new_cap ? new_cap : 1));
    if (!grown) {
        free(verts);
        free(corners);
        pthost_set_error("out of memory growing object table to %zu", new_cap);
        return -1;
    }
    out->objects = grown;
    const pt_object proto = out->objects[slot];

    /* the reference steps 3 corners per face whatever the face's real corner count
     * (displayfunc.c:342-344, 383) */
    uint32_t off = 0;
    for (uint32_t fi = 0; fi < nfaces; ++fi, off += 3) {
        if (off + 2 >= ncorners) {
            free(verts);
            free(corners);
            pthost_set_error("model '%s': face %u reads past the corner list", model_path, fi);
            return -1;
        }
        pt_object *tri = &out->objects[slot + 1 + fi];
        memset(tri, 0, sizeof *tri);
        tri->type = PT_TRIANGLE;
        tri->emission = proto.emission;
        tri->color = proto.color;
        tri->material_id = proto.material_id;
        pt_vec3 *dst[3] = {&tri->p1, &tri->p2, &tri->p3};
        for (int c = 0; c < 3; ++c) {
            int32_t vi = corners[off + c];
            if (vi < 0 || (uint32_t)vi >= nverts) {
                free(verts);
                free(corners);
                pthost_set_error("model '%s': vertex index %d out of range", model_path, vi);
                return -1;
            }
            pt_vec3 p = {verts[3 * vi + 0], verts[3 * vi + 1], verts[3 * vi + 2]};
            p.x = scale.x * p.x;
            p.y = scale.y * p.y;
            p.z = scale.z * p.z;
            p.x = p.x + pos.x;
            p.y = p.y + pos.y;
            p.z = p.z + pos.z;
            *dst[c] = p;
        }
        tri->area = triangle_area(tri);
        if (emits(&tri->emission)) out->light_count++;
    }
    *added += nfaces;
    free(verts);
    free(corners);
    return 0;
}

int pthost_load_scene(const char *path, const char *model_dir, int width, int height,
                      pthost_scene *out) {
    if (!path || !out) {
        pthost_set_error("pthost_load_scene: null argument");
        return -1;
    }
    memset(out, 0, sizeof *out);
    FILE *f = fopen(path, "r");
    if (!f) {
        pthost_set_error("Failed to open file: %s", path);
        return -1;
    }
    title_from_path(out->title, sizeof out->title, path);

    /* part 1: optional gamma / sky line (displayfunc.c:117-124) */
    int c = fscanf(f, "gamma %f sky1 %f %f %f sky2 %f %f %f ", &out->scene.gamma, &out->scene.sky1.x,
                   &out->scene.sky1.y, &out->scene.sky1.z, &out->scene.sky2.x, &out->scene.sky2.y,
                   &out->scene.sky2.z);
    if (c != 7) {
        out->scene.gamma = 2.2f;
        out->scene.sky1 = (pt_vec3){0.15f, 0.3f, 0.5f};
        out->scene.sky2 = (pt_vec3){1.f, 1.f, 1.f};
    }

    /* part 2: camera (displayfunc.c:127-139) */
    c = fscanf(f, "camera %f %f %f %f %f %f ", &out->camera.orig.x, &out->camera.orig.y,
               &out->camera.orig.z, &out->camera.target.x, &out->camera.target.y,
               &out->camera.target.z);
    out->camera.pitch = 0;
    out->camera.yaw = 0;
    out->camera.width = (float)width;
    out->camera.height = (float)height;
    if (c != 6) {
        pthost_set_error("Failed to read 6 camera parameters: %d", c);
        goto fail;
    }

    /* part 3/4: materials (displayfunc.c:142-194) */
    unsigned user_materials = 0;
    c = fscanf(f, "materials %u ", &user_materials);
    if (c < 1) {
        pthost_set_error("Failed to read the number of materials: %d", c);
        goto fail;
    }
    /* upstream adds without a bound (displayfunc.c:147): the sum wraps for counts near 2^32, the
     * table comes out shorter than the three built-ins and they are written past its end */
    if (user_materials > PTHOST_MAX_USER_MATERIALS) {
        pthost_set_error("Too many materials: %u (at most %u)", user_materials, PTHOST_MAX_USER_MATERIALS);
        goto fail;
    }
    out->scene.material_count = user_materials + N_BUILTIN_MATERIALS;
    out->materials = (pt_material *)calloc(out->scene.material_count, sizeof(pt_material));
    if (!out->materials) {
        pthost_set_error("out of memory for %u materials", out->scene.material_count);
        goto fail;
    }
    out->materials[0].type = PT_LAMBERTIAN;
    out->materials[1].type = PT_CONDUCTOR;
    out->materials[2].type = PT_DIELECTRIC;

    for (unsigned mi = N_BUILTIN_MATERIALS; mi < out->scene.material_count; ++mi) {
        pt_material *m = &out->materials[mi];
        char kind[100];
        int got = fscanf(f, "%99s", kind);
        if (got != 1) {
            pthost_set_error("Failed to read material description type. Expected 1, found %d value(s)", got);
            goto fail;
        }
        if (strcmp(kind, "matte") == 0) {
            float v[4];
            m->type = PT_MATTE;
            got = read_floats(f, v, 4);
            if (got != 4) {
                pthost_set_error("Failed to read matte material descriptor. Expected 4, found %d value(s)", got);
                goto fail;
            }
            m->kd = (pt_vec3){v[0], v[1], v[2]};
            m->sigma = v[3]; /* the reference's clamp(sigma, 0, .9) is a discarded expression (:177) */
            float s2 = m->sigma * m->sigma;
            m->A = 1.f - (s2 / (2.f * (s2 + 0.33f)));
            m->B = 0.45f * s2 / (s2 + 0.09f);
            if (fscanf(f, " ") == EOF) { /* the reference's format ends in "\n": skip white space */ }
        } else if (strcmp(kind, "plastic") == 0) {
            float v[7];
            m->type = PT_PLASTIC;
            got = read_floats(f, v, 7);
            if (got != 7) {
                pthost_set_error("Failed to read plastic material descriptor. Expected 4, found %d value(s)", got);
                goto fail;
            }
            m->kd = (pt_vec3){v[0], v[1], v[2]};
            m->ks = (pt_vec3){v[3], v[4], v[5]};
            m->roughness = v[6];
            if (fscanf(f, " ") == EOF) { /* as above */ }
        }
        /* any other word consumes this material slot and nothing else (displayfunc.c:168-193) */
    }

    /* part 5/6: objects (displayfunc.c:197-391) */
    unsigned declared = 0;
    c = fscanf(f, "objects %u ", &declared);
    if (c < 1) {
        pthost_set_error("Failed to read object count: %d", c);
        goto fail;
    }
    if (declared > PTHOST_MAX_OBJECTS) { /* what ptgpu_set_scene takes (2^24) */
        pthost_set_error("Too many objects: %u (at most %u)", declared, PTHOST_MAX_OBJECTS);
        goto fail;
    }
    out->objects = (pt_object *)calloc(declared ? declared : 1, sizeof(pt_object));
    if (!out->objects) {
        pthost_set_error("out of memory for %u objects", declared);
        goto fail;
    }
    uint32_t added = 0;
    out->light_count = 0;
    for (unsigned li = 0; li < declared; ++li) {
        uint32_t slot = li + added;
        pt_object *obj = &out->objects[slot];
        memset(obj, 0, sizeof *obj); /* the table may have been realloc()ed: no stale bytes */
        char kind[100];
        unsigned mat = 0;
        int got = fscanf(f, "%99s", kind);
        if (got != 1) {
            pthost_set_error("Failed to read object description type. Expected 1, found %d value(s)", got);
            goto fail;
        }
        if (strcmp(kind, "sphere") == 0) {
            float v[10];
            obj->type = PT_SPHERE;
            got = read_floats(f, v, 10);
            if (got == 10) got += fscanf(f, "%u ", &mat) == 1;
            if (got != 11) {
                pthost_set_error("Failed to read sphere #%u: %d", li, got);
                goto fail;
            }
            obj->radius = v[0];
            obj->center = (pt_vec3){v[1], v[2], v[3]};
            obj->emission = (pt_vec3){v[4], v[5], v[6]};
            obj->color = (pt_vec3){v[7], v[8], v[9]};
            if (material_index_from_id(mat, &obj->material_id)) {
                pthost_set_error("Failed to read material type for sphere #%u: %u", li, mat);
                goto fail;
            }
            obj->area = 4 * PI_F * obj->radius * obj->radius;
            if (emits(&obj->emission)) out->light_count++;
        } else if (strcmp(kind, "triangle") == 0) {
            float v[15];
            obj->type = PT_TRIANGLE;
            got = read_floats(f, v, 15);
            if (got == 15) got += fscanf(f, "%u ", &mat) == 1;
            if (got != 16) {
                pthost_set_error("Failed to read triangle #%u: %d", li, got);
                goto fail;
            }
            obj->p1 = (pt_vec3){v[0], v[1], v[2]};
            obj->p2 = (pt_vec3){v[3], v[4], v[5]};
            obj->p3 = (pt_vec3){v[6], v[7], v[8]};
            obj->emission = (pt_vec3){v[9], v[10], v[11]};
            obj->color = (pt_vec3){v[12], v[13], v[14]};
            if (material_index_from_id(mat, &obj->material_id)) {
                pthost_set_error("Failed to read material type for sphere #%u: %u", li, mat);
                goto fail;
            }
            obj->area = triangle_area(obj);
            if (emits(&obj->emission)) out->light_count++;
        } else if (strcmp(kind, "model") == 0) {
            char model_path[200];
            float v[12];
            obj->type = PT_MODEL;
            got = fscanf(f, "%199s", model_path) == 1;
            if (got == 1) got += read_floats(f, v, 12);
            if (got == 13) got += fscanf(f, "%u ", &mat) == 1;
            if (got != 14) {
                pthost_set_error("Failed to read model #%u: %d", li, got);
                goto fail;
            }
            pt_vec3 pos = {v[0], v[1], v[2]};
            pt_vec3 scale = {v[3], v[4], v[5]};
            obj->emission = (pt_vec3){v[6], v[7], v[8]};
            obj->color = (pt_vec3){v[9], v[10], v[11]};
            if (material_index_from_id(mat, &obj->material_id)) {
                pthost_set_error("Failed to read material type for sphere #%u: %u", li, mat);
                goto fail;
            }
            if (expand_model(out, declared, slot, &added, model_path, model_dir, pos, scale) != 0)
                goto fail;
        }
        else {
            /* any other word consumes this object slot and nothing else (displayfunc.c:220-387);
             * the reference leaves the slot as uninitialised heap, we make it an inert placeholder */
            obj->type = PT_MODEL;
        }
    }
    out->object_count = declared + added;
    fclose(f);
    return 0;

fail:
    fclose(f);
    pthost_free_scene(out);
    return -1;
}
