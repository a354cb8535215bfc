/*
 * radiance_oracle_impl.h - TEST INFRASTRUCTURE (see pt_oracle.h).
 *
 * The body of the CPU restatement.  It is included twice by radiance_oracle.c: once plain and
 * once with ORACLE_COUNTING defined, which adds the work counters of SURVEY.md 8-d without
 * touching the arithmetic.  FN(x) names the instance's functions, TALLY(field) bumps a counter.
 *
 * Rules followed so that gcc (no -ffast-math, SSE scalar float, no FMA on baseline x86-64)
 * produces the same bits as its build of the reference:
 *   - every expression keeps the reference's association and its float/double promotions
 *     (the reference calls sqrt/cos/sin/fabs/pow from <math.h> on float arguments, so those run
 *     in double) - the promotions are written out explicitly here;
 *   - values the reference funnels through `float k = (...)` in vsmul (vec.h:18) are rounded to
 *     float at the same point;
 *   - argument evaluation order where it is observable (two GetRandom() calls as arguments,
 *     geomfunc.h:239, 259) is gcc's: the LAST argument is evaluated first, so the first draw
 *     lands in the second parameter (confirmed by tests/test_oracle_vs_ref.py).
 */

/* ---- primitive tests ------------------------------------------------------------------- */

/* geomfunc.h:10-39.  Returns the hit distance or 0. */
static inline float FN(hit_sphere)(const pt_object *s, const pt_ray *r CTR_PARAM) {
    TALLY(sphere_tests);
    const v3 op = v3_sub(s->center, r->o);
    const float b = v3_dot(op, r->d);
    float det = b * b - v3_dot(op, op) + s->radius * s->radius;
    if (det < 0.f) return 0.f;
    det = (float)sqrt((double)det);
    float t = b - det;
    if (t > PT_EPSILON) return t;
    t = b + det;
    if (t > PT_EPSILON) return t;
    return 0.f;
}

/* Did hit_sphere answer with the FAR root (b + sqrt(det)) - what it does exactly when the ray starts inside the
 * sphere and does not leave it within EPSILON?  The same float operations as hit_sphere.  (Replay of the product's
 * ENCLOSED sets only, oracle_set_enclosing: a path vertex found by a ray for which some dropped sphere did NOT
 * answer that way may lie outside that sphere.) */
static inline int FN(sphere_far_root)(const pt_object *s, const pt_ray *r) {
    const v3 op = v3_sub(s->center, r->o);
    const float b = v3_dot(op, r->d);
    float det = b * b - v3_dot(op, op) + s->radius * s->radius;
    if (det < 0.f) return 0;
    det = (float)sqrt((double)det);
    if (b - det > PT_EPSILON) return 0;
    return (b + det > PT_EPSILON) ? 1 : 0;
}

/* geomfunc.h:45-78 (Moeller-Trumbore in pbrt's form, two-sided). */
static inline float FN(hit_triangle)(const pt_object *tri, const pt_ray *r CTR_PARAM) {
    TALLY(triangle_tests);
    const v3 e1 = v3_sub(tri->p2, tri->p1);
    const v3 e2 = v3_sub(tri->p3, tri->p1);
    const v3 s1 = v3_cross(r->d, e2);
    const float divisor = v3_dot(s1, e1);
    if (divisor == 0) return 0;
    const float inv = 1.f / divisor;

    const v3 d = v3_sub(r->o, tri->p1);
    const float b1 = v3_dot(d, s1) * inv;
    if (b1 < 0. || b1 > 1.) return 0;

    const v3 s2 = v3_cross(d, e1);
    const float b2 = v3_dot(r->d, s2) * inv;
    if (b2 < 0. || b1 + b2 > 1.) return 0;

    const float t = v3_dot(e2, s2) * inv;
    if (t > PT_EPSILON) return t;
    return 0;
}

/* geomfunc.h:126-157: closest hit, strict <, first index wins ties, *id untouched on a miss */
static int FN(closest_hit)(const pt_object *objects, uint32_t n, const pt_ray *ray, float *t,
                           uint32_t *id, int *odd CTR_PARAM) {
    const float inf = (*t) = 1e20f;
    TALLY(extension_rays);
    if (odd) *odd = 0;
    for (uint32_t i = 0; i < n; i++) {
        float dist;
        if (odd && g_enclosing && g_enclosing[i] && objects[i].type == PT_SPHERE && !FN(sphere_far_root)(&objects[i], ray))
            *odd = 1; /* (replay of the ENCLOSED sets: the vertex this ray finds may be outside sphere i) */
        if (objects[i].type == PT_SPHERE)
            dist = FN(hit_sphere)(&objects[i], ray CTR_ARG);
        else if (objects[i].type == PT_TRIANGLE)
            dist = FN(hit_triangle)(&objects[i], ray CTR_ARG);
        else
            continue;
        if ((dist != 0.f) && (dist < *t)) {
            *t = dist;
            *id = i;
        }
    }
    return (*t < inf);
}

/* geomfunc.h:162-193: any hit closer than maxt, early out in index order.
 * `candidates` (NULL upstream) replays the product's shadow-ray culling on the CPU: objects whose
 * byte is 0 are skipped, objects whose byte is 2 (spheres the ray starts inside of) are tested with
 * the ray's point at maxt instead of the ray, objects whose byte is 3 or 4 (spheres) only by a ray whose origin may
 * be outside them (`origin_odd`) - 3: or lies on them (`origin_id` = the object the ray starts on;
 * oracle_set_shadow_candidates). */
static int FN(any_hit)(const pt_object *objects, uint32_t n, const pt_ray *ray,
                       float maxt, const uint8_t *candidates, uint32_t origin_id, int origin_odd CTR_PARAM) {
    TALLY(shadow_rays);
    for (uint32_t i = 0; i < n; i++) {
        float dist;
        if (candidates && !candidates[i]) continue;
        /* the ENCLOSED family's replay (include/ptgpu.h).  A sphere the rule dropped (4), dropped but for the rays
         * that start on it (3, ENCLOSED_SELF) or tests by the ray's end point (2, BALL) relies on the origin being
         * inside it: for a vertex that may not be (`origin_odd`: the ray that found it did not get the far root of
         * every dropped sphere, closest_hit) the ray is tested as upstream tests it. */
        if (candidates && candidates[i] >= 2) {
            const int as_upstream = origin_odd || (candidates[i] == 3 && i == origin_id);
            if (candidates[i] == 2) {
                /* PTGPU_CULL_BALL replay: the ray starts inside this sphere, so it crosses it before
                 * maxt exactly when its point at maxt lies outside (no ray test) */
                const v3 end = v3_add(ray->o, v3_scale(maxt, ray->d));
                const v3 off = v3_sub(end, objects[i].center);
                if (v3_dot(off, off) > objects[i].radius * objects[i].radius) return 1;
            }
            if (!as_upstream) continue;
        }
        if (objects[i].type == PT_SPHERE)
            dist = FN(hit_sphere)(&objects[i], ray CTR_ARG);
        else if (objects[i].type == PT_TRIANGLE)
            dist = FN(hit_triangle)(&objects[i], ray CTR_ARG);
        else
            continue;
        if ((dist != 0.f) && (dist < maxt)) return 1;
    }
    return 0;
}

/* ---- direct lighting ------------------------------------------------------------------- */

/* geomfunc.h:199-306.  Visits every emissive sphere/triangle (in index order, until `lights`
 * of them have been seen), draws two numbers per light BEFORE any rejection test, and adds
 * emission * area * wi * wo / len^2 for each unoccluded light facing the point.  No 1/N, no 1/pi. */
static void FN(direct_light)(const pt_object *objects, uint32_t n, uint32_t lights, uint32_t *s0,
                             uint32_t *s1, const v3 *hit, const v3 *normal, uint32_t origin_id, int origin_odd,
                             v3 *result CTR_PARAM) {
    *result = v3_make(0.f, 0.f, 0.f);
    uint32_t seen = 0;
    for (uint32_t i = 0; i < n && seen < lights; i++) {
        const pt_object *light = &objects[i];
        if (light->type == PT_MODEL) continue;
        if (v3_is_black(light->emission)) continue;
        seen++;
        TALLY(light_samples);

        pt_ray shadow;
        shadow.o = *hit;
        v3 point, light_normal;

        /* first draw -> second parameter (u2), second draw -> first parameter (u1) */
        const float draw_a = FN(draw)(s0, s1 CTR_ARG);
        const float draw_b = FN(draw)(s0, s1 CTR_ARG);
        const float u1 = draw_b, u2 = draw_a;

        if (light->type == PT_TRIANGLE) {
            /* geomfunc.h:117-121 then :240-251 */
            const float su1 = (float)sqrt((double)u1);
            const float bu = 1.0f - su1;
            const float bv = u2 * su1;
            v3 e1 = v3_sub(light->p2, light->p1);
            v3 e2 = v3_sub(light->p3, light->p1);
            e1 = v3_scale(bu, e1);
            e2 = v3_scale(bv, e2);
            point = light->p1;
            point = v3_add(point, e1);
            point = v3_add(point, e2);
            /* normal from the SCALED edges: a zero barycentric gives 0/0 = NaN (quirk A.3-5) */
            light_normal = v3_cross(e1, e2);
            light_normal = v3_normalize_promoted(light_normal);
        } else {
            /* geomfunc.h:98-110 then :259-264 */
            const float zz = 1.f - 2.f * u1;
            const float rr = (float)sqrt((double)fmax_macro(0.f, 1.f - zz * zz));
            const float phi = 2.f * PI_F * u2;
            const float xx = (float)((double)rr * cos((double)phi));
            const float yy = (float)((double)rr * sin((double)phi));
            const v3 unit = v3_make(xx, yy, zz);
            point = v3_scale(light->radius, unit);
            point = v3_add(point, light->center);
            light_normal = unit;
        }

        shadow.d = v3_sub(point, *hit);
        const float len = (float)sqrt((double)v3_dot(shadow.d, shadow.d));
        shadow.d = v3_scale(1.f / len, shadow.d);

        float wo = v3_dot(shadow.d, light_normal);
        if (wo > 0.f) continue; /* back side of the light */
        /* culled replay only for samples that provably face the light (a NaN normal passes the
         * test above from either side) */
        const uint8_t *candidates =
            (g_shadow_keep && wo <= 0.f) ? g_shadow_keep + (size_t)(seen - 1) * n : NULL;
        wo = -wo;
        wo = clamp_macro(wo, 0.f, 1.f);

        const float wi = v3_dot(shadow.d, *normal);
#ifndef ORACLE_COUNTING
        if (g_mismatch_log && candidates && (wi > 0.f)) { /* diagnostic: culled scan against the full scan */
            const int culled = FN(any_hit)(objects, n, &shadow, len - PT_EPSILON, candidates, origin_id, origin_odd CTR_ARG);
            int full = 0;
            uint32_t first = UINT32_MAX;
            float first_t = 0.f;
            for (uint32_t k = 0; k < n && !full; k++) {
                float dist = 0.f;
                if (objects[k].type == PT_SPHERE) dist = FN(hit_sphere)(&objects[k], &shadow CTR_ARG);
                else if (objects[k].type == PT_TRIANGLE) dist = FN(hit_triangle)(&objects[k], &shadow CTR_ARG);
                if ((dist != 0.f) && (dist < len - PT_EPSILON)) { full = 1; first = k; first_t = dist; }
            }
            if (culled != full && g_mismatch_count < g_mismatch_capacity) {
                float *r = g_mismatch_log + 12 * (size_t)g_mismatch_count++;
                r[0] = shadow.o.x; r[1] = shadow.o.y; r[2] = shadow.o.z;
                r[3] = shadow.d.x; r[4] = shadow.d.y; r[5] = shadow.d.z;
                r[6] = len - PT_EPSILON; r[7] = (float)(seen - 1); r[8] = (float)origin_id;
                r[9] = (float)first; r[10] = first_t; r[11] = (float)culled;
            }
        }
#endif
#ifndef ORACLE_COUNTING
        if (g_ray_log && (wi > 0.f)) { /* answer = 1 + the first occluder of the reference's scan (0: none), id = its distance */
            float first = 0.f, first_t = 0.f;
            for (uint32_t k = 0; k < n && first == 0.f; k++) {
                float dist = 0.f;
                if (objects[k].type == PT_SPHERE) dist = FN(hit_sphere)(&objects[k], &shadow CTR_ARG);
                else if (objects[k].type == PT_TRIANGLE) dist = FN(hit_triangle)(&objects[k], &shadow CTR_ARG);
                if ((dist != 0.f) && (dist < len - PT_EPSILON)) { first = (float)(k + 1); first_t = dist; }
            }
            log_ray(1, &shadow, len - PT_EPSILON, first, first_t);
        }
#endif
        if ((wi > 0.f) && (!FN(any_hit)(objects, n, &shadow, len - PT_EPSILON, candidates, origin_id, origin_odd CTR_ARG))) {
            const float s = light->area * wi * wo / (len * len);
            const v3 contribution = v3_scale(s, light->emission);
            *result = v3_add(*result, contribution);
        }
    }
}

/* geomfunc.h:424-454: cosine-weighted direction about `normal` (named "Uniform" upstream).
 * Two draws, r1 first.  The result is not re-normalised. */
static v3 FN(cosine_direction)(v3 normal, uint32_t *s0, uint32_t *s1 CTR_PARAM) {
    const float r1 = TWO_PI_F * FN(draw)(s0, s1 CTR_ARG);
    const float r2 = FN(draw)(s0, s1 CTR_ARG);
    const float r2s = (float)sqrt((double)r2);

    const v3 w = normal;
    v3 a;
    if (fabs((double)w.x) > .1f)
        a = v3_make(0.f, 1.f, 0.f);
    else
        a = v3_make(1.f, 0.f, 0.f);
    v3 u = v3_cross(a, w);
    u = v3_normalize_promoted(u);
    v3 v = v3_cross(w, u);

    u = v3_scale((float)(cos((double)r1) * (double)r2s), u);
    v = v3_scale((float)(sin((double)r1) * (double)r2s), v);
    v3 out = v3_add(u, v);
    const v3 ws = v3_scale((float)sqrt((double)(1 - r2)), w);
    out = v3_add(out, ws);
    return out;
}

/* Oren-Nayar / Lambert factor of the MATTE branch, geomfunc.h:688-726, evaluated in WORLD space
 * exactly as upstream does (quirk A.3-14).  sigma < 0 leaves `f` uninitialised upstream; the
 * restatement (and the HIP kernels) define it as zero. */
static v3 FN(matte_factor)(const pt_material *m, v3 ray_dir, v3 new_dir) {
    if (m->sigma == 0) return v3_scale(1, m->kd);
    if (!(m->sigma > 0)) return v3_make(0.f, 0.f, 0.f);

    const v3 wo = v3_scale(-1, ray_dir);
    const v3 wi = new_dir;
    const float sinthetai = (float)sqrt((double)fmax_macro(0.f, 1.f - (wi.z * wi.z)));
    const float sinthetao = (float)sqrt((double)fmax_macro(0.f, 1.f - (wo.z * wo.z)));

    float maxcos = 0.f;
    if (sinthetai > 1e-4 && sinthetao > 1e-4) { /* double comparison against 1e-4 */
        const float sinphii = sinthetai == 0 ? 0 : clamp_macro(wi.y / sinthetai, -1.f, 1.f);
        const float cosphii = sinthetai == 0 ? 1 : clamp_macro(wi.x / sinthetai, -1.f, 1.f);
        const float sinphio = sinthetao == 0 ? 0 : clamp_macro(wo.y / sinthetao, -1.f, 1.f);
        const float cosphio = sinthetao == 0 ? 1 : clamp_macro(wo.x / sinthetao, -1.f, 1.f);
        const float dcos = cosphii * cosphio + sinphii * sinphio;
        maxcos = fmax_macro(0.f, dcos);
    }

    float sinalpha, tanbeta;
    if (fabs((double)wi.z) > fabs((double)wo.z)) {
        sinalpha = sinthetao;
        tanbeta = (float)((double)sinthetai / fabs((double)wi.z));
    } else {
        sinalpha = sinthetai;
        tanbeta = (float)((double)sinthetao / fabs((double)wo.z));
    }
    return v3_scale(1 * (m->A + m->B * maxcos * sinalpha * tanbeta), m->kd);
}

/* The dielectric interface of geomfunc.h:626-680 (nc = 1, nt = 1.5, Schlick, roulette between
 * reflection and refraction).  Updates ray and throughput; one draw unless total internal
 * reflection. */
static void FN(dielectric)(pt_ray *ray, v3 hit, v3 normal, v3 nl, v3 color, v3 *throughput,
                           uint32_t *s0, uint32_t *s1 CTR_PARAM) {
    v3 refl_dir = v3_scale(2.f * v3_dot(normal, ray->d), normal);
    refl_dir = v3_sub(ray->d, refl_dir);
    const int into = (v3_dot(normal, nl) > 0);

    const float nc = 1.f, nt = 1.5f;
    const float nnt = into ? nc / nt : nt / nc;
    const float ddn = v3_dot(ray->d, nl);
    const float cos2t = 1.f - nnt * nnt * (1.f - ddn * ddn);

    if (cos2t < 0.f) { /* total internal reflection */
        *throughput = v3_mul(*throughput, color);
        ray->o = hit;
        ray->d = refl_dir;
        return;
    }

    const float kk = (float)((into ? 1 : -1) * ((double)(ddn * nnt) + sqrt((double)cos2t)));
    const v3 nkk = v3_scale(kk, normal);
    v3 trans = v3_scale(nnt, ray->d);
    trans = v3_sub(trans, nkk);
    trans = v3_normalize_promoted(trans);

    const float a = nt - nc;
    const float b = nt + nc;
    const float R0 = a * a / (b * b);
    const float c = 1 - (into ? -ddn : v3_dot(trans, normal));
    const float Re = R0 + (1 - R0) * c * c * c * c * c;
    const float Tr = 1.f - Re;
    const float P = .25f + .5f * Re;
    const float RP = Re / P;
    const float TP = Tr / (1.f - P);

    if (FN(draw)(s0, s1 CTR_ARG) < P) {
        *throughput = v3_scale(RP, *throughput);
        *throughput = v3_mul(*throughput, color);
        ray->o = hit;
        ray->d = refl_dir;
    } else {
        *throughput = v3_scale(TP, *throughput);
        *throughput = v3_mul(*throughput, color);
        ray->o = hit;
        ray->d = trans;
    }
}

/* ---- the two integrators --------------------------------------------------------------- */

/* geomfunc.h:494-798 (integrator 0) and :800-1014 (integrator 1).  They differ in one thing:
 * after direct lighting at a diffuse/matte hit the direct-lighting variant stops, the path tracer
 * continues with a cosine-sampled bounce.  Segments 0..6, no Russian roulette. */
static void FN(radiance)(const pt_object *objects, uint32_t n, uint32_t lights, const pt_ray *start,
                         uint32_t *s0, uint32_t *s1, v3 sky1, v3 sky2, const pt_material *materials,
                         int integrator, v3 *radiance CTR_PARAM) {
    pt_ray ray = *start;
    v3 L = v3_make(0.f, 0.f, 0.f);
    v3 throughput = v3_make(1.f, 1.f, 1.f);
    int specular_bounce = 1;

    for (unsigned depth = 0;; ++depth) {
        if (depth > PT_MAX_DEPTH) break;

        float t;
        uint32_t id = 0;
        int odd = 0;
#ifndef ORACLE_COUNTING
        if (g_ray_log) {
            g_ray_depth = (int)depth;
            float lt;
            uint32_t lid = 0;
            const int found = FN(closest_hit)(objects, n, &ray, &lt, &lid, NULL CTR_ARG);
            log_ray(0, &ray, 1e20f, found ? lt : 0.f, found ? (float)lid : -1.f);
        }
#endif
        if (!FN(closest_hit)(objects, n, &ray, &t, &id, &odd CTR_ARG)) {
            /* miss: throughput * sky, DISCARDING what was gathered so far (quirk A.3-2) */
            const v3 sky_dir = v3_make(-.41f, -.82f, .41f);
            const float pos = (v3_dot(ray.d, sky_dir) + 1) / 2;
            const v3 sky = v3_make(sky1.x * pos + sky2.x * (1 - pos), sky1.y * pos + sky2.y * (1 - pos),
                                   sky1.z * pos + sky2.z * (1 - pos));
            *radiance = v3_mul(throughput, sky);
            return;
        }

        const pt_object *obj = &objects[id];
        v3 hit = v3_scale(t, ray.d);
        hit = v3_add(ray.o, hit);

        v3 normal;
        if (obj->type == PT_SPHERE) {
            normal = v3_sub(hit, obj->center);
            normal = v3_normalize_promoted(normal);
        } else {
            const v3 e1 = v3_sub(obj->p2, obj->p1);
            const v3 e2 = v3_sub(obj->p3, obj->p1);
            normal = v3_cross(e1, e2);
        }
        normal = v3_normalize_promoted(normal);

        const float dp = v3_dot(normal, ray.d);
        const float flip = -1.f * ((dp) > 0 ? 1 : -1); /* sign(0) = -1 upstream (vec.h:32) */
        const v3 nl = v3_scale(flip, normal);

        if (!v3_is_black(obj->emission)) {
            if (specular_bounce) {
                v3 e = v3_scale((float)fabs((double)dp), obj->emission);
                e = v3_mul(throughput, e);
                L = v3_add(L, e);
            }
            break; /* emitters end the path */
        }

        const pt_material *m = &materials[obj->material_id];
        if (m->type == PT_LAMBERTIAN) {
            specular_bounce = 0;
            throughput = v3_mul(throughput, obj->color);
            v3 Ld;
            FN(direct_light)(objects, n, lights, s0, s1, &hit, &nl, id, odd, &Ld CTR_ARG);
            Ld = v3_mul(throughput, Ld);
            L = v3_add(L, Ld);
            if (integrator == PT_INTEGRATOR_DL) break;
            ray.o = hit;
            ray.d = FN(cosine_direction)(nl, s0, s1 CTR_ARG);
        } else if (m->type == PT_CONDUCTOR) {
            specular_bounce = 1;
            v3 refl = v3_scale(2.f * v3_dot(normal, ray.d), normal);
            refl = v3_sub(ray.d, refl);
            throughput = v3_mul(throughput, obj->color);
            ray.o = hit;
            ray.d = refl;
        } else if (m->type == PT_DIELECTRIC) {
            specular_bounce = 1;
            FN(dielectric)(&ray, hit, normal, nl, obj->color, &throughput, s0, s1 CTR_ARG);
        } else if (m->type == PT_MATTE) {
            specular_bounce = 0;
            const v3 new_dir = FN(cosine_direction)(nl, s0, s1 CTR_ARG); /* drawn BEFORE the lights */
            const v3 f = FN(matte_factor)(m, ray.d, new_dir);
            throughput = v3_mul(throughput, f);
            v3 Ld;
            FN(direct_light)(objects, n, lights, s0, s1, &hit, &nl, id, odd, &Ld CTR_ARG);
            Ld = v3_mul(throughput, Ld);
            L = v3_add(L, Ld);
            if (integrator == PT_INTEGRATOR_DL) break;
            ray.o = hit;
            ray.d = new_dir;
        }
        /* PLASTIC / METAL: upstream has no live branch (quirk A.3-15): the same ray is traced
         * again until the depth limit, then L is returned. */
    }
    *radiance = L;
}

/* main-cpu.c:54-110 for the rows [y0, y1). */
static void FN(render_rows)(const pt_object *objects, uint32_t n, uint32_t lights,
                            const pt_material *materials, const pt_camera *cam,
                            const pt_scene *scene, int width, int height, uint32_t *seeds,
                            pt_vec3 *colors, uint32_t *pixels, int current_sample, int integrator,
                            int seed_layout, int y0, int y1 CTR_PARAM) {
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < width; x++) {
            const int slot = (height - y - 1) * width + x;
            const int sslot = seed_layout == ORACLE_SEEDS_KERNEL_LAYOUT ? y * width + x : slot;
            uint32_t *s0 = &seeds[2 * sslot], *s1 = &seeds[2 * sslot + 1];
            pt_ray ray;
            FN(camera_ray)(cam, x, y, width, height, s0, s1, &ray CTR_ARG);
#ifndef ORACLE_COUNTING
            if (g_ray_log) g_ray_pixel = y * width + x; /* (serial renders only; unconditionally, every thread would write this line) */
#endif
            v3 sample;
            FN(radiance)(objects, n, lights, &ray, s0, s1, scene->sky1, scene->sky2, materials,
                         integrator, &sample CTR_ARG);
            TALLY(paths);
            pixels[y * width + x] = oracle_accumulate(&colors[slot], &sample, current_sample, scene->gamma);
        }
    }
}
