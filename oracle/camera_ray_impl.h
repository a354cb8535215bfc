/* camera_ray_impl.h - TEST INFRASTRUCTURE (see pt_oracle.h).  Primary ray of pixel (x, y),
 * main-cpu.c:57-82 (same math as GenerateCameraRay, path-tracing.cl:9-47): two jitter draws, the
 * un-normalised direction x*kcx + y*kcy + dir, origin = orig + 0.1*that, then normalise. */
static inline void FN(camera_ray)(const pt_camera *cam, int x, int y, int width, int height,
                                  uint32_t *s0, uint32_t *s1, pt_ray *out CTR_PARAM) {
    const float inv_w = 1.f / width;
    const float inv_h = 1.f / height;
    const float j1 = FN(draw)(s0, s1 CTR_ARG) - .5f;
    const float j2 = FN(draw)(s0, s1 CTR_ARG) - .5f;
    const float kcx = (x + j1) * inv_w - .5f;
    const float kcy = (y + j2) * inv_h - .5f;
    v3 dir = v3_make(cam->x.x * kcx + cam->y.x * kcy + cam->dir.x,
                     cam->x.y * kcx + cam->y.y * kcy + cam->dir.y,
                     cam->x.z * kcx + cam->y.z * kcy + cam->dir.z);
    v3 org = v3_scale(0.1f, dir);
    org = v3_add(org, cam->orig);
    out->o = org;
    out->d = v3_normalize_promoted(dir);
}
