/*
 * pt_enclosed.h - "well inside a sphere" for the ENCLOSED family of shadow-ray candidate sets (include/ptgpu.h,
 * PTGPU_CULL_ENCLOSED / _BALL / _ENCLOSED_SELF).  One definition, shared by the library's host code, its kernels
 * (the values travel in the sphere records) and the test oracle's replay of the sets, so that the three agree bit
 * for bit.  No reference counterpart: the reference scans every object (geomfunc.h:162-193).
 */
#ifndef PT_ENCLOSED_H
#define PT_ENCLOSED_H

/* "Well inside" for the ENCLOSED family: a point x is well inside the sphere (centre c, radius r) when
 * |x - c|^2, evaluated in float as dx*dx + dy*dy + dz*dz without contraction, is below this limit: the square of
 * r - 2e-6 (r + |cx| + |cy| + |cz|) - for the r = 1e4 walls 0.04 below the surface, sixteen times the depth (2.4e-3:
 * 1e-3 of rounding in c - x plus 28 units of the 1e8-sized terms over 2 r) at which the float discriminant
 * b*b - op.op + r*r can still round to a hit for a ray leaving inwards.
 * 0 = no usable limit (the sphere is then never dropped).  Shared by the library, the kernels (the value travels in
 * the sphere's record) and the test oracle's replay. */
static inline float pt_enclosed_inside_limit(float r, float cx, float cy, float cz) {
    const double ar = r < 0 ? -(double)r : (double)r;
    const double reach = ar + (cx < 0 ? -(double)cx : (double)cx) + (cy < 0 ? -(double)cy : (double)cy) +
                         (cz < 0 ? -(double)cz : (double)cz);
    const double q = ar - 2e-6 * reach;
    const float lim = (float)(q * q);
    return (q > 0.0 && lim > 0.f && lim < 3e38f) ? lim : 0.f;
}
/* The kernels' cheap first look at a vertex (never the decision): with out = (-2c, |c|^2 - limit + pad),
 * pad = 1e-6 (|c| + r)^2 far above the rounding of three multiply-adds on these magnitudes, a point x with
 * out.xyz . x + out.w < -|x|^2 is well inside for certain; every other vertex gets the exact test above. */
static inline void pt_enclosed_prefilter(float r, float cx, float cy, float cz, float limit, float out[4]) {
    const double ar = r < 0 ? -(double)r : (double)r;
    const double cc = (double)cx * cx + (double)cy * cy + (double)cz * cz;
    const double reach = ar + (cc > 0 ? __builtin_sqrt(cc) : 0.0);
    out[0] = -2.f * cx;
    out[1] = -2.f * cy;
    out[2] = -2.f * cz;
    out[3] = (float)(cc - (double)limit + 1e-6 * reach * reach);
}

#endif /* PT_ENCLOSED_H */
