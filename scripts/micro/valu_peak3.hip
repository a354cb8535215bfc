// Micro-benchmark 3: sustained issue rate (event timing, all CUs busy) of v_fma_f32 vs v_pk_fma_f32
// vs v_pk_mul_f32 with op_sel broadcast, pinned by inline asm.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void spin(float *out, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float a = 1.0001f + x0 * 1e-9f, b = 1e-4f;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6}, pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        } else if (MODE == 1) {
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));)
        } else if (MODE == 2) { // broadcast the low half of the second source to both halves
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %8 op_sel_hi:[1,0]\n"
                              "v_pk_mul_f32 %4, %4, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %5, %5, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %6, %6, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %7, %8 op_sel_hi:[1,0]"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));)
        } else { // mixed: 4 pk_fma + 2 fma + cmp + cndmask
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_fma_f32 %4, %4, %10, %11\n v_pk_fma_f32 %2, %2, %8, %9\n"
                              "v_cmp_lt_f32 vcc, %4, %5\n v_pk_fma_f32 %3, %3, %8, %9\n v_cndmask_b32 %6, %6, %7, vcc\n v_fma_f32 %5, %5, %10, %11"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(pa), "v"(pb), "v"(a), "v"(b) : "vcc");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int MODE>
double run(int blocks, int iters, float *d_out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float *d_out; hipMalloc(&d_out, sizeof(float) * 256 * 256 * 64);
    const int iters = 20000;
    for (int per_cu : {1, 2, 4, 8}) {
        const int blocks = 256 * per_cu;
        const double wi = (double)blocks * 4 * iters * 64;
        double m0 = run<0>(blocks, iters, d_out), m1 = run<1>(blocks, iters, d_out), m2 = run<2>(blocks, iters, d_out), m3 = run<3>(blocks, iters, d_out);
        printf("waves/SIMD %d: v_fma %.3f  v_pk_fma %.3f  v_pk_mul(op_sel) %.3f  mixed %.3f  G wave-instr/s/SIMD\n", per_cu,
               wi / m0 / 1e6 / 1024, wi / m1 / 1e6 / 1024, wi / m2 / 1e6 / 1024, wi / m3 / 1e6 / 1024);
    }
    return 0;
}
