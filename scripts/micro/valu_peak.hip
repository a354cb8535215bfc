// Micro-benchmark: sustained wave64 VALU issue rate on gfx950 for plain v_fma_f32, v_mul_f32 and a
// mix with v_cmp/v_cndmask, at several waves per SIMD.  Calibrates the VALU roofline used in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void spin(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) { // 8 independent FMAs
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
            } else if (MODE == 1) { // 8 independent MULs
                x0 *= a; x1 *= a; x2 *= a; x3 *= a; x4 *= a; x5 *= a; x6 *= a; x7 *= a;
            } else { // 4 fma + 2 mul + cmp + cndmask (8 VALU)
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 *= a; x5 *= a;
                x6 = (x0 < x7) ? x6 + a : x6;
                x7 += b;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
double run(int blocks, int iters, float *d_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0001f, 0.0001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0001f, 0.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *d_out; hipMalloc(&d_out, sizeof(float) * 256 * 256 * 64);
    const int iters = 20000;
    for (int per_cu : {1, 2, 4, 8}) { // 256-thread blocks per CU -> waves per SIMD
        const int blocks = 256 * per_cu;
        const double wave_instr = (double)blocks * 4 * iters * 64; // 64 VALU per iteration per wave
        double m0 = run<0>(blocks, iters, d_out), m1 = run<1>(blocks, iters, d_out), m2 = run<2>(blocks, iters, d_out);
        printf("waves/SIMD %d: fma %.3f  mul %.3f  mix %.3f  G wave-instr/s/SIMD  (fma => %.1f TFLOP/s)\n", per_cu,
               wave_instr / m0 / 1e6 / 1024, wave_instr / m1 / 1e6 / 1024, wave_instr / m2 / 1e6 / 1024,
               wave_instr * 128 / m0 / 1e9);
    }
    return 0;
}
