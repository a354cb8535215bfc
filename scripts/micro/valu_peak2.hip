// Micro-benchmark 2: cycles per wave64 VALU instruction on one SIMD, by instruction form, from
// s_memtime inside the kernel (clock independent).  Inline asm pins the exact opcodes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void spin(unsigned long long *cycles, float *out, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float a = 1.0001f + x0 * 1e-9f, b = 1e-4f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // VOP3 v_fma_f32, 8 independent
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        } else if (MODE == 1) { // VOP2 v_fmac_f32
            REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        } else if (MODE == 2) { // VOP2 v_mul_f32
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        } else if (MODE == 3) { // v_pk_fma_f32, 4 independent pairs x2
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));)
        } else if (MODE == 4) { // v_cmp + v_cndmask pairs
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                              "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");)
        } else if (MODE == 5) { // v_rcp_f32
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y;
}

template <int MODE>
void run(const char *name, unsigned long long *d_c, float *d_out) {
    const int iters = 4000;
    printf("%-18s", name);
    for (int waves : {1, 2, 4, 8}) { // waves per SIMD: blocks of 64*4*waves threads... use 256-thread blocks, `waves` per CU
        hipLaunchKernelGGL(spin<MODE>, dim3(256 * waves), dim3(256), 0, 0, d_c, d_out, iters);
        hipDeviceSynchronize();
        unsigned long long c;
        hipMemcpy(&c, d_c, sizeof c, hipMemcpyDeviceToHost);
        // the stamped wave issued iters*64 instructions while sharing its SIMD with `waves`-1 others
        printf("  %dw: %.2f cyc/inst/SIMD", waves, (double)c / ((double)iters * 64 * waves));
    }
    printf("\n");
}

int main() {
    unsigned long long *d_c; float *d_out;
    hipMalloc(&d_c, 8); hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8);
    run<0>("v_fma_f32 (VOP3)", d_c, d_out);
    run<1>("v_fmac_f32 (VOP2)", d_c, d_out);
    run<2>("v_mul_f32 (VOP2)", d_c, d_out);
    run<3>("v_pk_fma_f32", d_c, d_out);
    run<4>("v_cmp+v_cndmask", d_c, d_out);
    run<5>("v_rcp_f32", d_c, d_out);
    return 0;
}
