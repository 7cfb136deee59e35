// VALU issue-rate microbenchmark (gfx950): scalar fp32 FMA vs packed v_pk_fma_f32 vs v_exp_f32 / v_rcp_f32, one to four
// waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/valu_rate_test.cpp -o /tmp/valu_rate_test && /tmp/valu_rate_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = 1.0001f, c = 1e-4f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {            // 8 scalar FMAs (8 elements)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (MODE == 1) {     // 4 packed FMAs (8 elements)
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (MODE == 2) {     // 8 v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 3) {     // 8 v_rcp_f32
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 4) {     // 4 packed mul (8 elements)
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b};
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> void run(const char* name, int waves_per_simd) {
    float* out; hipMalloc(&out, 256 * 1024 * 64 * sizeof(float));
    const int iters = 20000, blocks = 256 * waves_per_simd;      // 256 threads = 4 waves per block = 1 per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // cycles per 8-element group per wave at 2.4 GHz (upper bound on the clock)
    printf("%-14s waves/SIMD %d: %.3f ms -> %.1f cycles per 8 elements per wave (if 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / iters / waves_per_simd);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("8x v_fma", w); run<1>("4x v_pk_fma", w); run<4>("4x v_pk_mul", w); run<2>("8x v_exp", w); run<3>("8x v_rcp", w);
    }
    return 0;
}
