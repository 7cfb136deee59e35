// mfma_clock.hip - calibration: what one v_mfma_f32_32x32x16_bf16 costs in s_memtime ticks and in wall time when every SIMD of
// every CU issues them back to back (4 independent accumulators per wave), i.e. the unit the cycle stamps of the node kernels are in.
//   hipcc -O3 --offload-arch=gfx950 mfma_clock.hip -o mfma_clock && ./mfma_clock [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// NA independent accumulators per wave (dependent MFMAs on one accumulator are NA issues apart)
template <int NA>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink, int iters) {
    f32x16 acc[NA];
    for (int i = 0; i < NA; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24 / NA; ++u)
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NA; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = t1 - t0; out[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = r1 - r0; }
}
int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, iters = 20000;
    unsigned long long* out; float* sink;
    hipMalloc(&out, wgs * 8 * sizeof(unsigned long long)); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        const int na = rep < 2 ? 4 : rep < 4 ? 2 : 1;
        hipEventRecord(e0);
        if (na == 4) hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(256), 0, 0, out, sink, iters);
        else if (na == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, out, sink, iters);
        else hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, out, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        const double n = 24.0 * iters;
        printf("%d accumulator(s) per wave, %d workgroups x 4 waves: %.1f s_memtime ticks per MFMA, %.2f ns per MFMA by s_memrealtime (100 MHz), %.2f ns by events -> "
               "%.0f MHz if an MFMA is 32 cycles; s_memtime rate %.0f MHz; %.1f TFLOP/s\n", na, wgs, h[0] / n, h[1] * 10.0 / n, ms * 1e6 / n,
               32.0 / (h[1] * 10.0 / n) * 1e3, h[0] / (h[1] * 10.0) * 1e3, wgs * 4 * n * 32768.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
