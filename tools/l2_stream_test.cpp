// L2 streaming microbenchmark: G workgroups (256 threads) each read the same S-byte buffer once, 16 B per lane, U loads in
// flight per thread.  mode 0: every workgroup walks it in the same order (what k_node's weight stream does);
// mode 1: workgroup b starts at a rotated offset (b * S / G).  Prints the aggregate rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ p, unsigned n4, int mode, float* out) {
    const unsigned rot = mode ? (unsigned)(((unsigned long long)blockIdx.x * n4) / gridDim.x) & ~255u : 0u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned base = 0; base < n4; base += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned i = base + u * 256 + threadIdx.x;
            i = i < n4 ? i : n4 - 1;
            unsigned j = i + rot; if (j >= n4) j -= n4;
            v[u] = p[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main(int argc, char** argv) {
    const size_t S = argc > 1 ? atol(argv[1]) : 1800000;
    const int G = argc > 2 ? atoi(argv[2]) : 236;
    const unsigned n4 = (unsigned)(S / 16);
    float4* d; float* o;
    hipMalloc(&d, (size_t)n4 * 16); hipMalloc(&o, 16);
    hipMemset(d, 0, (size_t)n4 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int U : {2, 4, 8, 16}) {
            auto launch = [&]() {
                if (U == 2) hipLaunchKernelGGL(k_stream<2>, dim3(G), dim3(256), 0, 0, d, n4, mode, o);
                else if (U == 4) hipLaunchKernelGGL(k_stream<4>, dim3(G), dim3(256), 0, 0, d, n4, mode, o);
                else if (U == 8) hipLaunchKernelGGL(k_stream<8>, dim3(G), dim3(256), 0, 0, d, n4, mode, o);
                else hipLaunchKernelGGL(k_stream<16>, dim3(G), dim3(256), 0, 0, d, n4, mode, o);
            };
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e0, 0);
            const int reps = 50;
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps;
            printf("S=%zu G=%d mode=%s U=%2d  %7.1f us  %6.2f TB/s aggregate  %5.1f GB/s per WG\n", S, G, mode ? "rotated " : "lockstep", U, us,
                   (double)S * G / us / 1e6, (double)S / us / 1e3);
        }
    return 0;
}
