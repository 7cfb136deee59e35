// stream_overlap.hip - do kernels of two streams run side by side on this device?  Each kernel is ONE workgroup spinning for ~1 ms.
//   hipcc -O3 --offload-arch=gfx950 stream_overlap.hip -o stream_overlap && ./stream_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void spin(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
    if (ticks == 1) sink[0] = 1;
}
int main() {
    int* sink; hipMalloc(&sink, 4);
    hipStream_t s[4]; for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[0], 1000ull, sink); hipDeviceSynchronize();
    for (int n : {1, 2, 4}) {
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < 10; ++rep) for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], 100000ull /* 1 ms at 100 MHz */, sink);
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%d stream(s) x 10 kernels of 1 ms each: %.2f ms (%s)\n", n, ms, ms < 10.0 * n * 0.7 ? "overlap" : "serial");
    }
    return 0;
}
