// split_check.hip - the two forms of the half engine's second piece agree bit for bit on the device:
//   old: v_fma_mix_f32 (a - (float)a0) then v_cvt_pk_f16_f32        new (round 6): v_fma_mixlo_f16 / v_fma_mixhi_f16
// over magnitudes from subnormal second pieces to the end of fp16's range.   hipcc --offload-arch=gfx950 -O3 -o build/split_check tools/micro/split_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ uint32_t cvt_pk(float a, float b) { typedef float f2 __attribute__((ext_vector_type(2))); const f2 v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2)); }
__global__ void k(const float* x, uint32_t* o_old, uint32_t* o_new, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    const uint32_t p0 = cvt_pk(a, b);
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(p0), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(p0), "v"(b));
    o_old[i] = cvt_pk(ra, rb);
    uint32_t r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p0), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(p0), "v"(b));
    o_new[i] = r;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;        // [0, 1)
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int e = (int)((s >> 20) % 44) - 27;                        // 2^-27 .. 2^16
        x[i] = (float)((u * 2.0 - 1.0) * std::ldexp(1.0, e));
    }
    float* dx; uint32_t *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 2); hipMalloc(&db, n * 2);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, da, db, n);
    std::vector<uint32_t> a(n / 2), b(n / 2);
    hipMemcpy(a.data(), da, n * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 2, hipMemcpyDeviceToHost);
    long bad = 0, sub = 0;
    for (int i = 0; i < n / 2; ++i) {
        if (a[i] != b[i]) { if (bad < 5) printf("mismatch at %d: x = %g %g old %08x new %08x\n", i, x[2 * i], x[2 * i + 1], a[i], b[i]); ++bad; }
        if ((a[i] & 0x7c00u) == 0 && (a[i] & 0x3ffu)) ++sub;
    }
    printf("split_check: %d pairs, %ld with a subnormal second piece, %ld mismatches\n", n / 2, sub, bad);
    return bad ? 1 : 0;
}
