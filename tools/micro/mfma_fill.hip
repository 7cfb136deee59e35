// mfma_fill.hip - how much vector work one wave per SIMD can hide under the MFMAs of a k_edge128 GEMM quarter.
// One 256-thread workgroup per CU (512 registers per wave).  An iteration is one quarter of a 128-row edge tile: 4 k-blocks x 48
// v_mfma_f32_32x32x16_bf16 (A fragments from three bf16 planes in LDS, split weight fragments from L2, as gemm_quarter of
// kernels_edge128.hip), and - depending on MODE - the BUILD of the next quarter's planes into the other LDS buffer: 32 elements per thread
// of SiLU(P[row] + Q[col] + w_r r + w_d d0) split into three bf16 pieces (gathers of the quarter after next in flight).
//   MODE 0  GEMM only                               MODE 1  build, barrier, GEMM (one after the other)
//   MODE 2  build code inside the k-block, MFMAs / memory operations pinned by sched_barrier fences that vector instructions may cross
//   MODE 3  build code inside the k-block, sched_group_barrier pattern (1 MFMA, FILL vector instructions) per k-block
//   MODE 4  the build cut by hand into pieces of about six issue slots, one piece pinned behind one MFMA (sched_barrier(0) after every piece)
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -I../../cmdgen_amd/csrc mfma_fill.hip -o mfma_fill && ./mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "cmdgen_split.h"

constexpr int H = 256, MT = 128, KQ = 64, PLDA = KQ + 8, PE = MT * PLDA;
constexpr unsigned NS = 16u * 192u;
#ifndef FILL
#define FILL 5
#endif

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Gath { float4 p[4], q[4]; };        // 64 rows: four 16-row groups, one float4 of P and of Q per thread and group

__device__ __forceinline__ void gather64(Gath& g, const float* __restrict__ T, const int* __restrict__ idx, int tid, int half, int qq) {
    const int c4 = tid & 15, rsub = tid >> 4;
    const unsigned cofs = (unsigned)(qq * KQ + 4 * c4) * 4u;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int e = half * 64 + ps * 16 + rsub;
        const int2 rc = *reinterpret_cast<const int2*>(idx + 2 * e);
        g.p[ps] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T) + ((unsigned)rc.x * (unsigned)(H * 4) + cofs));
        g.q[ps] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T) + ((unsigned)rc.y * (unsigned)(H * 4) + cofs));
    }
}
// one 16-row group: 4 elements per thread -> three planes
__device__ __forceinline__ void build_group(unsigned short* planes, const float* rd, int tid, int half, int ps, const float4& P, const float4& Q, const float4& wr4, const float4& wd4) {
    const int c4 = tid & 15, rsub = tid >> 4;
    const int e = half * 64 + ps * 16 + rsub;
    const float2 r2 = *reinterpret_cast<const float2*>(rd + 2 * e);
    const float r = r2.x, d0 = r2.y;
    const float4 a = make_float4(silu_f(P.x + Q.x + wr4.x * r + wd4.x * d0), silu_f(P.y + Q.y + wr4.y * r + wd4.y * d0),
                                 silu_f(P.z + Q.z + wr4.z * r + wd4.z * d0), silu_f(P.w + Q.w + wr4.w * r + wd4.w * d0));
    split_store4(planes, PE, e * PLDA + 4 * c4, a);
}


// ---- MODE 4: the build of a 16-row group cut into eleven pieces of about six issue slots each, one piece behind one MFMA, everything pinned
struct GrpState { float t[4], u[4], a[4], rr[4]; unsigned p0[2], p1[2], p2[2]; float r, d0; int off; };
template <int PIECE>
__device__ __forceinline__ void build_piece(GrpState& g, unsigned short* planes, const float* rd, int tid, int half, int ps, const float4& P, const float4& Q, const float4& wr4, const float4& wd4) {
    const float wr[4] = {wr4.x, wr4.y, wr4.z, wr4.w}, wd[4] = {wd4.x, wd4.y, wd4.z, wd4.w};
    const float Pv[4] = {P.x, P.y, P.z, P.w}, Qv[4] = {Q.x, Q.y, Q.z, Q.w};
    constexpr float NL2E = -1.4426950408889634f;
    if constexpr (PIECE == 0) {
        const int c4 = tid & 15, rsub = tid >> 4, e = half * 64 + ps * 16 + rsub;
        const float2 r2 = *reinterpret_cast<const float2*>(rd + 2 * e);
        g.r = r2.x; g.d0 = r2.y; g.off = e * PLDA + 4 * c4;
#pragma unroll
        for (int i = 0; i < 4; ++i) g.t[i] = Pv[i] + Qv[i];
        g.t[0] = __fmaf_rn(wr[0], g.r, g.t[0]); g.t[1] = __fmaf_rn(wr[1], g.r, g.t[1]);
    } else if constexpr (PIECE == 1) {
        g.t[2] = __fmaf_rn(wr[2], g.r, g.t[2]); g.t[3] = __fmaf_rn(wr[3], g.r, g.t[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) g.t[i] = __fmaf_rn(wd[i], g.d0, g.t[i]);
    } else if constexpr (PIECE == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g.u[i] = g.t[i] * NL2E;
        g.u[0] = __builtin_amdgcn_exp2f(g.u[0]);
    } else if constexpr (PIECE == 3) {
        g.u[1] = __builtin_amdgcn_exp2f(g.u[1]); g.u[2] = __builtin_amdgcn_exp2f(g.u[2]); g.u[3] = __builtin_amdgcn_exp2f(g.u[3]);
    } else if constexpr (PIECE == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g.u[i] = g.u[i] + 1.0f;
        g.u[0] = __builtin_amdgcn_rcpf(g.u[0]);
    } else if constexpr (PIECE == 5) {
        g.u[1] = __builtin_amdgcn_rcpf(g.u[1]); g.u[2] = __builtin_amdgcn_rcpf(g.u[2]); g.u[3] = __builtin_amdgcn_rcpf(g.u[3]);
    } else if constexpr (PIECE == 6) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g.a[i] = g.t[i] * g.u[i];
        g.p0[0] = cvt_pk_bf16(g.a[0], g.a[1]); g.p0[1] = cvt_pk_bf16(g.a[2], g.a[3]);
    } else if constexpr (PIECE == 7) {
        g.rr[0] = g.a[0] - __uint_as_float(g.p0[0] << 16); g.rr[1] = g.a[1] - __uint_as_float(g.p0[0] & 0xffff0000u);
        g.rr[2] = g.a[2] - __uint_as_float(g.p0[1] << 16);
    } else if constexpr (PIECE == 8) {
        g.rr[3] = g.a[3] - __uint_as_float(g.p0[1] & 0xffff0000u);
        g.p1[0] = cvt_pk_bf16(g.rr[0], g.rr[1]); g.p1[1] = cvt_pk_bf16(g.rr[2], g.rr[3]);
        g.rr[0] = g.rr[0] - __uint_as_float(g.p1[0] << 16);
    } else if constexpr (PIECE == 9) {
        g.rr[1] = g.rr[1] - __uint_as_float(g.p1[0] & 0xffff0000u);
        g.rr[2] = g.rr[2] - __uint_as_float(g.p1[1] << 16); g.rr[3] = g.rr[3] - __uint_as_float(g.p1[1] & 0xffff0000u);
    } else if constexpr (PIECE == 10) {
        g.p2[0] = cvt_pk_bf16(g.rr[0], g.rr[1]); g.p2[1] = cvt_pk_bf16(g.rr[2], g.rr[3]);
        *reinterpret_cast<uint2*>(planes + g.off) = make_uint2(g.p0[0], g.p0[1]);
        *reinterpret_cast<uint2*>(planes + PE + g.off) = make_uint2(g.p1[0], g.p1[1]);
        *reinterpret_cast<uint2*>(planes + 2 * PE + g.off) = make_uint2(g.p2[0], g.p2[1]);
    }
}

// calibration: FILL instructions in EVERY gap.  MODE 5: independent v_fma; 6: v_exp; 7: one dependent v_fma chain
template <int MODE>
__device__ __forceinline__ void calib(float (&dm)[8]) {
#pragma unroll
    for (int f = 0; f < FILL; ++f) {
        if (MODE == 5) dm[f] = __fmaf_rn(dm[f], 1.0001f, 0.5f);
        else if (MODE == 6) dm[f] = __builtin_amdgcn_exp2f(dm[f]);
        else dm[0] = __fmaf_rn(dm[0], 1.0001f, 0.5f);
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink, const float* __restrict__ T, const int* __restrict__ idx, const sbf16x8* __restrict__ W, int iters) {
    __shared__ unsigned short planes[2][3 * PE + 64];
    __shared__ float rd[2 * MT];
    __shared__ int lidx[2 * MT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * (3 * PE + 64); i += 256) (&planes[0][0])[i] = (unsigned short)(0x3c00 + (i * 37 & 0xff));       // finite bf16 values
    for (int i = tid; i < 2 * MT; i += 256) { rd[i] = 0.01f * (i & 15); lidx[i] = idx[(blockIdx.x * 2 * MT + i) & 0xffff]; }
    __syncthreads();
    sf32x16 acc[4][2];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) { acc[m][0][r] = 0.f; acc[m][1][r] = 0.f; }
    const sbf16x8* wb = W + (size_t)(2 * wave) * 16 * 192 + lane;
    sbf16x8 bs[2][2][3];
#pragma unroll
    for (int i = 0; i < 6; ++i) bs[0][i & 1][i >> 1] = wb[(unsigned)(i & 1) * NS + (unsigned)(i >> 1) * 64u];
    const float4 wr4 = make_float4(0.01f, 0.02f, 0.03f, 0.04f), wd4 = make_float4(0.04f, 0.03f, 0.02f, 0.01f);
    float dm[8];
    for (int i = 0; i < 8; ++i) dm[i] = 0.001f * (tid + i);
    Gath gx, gy;
    gather64(gx, T, lidx, tid, 0, 0); gather64(gy, T, lidx, tid, 1, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const int q = it & 3, pb = it & 1;
        unsigned short* dst = planes[pb ^ 1];
        if (MODE == 1) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) build_group(dst, rd, tid, 0, ps, gx.p[ps], gx.q[ps], wr4, wd4);
            gather64(gx, T, lidx, tid, 0, (q + 1) & 3);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) build_group(dst, rd, tid, 1, ps, gy.p[ps], gy.q[ps], wr4, wd4);
            gather64(gy, T, lidx, tid, 1, (q + 1) & 3);
            lds_barrier();
        }
        const unsigned short* ap = planes[pb] + (lane & 31) * PLDA + (lane >> 5) * 8;
        sbf16x8 a[2][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) a[0][s] = *reinterpret_cast<const sbf16x8*>(ap + s * PE);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const sbf16x8* qn = wb + (unsigned)((4 * q + kq + 1) & 15) * 192u;
            GrpState gs0, gs1;
            if (MODE == 2 || MODE == 3) {
                // the build of two 16-row groups rides in this k-block; the gathers of the quarter after next follow a batch's last use
                if (kq < 2) { build_group(dst, rd, tid, 0, 2 * kq, gx.p[2 * kq], gx.q[2 * kq], wr4, wd4); build_group(dst, rd, tid, 0, 2 * kq + 1, gx.p[2 * kq + 1], gx.q[2 * kq + 1], wr4, wd4); }
                else { build_group(dst, rd, tid, 1, 2 * kq - 4, gy.p[2 * kq - 4], gy.q[2 * kq - 4], wr4, wd4); build_group(dst, rd, tid, 1, 2 * kq - 3, gy.p[2 * kq - 3], gy.q[2 * kq - 3], wr4, wd4); }
                if (kq == 1) gather64(gx, T, lidx, tid, 0, (q + 1) & 3);
                if (kq == 3) gather64(gy, T, lidx, tid, 1, (q + 1) & 3);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int itn = kq * 4 + m, cs = itn & 1, nx = cs ^ 1, bc = kq & 1, bn = bc ^ 1;
                const bool more_a = (m + 1 < 4) || (kq < 3);
                const unsigned short* an = ap + ((m + 1 < 4) ? (m + 1) * 32 * PLDA + kq * 16 : (kq + 1) * 16);
                const int b0 = m * 3;
#define E_MF(N, AI, BI) acc[m][N] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][AI], bs[bc][N][BI], acc[m][N], 0, 0, 0);
#define E_LA(S) if (more_a) a[nx][S] = *reinterpret_cast<const sbf16x8*>(an + (S) * PE);
#define E_LB(I) if ((I) < 6 && (I) >= b0 && (I) < b0 + 3) bs[bn][(I) & 1][(I) >> 1] = qn[(unsigned)((I) & 1) * NS + (unsigned)((I) >> 1) * 64u];
#define FENCE() do { if (MODE == 2) __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400); else if (MODE != 3) __builtin_amdgcn_sched_barrier(0); } while (0)
                // MODE 4: MFMA j of the k-block (j = 12 m + 0 .. 11) is followed by piece j of the build schedule: pieces 0 .. 10 of group 2 kq in
                // slots 0 .. 21 (even), of group 2 kq + 1 in the odd ones
#define E_FILL(J) do { if (MODE == 4) { constexpr int j_ = (J); const int mj_ = 12 * m + j_;                                                             \
                    if (mj_ < 22) { const int pc_ = mj_ >> 1;                                                                                           \
                        const int gi_ = (mj_ & 1), grp_ = 2 * kq + gi_, hf_ = grp_ >> 2, ps_ = grp_ & 3;                                               \
                        GrpState& gst_ = gi_ ? gs1 : gs0; Gath& gg_ = hf_ ? gy : gx;                                                                   \
                        switch (pc_) {                                                                                                                 \
                            case 0: build_piece<0>(gst_, dst, rd, tid, hf_, ps_, gg_.p[ps_], gg_.q[ps_], wr4, wd4); break;                             \
                            case 1: build_piece<1>(gst_, dst, rd, tid, hf_, ps_, gg_.p[ps_], gg_.q[ps_], wr4, wd4); break;                             \
                            case 2: build_piece<2>(gst_, dst, rd, tid, hf_, ps_, gg_.p[ps_], gg_.q[ps_], wr4, wd4); break;                             \
                            case 3: build_piece<3>(gst_, dst, rd, tid, hf_, ps_, gg_.p[ps_], gg_.q[ps_], wr4, wd4); break;                             \
                            case 4: build_piece<4>(gst_, dst, rd, tid, hf_, ps_, gg_.p[ps_], gg_.q[ps_], wr4, wd4); break;                             \
                            case 5: build_piece<5>(gst_, dst, rd, tid, hf_, ps_, gg_.p[ps_], gg_.q[ps_], wr4, wd4); break;                             \
                            case 6: build_piece<6>(gst_, dst, rd, tid, hf_, ps_, gg_.p[ps_], gg_.q[ps_], wr4, wd4); break;                             \
                            case 7: build_piece<7>(gst_, dst, rd, tid, hf_, ps_, gg_.p[ps_], gg_.q[ps_], wr4, wd4); break;                             \
                            case 8: build_piece<8>(gst_, dst, rd, tid, hf_, ps_, gg_.p[ps_], gg_.q[ps_], wr4, wd4); break;                             \
                            case 9: build_piece<9>(gst_, dst, rd, tid, hf_, ps_, gg_.p[ps_], gg_.q[ps_], wr4, wd4); break;                             \
                            default: build_piece<10>(gst_, dst, rd, tid, hf_, ps_, gg_.p[ps_], gg_.q[ps_], wr4, wd4); break; } }                       \
                    if (mj_ == 24 && kq == 1) gather64(gx, T, lidx, tid, 0, (q + 1) & 3);                                                              \
                    if (mj_ == 24 && kq == 3) gather64(gy, T, lidx, tid, 1, (q + 1) & 3);                                                              \
                    __builtin_amdgcn_sched_barrier(0); }                                                                                                \
                    if (MODE >= 5) { calib<MODE>(dm); __builtin_amdgcn_sched_barrier(0); } } while (0)
                E_LA(2) E_MF(0, 2, 0) E_FILL(0); E_MF(1, 2, 0) E_FILL(1); FENCE();
                E_LA(1) E_MF(0, 1, 1) E_FILL(2); E_MF(1, 1, 1) E_FILL(3); FENCE();
                E_LA(0) E_MF(0, 0, 2) E_FILL(4); E_MF(1, 0, 2) E_FILL(5); FENCE();
                E_LB(b0) E_LB(b0 + 3) E_MF(0, 1, 0) E_FILL(6); E_MF(1, 1, 0) E_FILL(7); FENCE();
                E_LB(b0 + 1) E_LB(b0 + 4) E_MF(0, 0, 1) E_FILL(8); E_MF(1, 0, 1) E_FILL(9); FENCE();
                E_LB(b0 + 2) E_LB(b0 + 5) E_MF(0, 0, 0) E_FILL(10); E_MF(1, 0, 0) E_FILL(11); FENCE();
#undef E_MF
#undef E_LA
#undef E_LB
#undef E_FILL
            }
            if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, FILL, 0);
                    if ((i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // a DS read every other MFMA (A fragments, edge records)
                    if ((i & 7) == 3) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);       // VMEM reads (weight fragments, gathers)
                    if ((i & 7) == 7) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // DS writes of the build
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][0][r] + acc[m][1][r];
    s += gx.p[0].x + gy.q[3].w;
    for (int i = 0; i < 8; ++i) s += dm[i];
    if (s == 12345.f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, iters = 4000, rows = argc > 2 ? atoi(argv[2]) : 65536;
    unsigned long long* out; float *sink, *T; int* idx; sbf16x8* W;
    hipMalloc(&out, wgs * 4 * sizeof(unsigned long long)); hipMalloc(&sink, 4);
    hipMalloc(&T, (size_t)rows * H * 4); hipMalloc(&idx, 65536 * 4); hipMalloc(&W, 256 * 256 * 6);
    std::vector<float> hT((size_t)rows * H); for (size_t i = 0; i < hT.size(); ++i) hT[i] = ((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    std::vector<int> hi(65536); for (int i = 0; i < 65536; ++i) hi[i] = (int)((i * 2654435761u >> 7) % rows);
    std::vector<unsigned short> hw(256 * 256 * 3); for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned short)(0x3c00 + (i * 13 & 0x7f));
    hipMemcpy(T, hT.data(), hT.size() * 4, hipMemcpyHostToDevice); hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 8; ++mode) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, out, sink, T, idx, W, iters);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, out, sink, T, idx, W, iters);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, out, sink, T, idx, W, iters);
            else if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(wgs), dim3(256), 0, 0, out, sink, T, idx, W, iters);
            else if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(wgs), dim3(256), 0, 0, out, sink, T, idx, W, iters);
            else if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(wgs), dim3(256), 0, 0, out, sink, T, idx, W, iters);
            else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(256), 0, 0, out, sink, T, idx, W, iters);
            else hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), 0, 0, out, sink, T, idx, W, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(wgs * 4); hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("mode %d (fill %d): %.0f cycles per quarter (median wave; 192 MFMAs = 6144 at 32 each), %.2f us per quarter by events, hipErr %d\n", mode, FILL, (double)h[h.size() / 2] / iters, ms * 1e3 / iters, (int)hipGetLastError());
        }
    return 0;
}
