// kernels_node16w.hip - k_node for SMALL batches: a 16-row tile on EIGHT waves (two per SIMD).
//
// Why.  At 64 C-alpha pockets the node launch has 236 tiles of 16 rows for 256 CUs: one workgroup per CU, and k_node<256, 16> (kernels_egnn.hip)
// runs it on four waves - ONE wave per SIMD, each owning 64 output columns.  Measured (profiles/r04_p_node16_eight_waves.txt): ~1000 cycles per
// 32-k block where its 24 MFMAs need 384; a third less weight traffic buys 6 %, NO weight traffic at all 20 % - the launch is neither bound by
// the matrix pipe nor by the weight stream but by one wave's serial issue: every LDS round trip, every s_waitcnt, every VALU instruction of the
// A-side split (4 cycles each when a wave is alone on its SIMD, 2 when two alternate; MI355X_MICROARCH.md constants table) stands in the
// critical path of the only wave the SIMD has.  Here the same tile runs on eight waves of 32 columns each: two waves per SIMD cover each
// other's stalls, the A-side split is issued at the two-wave rate, and a wave's MFMA chain per k-block is 12 long instead of 24.
// Arithmetic, weight packs (WPack::ws16), LDS images and epilogues are those of Eng<16, true> / node_tile_body: GCL.node_model
// (egnn_new.py:48-58) and the projections of the first edge / coordinate MLP layers; every accumulator sees the MFMAs of k_node<256, 16, false, true>
// in the same order.
#include "cmdgen_dev.h"
#include "cmdgen_split.h"
#include <hip/hip_ext.h>

namespace {
constexpr int NW_H = 256, NW_MT = 16, NW_LD = NW_H + 4 /* LDA(H), kernels_egnn.hip */, NW_RD = 4;

__device__ __forceinline__ void nw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct NwFrag { const sbf16x8* p; unsigned ns; };       // a wave's first 16-column tile (k-block kb0); ns = stride between n-tiles (16-byte units)
__device__ __forceinline__ NwFrag nw_frag(const void* Ws16, int kb32_total, int kb0, int nt0) {
    const int lane = threadIdx.x & 63;
    NwFrag f;
    f.p = reinterpret_cast<const sbf16x8*>(Ws16) + ((size_t)nt0 * kb32_total + kb0) * 192 + lane;
    f.ns = (unsigned)kb32_total * 192u;
    return f;
}
struct NwRing { sbf16x8 b[NW_RD][2][3]; };              // ring of four k-blocks x [2 n-tiles][3 pieces]
__device__ __forceinline__ void nw_load_set(const sbf16x8* q, unsigned ns, sbf16x8 (&dst)[2][3]) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int s = 0; s < 3; ++s) dst[n][s] = q[n * ns + s * 64];
}
// a GEMM enters with its k-blocks 0, 1, 2 in sets 0, 1, 2 and leaves with those of `next` there
__device__ __forceinline__ void nw_prefetch(const NwFrag& f, NwRing& c) {
    nw_load_set(f.p, f.ns, c.b[0]); nw_load_set(f.p + 192, f.ns, c.b[1]); nw_load_set(f.p + 384, f.ns, c.b[2]);
}

// acc[n] += A(lds fp32 image, 16 rows) x W_n^T over KB32 * 32 k-values for the wave's two 16-column tiles (tile_gemm_rsplit16 at half the width)
template <int KB32>
__device__ __forceinline__ void nw_gemm(const float* ldsA, const NwFrag cur, const NwFrag next, sf32x4 (&acc)[2], NwRing& ring) {
    static_assert(KB32 % 4 == 0, "K must be a multiple of 128");
    const int lane = threadIdx.x & 63;
    const float* ap = ldsA + (lane & 15) * NW_LD + (lane >> 4) * 4;
    float4 raw[2][2];
    sbf16x8 a[2][3];
#define NW_LOADA(SET, PTR) { raw[SET][0] = *reinterpret_cast<const float4*>(PTR); raw[SET][1] = *reinterpret_cast<const float4*>((PTR) + 16); }
#define NW_SPLIT(DST, SET) split8(raw[SET][0], raw[SET][1], a[DST][0], a[DST][1], a[DST][2]);
    // small terms first; the two n-tiles alternate so that consecutive MFMAs never share an accumulator
#define NW_MFMAS(AS, BS)                                                                                                    \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS][2], ring.b[BS][n][0], acc[n], 0, 0, 0); \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS][1], ring.b[BS][n][1], acc[n], 0, 0, 0); \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS][0], ring.b[BS][n][2], acc[n], 0, 0, 0); \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS][1], ring.b[BS][n][0], acc[n], 0, 0, 0); \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS][0], ring.b[BS][n][1], acc[n], 0, 0, 0); \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS][0], ring.b[BS][n][0], acc[n], 0, 0, 0);
    // the next block's split (44 VALU operations) spread over this block's twelve MFMAs
#define NW_INTERLEAVE()                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 12; ++i) {                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); }
    // block i: MFMAs on set i % 4; set (i + 3) % 4 <- weight block i + 3 (or block i + 3 - KB32 of `next`);
    // raw[i % 2] <- A block i + 2; a[(i + 1) % 2] <- split of raw[(i + 1) % 2]
#define NW_BLOCK(I)                                                                                                         \
    {                                                                                                                       \
        const bool tail = kb + (I) + 3 >= KB32;                      /* wave-uniform */                                     \
        const sbf16x8* q = tail ? next.p + (unsigned)(kb + (I) + 3 - KB32) * 192u : cur.p + (unsigned)(kb + (I) + 3) * 192u; \
        nw_load_set(q, tail ? next.ns : cur.ns, ring.b[((I) + 3) & 3]);                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        if (kb + (I) + 1 < KB32) { NW_SPLIT(((I) + 1) & 1, ((I) + 1) & 1) }                                                 \
        NW_MFMAS((I) & 1, (I) & 3)                                                                                          \
        NW_INTERLEAVE()                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        if (kb + (I) + 2 < KB32) { NW_LOADA((I) & 1, ap + (kb + (I) + 2) * 32) }                                            \
    }
    NW_LOADA(0, ap)
    NW_LOADA(1, ap + 32)
    NW_SPLIT(0, 0)
#pragma unroll 1
    for (int kb = 0; kb < KB32; kb += 4) {
        NW_BLOCK(0) NW_BLOCK(1) NW_BLOCK(2) NW_BLOCK(3)
    }
#undef NW_LOADA
#undef NW_SPLIT
#undef NW_MFMAS
#undef NW_INTERLEAVE
#undef NW_BLOCK
}

// the wave's accumulators: lane l, reg r of tile n -> row 4 (l >> 4) + r, column 32 wave + 16 n + (l & 15)
template <class F>
__device__ __forceinline__ void nw_foreach(const sf32x4 (&acc)[2], int wave, F f) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) f(4 * (lane >> 4) + r, wave * 32 + n * 16 + (lane & 15), n, acc[n][r]);
}
struct NwCol { float v[2]; };
__device__ __forceinline__ NwCol nw_col(const float* __restrict__ vec, int wave) {
    const int lane = threadIdx.x & 63;
    NwCol c;
#pragma unroll
    for (int n = 0; n < 2; ++n) c.v[n] = vec[wave * 32 + n * 16 + (lane & 15)];
    return c;
}
__device__ __forceinline__ void nw_zero(sf32x4 (&acc)[2]) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[n][r] = 0.0f;
}
__device__ __forceinline__ void nw_store(const sf32x4 (&acc)[2], int wave, float* __restrict__ out, int row0, int nvalid, const NwCol* bias) {
    nw_foreach(acc, wave, [&](int row, int col, int n, float v) {
        if (row < nvalid) out[(size_t)(row0 + row) * NW_H + col] = v + (bias ? bias->v[n] : 0.f);
    });
}
}  // namespace

__global__ __launch_bounds__(512) void k_node16w(Layout lay, Work w, Dims d, LayerW lw, LayerW lw_next, int layer, int has_next_arg) {
    __shared__ __attribute__((aligned(16))) float buf0[NW_MT * NW_LD];      // h (kept for the residual)
    __shared__ __attribute__((aligned(16))) float buf1[NW_MT * NW_LD];      // agg / nf -> T = SiLU(.) -> h_new
    const int has_next = has_next_arg & 1;                                   // (bits 1..29: the dead-tile threshold of the plane tiles, unused here)
    const bool skip_pc = ((has_next_arg >> 30) & 1) != 0;                   // not the last GCL of its block (inv_sublayers > 1): no P_c | Q_c
    const int tid = threadIdx.x, wave = tid >> 6;
    const int row0 = (int)blockIdx.x * NW_MT, nvalid = min(NW_MT, lay.N - row0);
    const bool want_pc = row0 < lay.Nm;                                      // the tile holds receivers that move
    constexpr int KB = NW_H / 32;
    // the chain of GEMMs of this tile; each one's last blocks fetch the next one's first fragments
    const NwFrag f3a = nw_frag(lw.W3.ws16, 2 * KB, 0, 2 * wave), f3b = nw_frag(lw.W3.ws16, 2 * KB, KB, 2 * wave);
    const NwFrag f4 = nw_frag(lw.W4.ws16, KB, 0, 2 * wave);
    const NwFrag fcp = nw_frag(lw.Wpq_c.ws16, KB, 0, 2 * wave), fcq = nw_frag(lw.Wpq_c.ws16, KB, 0, NW_H / 16 + 2 * wave);
    const NwFrag fnp = nw_frag(lw_next.Wpq_e.ws16, KB, 0, 2 * wave), fnq = nw_frag(lw_next.Wpq_e.ws16, KB, 0, NW_H / 16 + 2 * wave);
    NwRing ring;
    nw_prefetch(f3a, ring);
    const NwCol b3v = nw_col(lw.b3, wave), b4v = nw_col(lw.b4, wave), b6v = nw_col(lw.b6, wave), b1nv = nw_col(lw_next.b1, wave);
    // materialise the phar coordinates entering this block (node_pos, kernels_egnn.hip: X[l] = X[l-1] + ACC[l-1] / normalization_factor)
    if (layer >= 1 && tid < NW_MT) {
        const int n = row0 + tid;
        if (tid < nvalid && n < lay.Nm) {
            const float4 p = (layer == 1) ? w.X0[n] : w.XL[(size_t)(layer - 1) * lay.Nm + n];
            const float4 a = w.ACC[(size_t)(layer - 1) * lay.Nm + n];
            const float dv = agg_div(w, d, n);
            w.XL[(size_t)layer * lay.Nm + n] = make_float4(p.x + a.x / dv, p.y + a.y / dv, p.z + a.z / dv, 0.f);
        }
    }
    // h and agg of the tile: all global loads in flight together, then the LDS writes; agg is zero between blocks
    {
        float4 hv[2], av[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 512 * j, r = idx >> 6, c4 = idx & 63;
            hv[j] = make_float4(0.f, 0.f, 0.f, 0.f); av[j] = hv[j];
            if (r < nvalid) {
                hv[j] = reinterpret_cast<const float4*>(w.h + (size_t)(row0 + r) * NW_H)[c4];
                av[j] = reinterpret_cast<const float4*>(w.agg + (size_t)(row0 + r) * NW_H)[c4];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 512 * j, r = idx >> 6, c4 = idx & 63;
            if (r < nvalid) reinterpret_cast<float4*>(w.agg + (size_t)(row0 + r) * NW_H)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 v = av[j];
            const float dv = r < nvalid ? agg_div(w, d, row0 + r) : 1.0f;
            v.x /= dv; v.y /= dv; v.z /= dv; v.w /= dv;
            *reinterpret_cast<float4*>(buf0 + r * NW_LD + 4 * c4) = hv[j];
            *reinterpret_cast<float4*>(buf1 + r * NW_LD + 4 * c4) = v;
        }
    }
#if CMDGEN_STAMPS == 7      // diagnostic build: per-phase cycle stamps into w.dbg (tools/node_stamps.py; waves 0..3 report)
    unsigned long long nst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nst_t = __builtin_amdgcn_s_memtime();
    const unsigned long long nst_begin = nst_t;
#define NSTAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); nst_[i] += n_ - nst_t; nst_t = n_; } while (0)
#else
#define NSTAMP(i) do {} while (0)
#endif
    nw_barrier();
    NSTAMP(0);
    sf32x4 acc[2];
    nw_zero(acc);
    nw_gemm<KB>(buf0, f3a, f3b, acc, ring);                                  // h part of [h | agg]
    nw_gemm<KB>(buf1, f3b, f4, acc, ring);                                   // agg part
    NSTAMP(1);
    nw_barrier();
    nw_foreach(acc, wave, [&](int row, int col, int n, float v) { buf1[row * NW_LD + col] = silu_f(v + b3v.v[n]); });
    nw_barrier();
    NSTAMP(2);
    nw_zero(acc);
    const NwFrag fc = skip_pc ? fnp : want_pc ? fcp : fcq;                   // the GEMM behind W4
    nw_gemm<KB>(buf1, f4, fc, acc, ring);
    NSTAMP(3);
    nw_barrier();
    nw_foreach(acc, wave, [&](int row, int col, int n, float v) {
        float hn = 0.f;
        if (row < nvalid) {
            hn = buf0[row * NW_LD + col] + (v + b4v.v[n]);                   // residual (egnn_new.py:57)
            w.h[(size_t)(row0 + row) * NW_H + col] = hn;
        }
        buf1[row * NW_LD + col] = hn;
    });
    nw_barrier();
    NSTAMP(4);
    // coordinate-MLP projections: P_c only where the tile holds phar rows (receivers that move); then P | Q of the next block's edge MLP
    // (results kept in registers and stored after the last GEMM: measured, no gain - profiles/r04_p_node16_eight_waves.txt)
    if (want_pc && !skip_pc) {
        nw_zero(acc);
        nw_gemm<KB>(buf1, fcp, fcq, acc, ring);
        nw_store(acc, wave, w.Pc, row0, nvalid, &b6v);
    }
    if (!skip_pc) {
        nw_zero(acc);
        nw_gemm<KB>(buf1, fcq, has_next ? fnp : fcq, acc, ring);
        nw_store(acc, wave, w.Qc, row0, nvalid, nullptr);
    }
    NSTAMP(5);
    if (has_next) {
        nw_zero(acc);
        nw_gemm<KB>(buf1, fnp, fnq, acc, ring);
        nw_store(acc, wave, w.P, row0, nvalid, &b1nv);
        nw_zero(acc);
        nw_gemm<KB>(buf1, fnq, fnq, acc, ring);
        nw_store(acc, wave, w.Q, row0, nvalid, nullptr);
    }
    NSTAMP(6);
#if CMDGEN_STAMPS == 7
    if ((tid & 63) == 0 && wave < 4) {
        for (int i = 0; i < 7; ++i) atomicAdd(&w.dbg[wave * 8 + i], nst_[i]);
        atomicAdd(&w.dbg[32 + wave], __builtin_amdgcn_s_memtime() - nst_begin);
        atomicAdd(&w.dbg[40], 1ull);
    }
#endif
#undef NSTAMP
}

// launcher: true when the eight-wave kernel took the launch (H = 256, 16-row tiles on the split engine, sampler)
bool cmdgen_launch_node16w(const EvalLaunch& a, int l, hipStream_t s) {
    if (a.d.H != 256 || a.node_mt != 16 || !a.split16 || !a.node16w || a.save || !a.layers[unit_of(a, l)].W3.ws16) return false;
    const int nt = (a.lay.N + NW_MT - 1) / NW_MT;
    if (a.pe_start) hipExtLaunchKernelGGL(k_node16w, dim3(nt), dim3(512), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, a.layers[unit_of(a, l)],
                                          a.layers[unit_has_next(a, l) ? unit_of(a, l) + 1 : unit_of(a, l)], l, node_flags(a, l));
    else hipLaunchKernelGGL(k_node16w, dim3(nt), dim3(512), 0, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)], a.layers[unit_has_next(a, l) ? unit_of(a, l) + 1 : unit_of(a, l)], l, node_flags(a, l));
    return true;
}
