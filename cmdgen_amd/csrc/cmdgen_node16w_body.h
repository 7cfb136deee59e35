// cmdgen_node16w_body.h - the body of kernels_node16w.hip, included once per matrix engine (NW_NPL = 3: three bf16 pieces per operand, six
// MFMAs per product; 2: two fp16 pieces, three MFMAs - the "half" engine of cmdgen_split.h) inside a namespace of its own.  No include guard.
#ifndef NW_RD_HALF
#define NW_RD_HALF 8
#endif
// NW_RD: depth of the weight ring (k-blocks of fragments in registers, NW_RD - 1 ahead of the MFMAs).  The tile is bound by its weight stream
// (1.8 MB per 16 rows through the CU's L1), and what a CU has in flight sets the rate it gets: eight waves x 3 blocks x 4 KB = 96 KB against
// ~0.8 us of L2 latency gave ~113 GB/s in the GEMM phases where the L1 fills at 154 (64 B/clk); the half engine's registers allow 7 blocks ahead
// (one whole GEMM: 218 registers), the three-piece engine stays at 3.
constexpr int NW_H = 256, NW_MT = 16, NW_LD = NW_H + 4 /* LDA(H), kernels_egnn.hip */, NW_RD = NW_NPL == 2 ? NW_RD_HALF : 4;
constexpr int NPL = NW_NPL;                  // pieces per operand: 3 (bf16 split, six MFMAs per product) or 2 (fp16 "half" engine, three)
constexpr unsigned KBS = 64u * NPL;          // 16-byte units per k-block of a 16-column tile in the packed split weight
#if NW_NPL == 3
typedef sbf16x8 wfrag;
#define NW_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)
#else
typedef sf16x8 wfrag;
#define NW_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)
#endif
__device__ __forceinline__ const void* nw_pack(const WPack& W) { return NPL == 3 ? W.ws16 : W.wh16; }
__device__ __forceinline__ float nw_scale(const WPack& W) { return NPL == 3 ? 1.0f : W.wh_scale; }
__device__ __forceinline__ float nw_inv(const WPack& W) { return NPL == 3 ? 1.0f : W.wh_inv; }

__device__ __forceinline__ void nw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct NwFrag { const wfrag* p; unsigned ns; };       // a wave's first 16-column tile (k-block kb0); ns = stride between n-tiles (16-byte units)
__device__ __forceinline__ NwFrag nw_frag(const void* Ws16, int kb32_total, int kb0, int nt0) {
    const int lane = threadIdx.x & 63;
    NwFrag f;
    f.p = reinterpret_cast<const wfrag*>(Ws16) + ((size_t)nt0 * kb32_total + kb0) * KBS + lane;
    f.ns = (unsigned)kb32_total * KBS;
    return f;
}
struct NwRing { wfrag b[NW_RD][2][NPL]; };              // ring of NW_RD k-blocks x [2 n-tiles][NPL pieces]
__device__ __forceinline__ void nw_load_set(const wfrag* q, unsigned ns, wfrag (&dst)[2][NPL]) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int s = 0; s < NPL; ++s) dst[n][s] = q[n * ns + s * 64];
}
// a GEMM enters with its k-blocks 0 .. NW_RD - 2 in sets 0 .. NW_RD - 2 and leaves with those of `next` there
__device__ __forceinline__ void nw_prefetch(const NwFrag& f, NwRing& c) {
#pragma unroll
    for (int i = 0; i < NW_RD - 1; ++i) nw_load_set(f.p + i * KBS, f.ns, c.b[i]);
}

// eight k-values of one row (two float4) -> the NPL fragments of the 16 x 16 x 32 MFMA
#if NW_NPL == 3
__device__ __forceinline__ void nw_split8(const float4& lo, const float4& hi, wfrag (&p)[NPL]) { split8(lo, hi, p[0], p[1], p[2]); }
#else
__device__ __forceinline__ void nw_split8(const float4& lo, const float4& hi, wfrag (&p)[NPL]) {
    {
        uint32_t a[4], b[4];
        split2_pair(lo.x, lo.y, a[0], b[0]); split2_pair(lo.z, lo.w, a[1], b[1]);
        split2_pair(hi.x, hi.y, a[2], b[2]); split2_pair(hi.z, hi.w, a[3], b[3]);
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4 va = {a[0], a[1], a[2], a[3]}, vb = {b[0], b[1], b[2], b[3]};
        p[0] = __builtin_bit_cast(wfrag, va); p[1] = __builtin_bit_cast(wfrag, vb);
    }
}
#endif

// acc[n] += A(lds fp32 image, 16 rows) x W_n^T over KB32 * 32 k-values for the wave's two 16-column tiles (tile_gemm_rsplit16 at half the width)
template <int KB32>
__device__ __forceinline__ void nw_gemm(const float* ldsA, const NwFrag cur, const NwFrag next, sf32x4 (&acc)[2], NwRing& ring) {
    static_assert(KB32 % NW_RD == 0 && (NW_RD == 4 || NW_RD == 8), "K must be a multiple of the ring's k-range");
    const int lane = threadIdx.x & 63;
    const float* ap = ldsA + (lane & 15) * NW_LD + (lane >> 4) * 4;
    float4 raw[2][2];
    wfrag a[2][NPL];
#define NW_LOADA(SET, PTR) { raw[SET][0] = *reinterpret_cast<const float4*>(PTR); raw[SET][1] = *reinterpret_cast<const float4*>((PTR) + 16); }
#define NW_SPLIT(DST, SET) nw_split8(raw[SET][0], raw[SET][1], a[DST]);
    // small terms first; the two n-tiles alternate so that consecutive MFMAs never share an accumulator
#define NW_MF(AS, AI, BS, BI) _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[n] = NW_MFMA(a[AS][AI], ring.b[BS][n][BI], acc[n]);
#if NW_NPL == 3
#define NW_MFMAS(AS, BS) NW_MF(AS, 2, BS, 0) NW_MF(AS, 1, BS, 1) NW_MF(AS, 0, BS, 2) NW_MF(AS, 1, BS, 0) NW_MF(AS, 0, BS, 1) NW_MF(AS, 0, BS, 0)
    // the next block's split (44 VALU operations) spread over this block's twelve MFMAs
#define NW_INTERLEAVE()                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 12; ++i) {                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); }
#else
#define NW_MFMAS(AS, BS) NW_MF(AS, 1, BS, 0) NW_MF(AS, 0, BS, 1) NW_MF(AS, 0, BS, 0)
    // the next block's split (16 VALU operations) spread over this block's six MFMAs
#define NW_INTERLEAVE()                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
#endif
    // block i: MFMAs on set i % NW_RD; set (i + NW_RD - 1) % NW_RD <- weight block i + NW_RD - 1 (or that block - KB32 of `next`);
    // raw[i % 2] <- A block i + 2; a[(i + 1) % 2] <- split of raw[(i + 1) % 2]
#define NW_BLOCK(I)                                                                                                         \
    {                                                                                                                       \
        const bool tail = kb + (I) + NW_RD - 1 >= KB32;              /* wave-uniform */                                     \
        const wfrag* q = tail ? next.p + (unsigned)(kb + (I) + NW_RD - 1 - KB32) * KBS : cur.p + (unsigned)(kb + (I) + NW_RD - 1) * KBS; \
        nw_load_set(q, tail ? next.ns : cur.ns, ring.b[((I) + NW_RD - 1) % NW_RD]);                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        if (kb + (I) + 1 < KB32) { NW_SPLIT(((I) + 1) & 1, ((I) + 1) & 1) }                                                 \
        NW_MFMAS((I) & 1, (I) % NW_RD)                                                                                      \
        NW_INTERLEAVE()                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        if (kb + (I) + 2 < KB32) { NW_LOADA((I) & 1, ap + (kb + (I) + 2) * 32) }                                            \
    }
    NW_LOADA(0, ap)
    NW_LOADA(1, ap + 32)
    NW_SPLIT(0, 0)
#pragma unroll 1
    for (int kb = 0; kb < KB32; kb += NW_RD) {
        NW_BLOCK(0) NW_BLOCK(1) NW_BLOCK(2) NW_BLOCK(3)
        if constexpr (NW_RD == 8) { NW_BLOCK(4) NW_BLOCK(5) NW_BLOCK(6) NW_BLOCK(7) }
    }
#undef NW_LOADA
#undef NW_SPLIT
#undef NW_MFMAS
#undef NW_MF
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
// out = acc * inv + bias (inv: the inverse of the weight pack's power-of-two scale; exact)
__device__ __forceinline__ void nw_store(const sf32x4 (&acc)[2], int wave, float* __restrict__ out, int row0, int nvalid, const NwCol* bias, float inv) {
    nw_foreach(acc, wave, [&](int row, int col, int n, float v) {
        if (row < nvalid) out[(size_t)(row0 + row) * NW_H + col] = __fmaf_rn(v, inv, bias ? bias->v[n] : 0.f);
    });
}
// SiLU(a / sc) for an accumulator carrying the scale sc (c1 = -log2(e) / sc): five operations, the bits of silu_f(a / sc)
__device__ __forceinline__ float nw_silu_scaled(float a, float c1, float sc) {
    const float u = __builtin_amdgcn_exp2f(a * c1);
    return a * __builtin_amdgcn_rcpf(__fmaf_rn(u, sc, sc));
}
// SAVE (training forward, half engine only): the tile also leaves agg / nf, the node MLP's pre-activation and activation and the new h in the
// activation store (TrainSave), and the packs' power-of-two scales come from the device (WPack::wh_dev: the packs are re-made every step)
template <bool SAVE>
__global__ __launch_bounds__(512) void k_node16w(Layout lay, Work w, Dims d, LayerW lw, LayerW lw_next, int layer, int has_next_arg, TrainSave sv) {
    __shared__ __attribute__((aligned(16))) float buf0[NW_MT * NW_LD];      // h (kept for the residual)
    __shared__ __attribute__((aligned(16))) float buf1[NW_MT * NW_LD];      // agg / nf -> T = SiLU(.) -> h_new
    const int has_next = has_next_arg & 1;                                   // (bits 1..29: the dead-tile threshold of the plane tiles, unused here)
    const bool skip_pc = ((has_next_arg >> 30) & 1) != 0;                   // not the last GCL of its block (inv_sublayers > 1): no P_c | Q_c
    const int tid = threadIdx.x, wave = tid >> 6;
    const int row0 = (int)blockIdx.x * NW_MT, nvalid = min(NW_MT, lay.N - row0);
    const bool want_pc = row0 < lay.Nm;                                      // the tile holds receivers that move
    constexpr int KB = NW_H / 32;
    // the chain of GEMMs of this tile; each one's last blocks fetch the next one's first fragments
    const NwFrag f3a = nw_frag(nw_pack(lw.W3), 2 * KB, 0, 2 * wave), f3b = nw_frag(nw_pack(lw.W3), 2 * KB, KB, 2 * wave);
    const NwFrag f4 = nw_frag(nw_pack(lw.W4), KB, 0, 2 * wave);
    const NwFrag fcp = nw_frag(nw_pack(lw.Wpq_c), KB, 0, 2 * wave), fcq = nw_frag(nw_pack(lw.Wpq_c), KB, 0, NW_H / 16 + 2 * wave);
    const NwFrag fnp = nw_frag(nw_pack(lw_next.Wpq_e), KB, 0, 2 * wave), fnq = nw_frag(nw_pack(lw_next.Wpq_e), KB, 0, NW_H / 16 + 2 * wave);
    NwRing ring;
    nw_prefetch(f3a, ring);
    NwCol b3v = nw_col(lw.b3, wave);
    const NwCol b4v = nw_col(lw.b4, wave), b6v = nw_col(lw.b6, wave), b1nv = nw_col(lw_next.b1, wave);
    float sc3 = nw_scale(lw.W3), inv3 = nw_inv(lw.W3), inv4 = nw_inv(lw.W4), invc = nw_inv(lw.Wpq_c), invn = nw_inv(lw_next.Wpq_e);
    if constexpr (SAVE) { sc3 = lw.W3.wh_dev[0]; inv3 = lw.W3.wh_dev[1]; inv4 = lw.W4.wh_dev[1]; invc = lw.Wpq_c.wh_dev[1]; invn = lw_next.Wpq_e.wh_dev[1]; }
    const float c13 = -1.4426950408889634f * inv3;
    b3v.v[0] *= sc3; b3v.v[1] *= sc3;                                      // the accumulators carry their weight pack's scale
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
            if (SAVE && r < nvalid) reinterpret_cast<float4*>(sv.aggn + ((size_t)sv.slot * lay.N + row0 + r) * NW_H)[c4] = v;
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
    nw_foreach(acc, wave, [&](int row, int col, int n, float v) {
        const float t = nw_silu_scaled(v + b3v.v[n], c13, sc3);
        buf1[row * NW_LD + col] = t;
        if (SAVE && row < nvalid) {
            const size_t o = ((size_t)sv.slot * lay.N + row0 + row) * NW_H + col;
            sv.pre3[o] = (v + b3v.v[n]) * inv3; sv.nact[o] = t;
        }
    });
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
            hn = buf0[row * NW_LD + col] + __fmaf_rn(v, inv4, b4v.v[n]);        // residual (egnn_new.py:57)
            w.h[(size_t)(row0 + row) * NW_H + col] = hn;
            if (SAVE) sv.h[((size_t)(sv.slot + 1) * lay.N + row0 + row) * NW_H + col] = hn;      // h entering block layer + 1
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
        nw_store(acc, wave, w.Pc, row0, nvalid, &b6v, invc);
    }
    if (!skip_pc) {
        nw_zero(acc);
        nw_gemm<KB>(buf1, fcq, has_next ? fnp : fcq, acc, ring);
        nw_store(acc, wave, w.Qc, row0, nvalid, nullptr, invc);
    }
    NSTAMP(5);
    if (has_next) {
        nw_zero(acc);
        nw_gemm<KB>(buf1, fnp, fnq, acc, ring);
        nw_store(acc, wave, w.P, row0, nvalid, &b1nv, invn);
        nw_zero(acc);
        nw_gemm<KB>(buf1, fnq, fnq, acc, ring);
        nw_store(acc, wave, w.Q, row0, nvalid, nullptr, invn);
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

