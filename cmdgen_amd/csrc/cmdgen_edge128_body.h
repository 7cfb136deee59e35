// cmdgen_edge128_body.h - the body of kernels_edge128.hip, included once per matrix engine (E128_NPL = 3: three bf16 pieces per operand,
// six MFMAs per product; 2: two fp16 pieces, three MFMAs - cmdgen_split.h) inside a namespace of its own.  No include guard on purpose.

constexpr int H = 256;
#ifndef CMDGEN_STAMP_COORD
#define CMDGEN_STAMP_COORD 0      // diagnostic builds: which of the two kernels records its phase stamps
#endif
// issue priority of a wave inside its GEMM quarters / everywhere else (s_setprio)
#ifndef CMDGEN_E128_GPRIO
#define CMDGEN_E128_GPRIO 1
#endif
#ifndef CMDGEN_E128_VPRIO
#define CMDGEN_E128_VPRIO 0
#endif
#ifndef CMDGEN_E128_AHEAD
#define CMDGEN_E128_AHEAD 1     // the first 64 rows of the next quarter are gathered before the GEMM over this one (half engine)
#endif
#ifndef CMDGEN_E128_WPS
#define CMDGEN_E128_WPS 2       // workgroups per CU the register allocation aims at (3 with 64-row tiles: diagnostic builds)
#endif
#ifndef CMDGEN_E128_MT
#define CMDGEN_E128_MT 128
#endif
#ifndef CMDGEN_E128_FUSED
#define CMDGEN_E128_FUSED 0     // 1 (half engine; set by kernels_edge128.hip): the build of quarter q + 1 is issued INSIDE the GEMM over quarter q (double-buffered planes; see x_main)
#endif
constexpr int MT = CMDGEN_E128_MT;      // rows per tile (at most): 128, 96 or 64
constexpr int MTL = 128;                // rows the per-tile index arrays hold (the index phase handles two rows per lane of wave 0)
constexpr int KQ = 64;                  // k-values per build / GEMM pass
constexpr int PLDA = KQ + 8;            // bf16 per plane row: 144 B, conflict-free ds_read_b128 over 16 consecutive rows
constexpr int PE = MT * PLDA;           // bf16 per plane
constexpr int NPL = E128_NPL;           // pieces per operand = planes of the A image: 3 (bf16 split, six MFMAs per product) or 2 (fp16 "half" engine, three)
constexpr unsigned KBS = 64u * NPL;     // 16-byte units per k-block of a 32-column tile in the packed split weight
constexpr unsigned NS = 16u * KBS;      // ... between the two 32-column tiles of a wave ([H][H] weight: 16 k-blocks)
constexpr bool FUSED = E128_NPL == 2 && CMDGEN_E128_FUSED != 0 && CMDGEN_E128_MT == 128;
constexpr int PLF = 128 * KQ;           // fused form: fp16 per plane - rows of 128 B, unpadded, 16-byte chunks XOR-swizzled by (row & 7); planes[buffer 2][piece 2][PLF]
#if E128_NPL == 3
typedef sbf16x8 efrag;
#define E128_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#else
typedef sf16x8 efrag;
#define E128_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#endif

struct alignas(16) EdgeRec { int row, col; float r, d0; };

struct alignas(16) E128Lds {
    unsigned short planes[FUSED ? 4 * PLF : NPL * PE + 64];   // the NPL planes (bf16 / fp16 pieces) of the quarter in flight (+ the A prefetch's overshoot); fused form: two quarters (64 KB)
    EdgeRec e[MTL];                              // (receiver, sender, radial, d0) of the tile's rows; -1 / -1 / 0 / 0 beyond its end
    float cd[MTL][4];                            // coordinate kernel: coord_diff of the row, later coord_diff * tanh(phi) * range
    float part[4][MTL];                          // the four waves' partial row dots
    float gw[4][MTL];                            // gate (message kernel) / tanh(phi) * range (coordinate kernel) of each row, one copy per wave (each wave fills and reads its own: no barrier)
    float wrd[2 * H];                           // radial / d0 columns of the first layer
    float colv[2 * H];                          // per-column constants of the epilogue: bias of the second layer (times the engine's scale), the row dot's weight vector
    int segrow[MTL];                             // receiver of each segment of the tile
    int segstart[MTL + 1];                       // first row of each segment (coordinate kernel)
    unsigned char seg[MTL];                      // segment index of each row (255 beyond the tile's end)
    int meta[4];                                // [0] segments, [1] live, [2] rows of the tile
    int smask[4];                               // bit e: row e of the tile opens a segment (message kernel); bit ne: end of the listed rows
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// lazily updated positions, as kernels_egnn.hip forms them (same expression, same bits).  Branch-free: the loads of a pocket node and of a
// moving node are the same instructions with other addresses (a pocket node adds a zero), so the four position loads of an edge leave together
// instead of one divergent path after the other (each with its own round trip).
__device__ __forceinline__ float4 pos_lazy(const Layout& lay, const Work& w, const Dims& d, int n, int layer) {
    const bool mov = n < lay.Nm;
    if (layer == 0) return *(mov ? w.X0 + n : w.XP + (n - lay.Nl));
    const float4* pp = mov ? (layer == 1 ? w.X0 + n : w.XL + ((size_t)(layer - 1) * lay.Nm + n)) : w.XP + (n - lay.Nl);
    const float4 p = *pp;
    float4 a = w.ACC[(size_t)(layer - 1) * lay.Nm + (mov ? n : 0)];
    const float dv = d.agg_mean ? w.adiv[mov ? n : 0] : d.norm_factor;
    if (!mov) return p;
    return make_float4(p.x + a.x / dv, p.y + a.y / dv, p.z + a.z / dv, 0.f);
}
__device__ __forceinline__ float4 pos_mat(const Layout& lay, const Work& w, int n, int layer) {
    if (n >= lay.Nm) return w.XP[n - lay.Nl];
    return layer == 0 ? w.X0[n] : w.XL[(size_t)layer * lay.Nm + n];
}

// ---- tile build: columns [64 q, 64 q + 64) of SiLU(P[row] + Q[col] + w_r r + w_d d0) as three bf16 planes, in two batches of 64 rows
// (8 gathered float4 per thread and batch: with the 128 accumulators and the weight fragments a whole tile's 16 would spill).  The
// gather of a batch and its use are separate calls so that a batch can be in flight during the GEMM over the previous quarter.
// Rows beyond the tile's end repeat its last row (finite values in rows nobody reads; no divergent code).  Thread -> 16 bytes of a quarter
// row (16 lanes per row, 4 consecutive rows per wave instruction: edges of one receiver share their P row's cache lines).
// The rows are addressed through buffer descriptors (base in scalar registers, one 32-bit byte offset per lane: ONE vector instruction per
// address, row << 10 | column bytes, instead of the three 64-bit adds of a flat pointer - 30 loads per thread and quarter).
typedef unsigned gq4 __attribute__((ext_vector_type(4)));
struct Gath { gq4 p[4], q[4]; };
struct RowBufs { __amdgpu_buffer_rsrc_t P, Q; };
__device__ __forceinline__ RowBufs row_bufs(const float* P, const float* Q, int n_rows) {
    RowBufs b;
    const int bytes = n_rows * (H * 4);                                  // (< 2 GiB: 2M nodes of 1 KiB rows)
    b.P = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, bytes, 0x00020000);
    b.Q = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Q), 0, bytes, 0x00020000);
    return b;
}
template <int NMT>
__device__ __forceinline__ void gather_half(const E128Lds& L, const int tid, const int q, const int half, const int ne, const RowBufs& rb, Gath& g) {
    const int c4 = tid & 15, rsub = tid >> 4;
    const unsigned cofs = (unsigned)(q * KQ + 4 * c4) * 4u;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps)
        if (half * 64 + ps * 16 < 32 * NMT) {
            const int e = min(half * 64 + ps * 16 + rsub, ne - 1);
            const int2 rc = *reinterpret_cast<const int2*>(&L.e[e]);
            g.p[ps] = __builtin_amdgcn_raw_buffer_load_b128(rb.P, (int)(((unsigned)rc.x << 10) + cofs), 0, 0);
            g.q[ps] = __builtin_amdgcn_raw_buffer_load_b128(rb.Q, (int)(((unsigned)rc.y << 10) + cofs), 0, 0);
        }
}
template <int NMT>
__device__ __forceinline__ void store_half(E128Lds& L, const int tid, const int q, const int half, const int ne, const Gath& g) {
    const int c4 = tid & 15, rsub = tid >> 4;
    const int col = q * KQ + 4 * c4;
    const float4 wr4 = *reinterpret_cast<const float4*>(L.wrd + col), wd4 = *reinterpret_cast<const float4*>(L.wrd + H + col);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps)
        if (half * 64 + ps * 16 < 32 * NMT) {
            const int e = half * 64 + ps * 16 + rsub;
            const float2 rd = *reinterpret_cast<const float2*>(&L.e[min(e, ne - 1)].r);
            const float r = rd.x, d0 = rd.y;
#define G_F(V, I) __uint_as_float(V[ps][I])
            const float4 a = make_float4(silu_f(G_F(g.p, 0) + G_F(g.q, 0) + wr4.x * r + wd4.x * d0), silu_f(G_F(g.p, 1) + G_F(g.q, 1) + wr4.y * r + wd4.y * d0),
                                         silu_f(G_F(g.p, 2) + G_F(g.q, 2) + wr4.z * r + wd4.z * d0), silu_f(G_F(g.p, 3) + G_F(g.q, 3) + wr4.w * r + wd4.w * d0));
#undef G_F
            if constexpr (NPL == 3) split_store4(L.planes, PE, e * PLDA + 4 * c4, a); else split_store4_half(L.planes, PE, e * PLDA + 4 * c4, a);
        }
}

// one 16-byte weight fragment: descriptor base (the wave's first 32-column tile) + lane * 16 + a scalar byte offset: no address arithmetic
// in vector registers, one register (lane * 16) for every weight address of the kernel
__device__ __forceinline__ efrag w_frag(const __amdgpu_buffer_rsrc_t rw, const int lane, const int soff) {
    return __builtin_bit_cast(efrag, __builtin_amdgcn_raw_buffer_load_b128(rw, lane << 4, soff, 0));
}

// ---- one quarter of the tile product: acc[m][n] += planes(rows 32 m .., k-blocks 4 q .. 4 q + 3) x W^T for m < NMT.
// bs[0] holds the weight fragments of k-block 4 q on entry and of k-block 4 q + 4 on exit (q < 3).  One load pinned beside every pair of MFMAs; per accumulator the six products of a k-block keep the
// order of cmdgen_split.h (small terms first).  wb: the wave's first 32-column tile, k-block 0, this lane.
template <int NMT>
__device__ __forceinline__ void gemm_quarter(const unsigned short* planes, const int lane, const int q, const __amdgpu_buffer_rsrc_t rw,
                                             sf32x16 (&acc)[NMT][2], efrag (&bs)[2][2][NPL]) {
    const unsigned short* ap = planes + (lane & 31) * PLDA + (lane >> 5) * 8;
    efrag a[2][NPL];
#pragma unroll
    for (int s = 0; s < NPL; ++s) a[0][s] = *reinterpret_cast<const efrag*>(ap + s * PE);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        const int qn = (int)(((unsigned)((4 * q + kq + 1) & 15) * KBS) << 4);    // byte offset of the next k-block in the wave's weight tiles (scalar)
        constexpr int NBL = 2 * NPL;                                            // weight loads per k-block: (n, s) = (i & 1, i >> 1)
        constexpr int BPG = (NBL + NMT - 1) / NMT;                              // ... per row group: all of them early in the k-block
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
            const int it = kq * NMT + m, cs = it & 1, nx = cs ^ 1, bc = kq & 1, bn = bc ^ 1;
            const bool more_a = (m + 1 < NMT) || (kq < 3);
            const bool more_b = kq < 3 || q < 3;                                // wave-uniform: not past the tile's last k-block
            const unsigned short* an = ap + ((m + 1 < NMT) ? (m + 1) * 32 * PLDA + kq * 16 : (kq + 1) * 16);
            const int b0 = m * BPG;                                             // first weight load of this group
#define E_MF(N, AI, BI) acc[m][N] = E128_MFMA(a[cs][AI], bs[bc][N][BI], acc[m][N]);
#define E_LA(S) if (more_a) a[nx][S] = *reinterpret_cast<const efrag*>(an + (S) * PE);
#define E_LB(I) if ((I) < NBL && (I) >= b0 && (I) < b0 + BPG && more_b) bs[bn][(I) & 1][(I) >> 1] = w_frag(rw, lane, qn + (int)((((unsigned)((I) & 1) * NS + (unsigned)((I) >> 1) * 64u)) << 4));
            if constexpr (NPL == 3) {
                E_LA(2) E_MF(0, 2, 0) E_MF(1, 2, 0) __builtin_amdgcn_sched_barrier(0);
                E_LA(1) E_MF(0, 1, 1) E_MF(1, 1, 1) __builtin_amdgcn_sched_barrier(0);
                E_LA(0) E_MF(0, 0, 2) E_MF(1, 0, 2) __builtin_amdgcn_sched_barrier(0);
                E_LB(b0) E_LB(b0 + 3) E_MF(0, 1, 0) E_MF(1, 1, 0) __builtin_amdgcn_sched_barrier(0);
                E_LB(b0 + 1) E_LB(b0 + 4) E_MF(0, 0, 1) E_MF(1, 0, 1) __builtin_amdgcn_sched_barrier(0);
                E_LB(b0 + 2) E_LB(b0 + 5) E_MF(0, 0, 0) E_MF(1, 0, 0) __builtin_amdgcn_sched_barrier(0);
            } else {
                // half engine: a1 b0 + a0 b1 + a0 b0 (small terms first)
                E_LA(1) E_LB(b0 + 2) E_MF(0, 1, 0) E_MF(1, 1, 0) __builtin_amdgcn_sched_barrier(0);
                E_LA(0) E_LB(b0) E_LB(b0 + 3) E_MF(0, 0, 1) E_MF(1, 0, 1) __builtin_amdgcn_sched_barrier(0);
                E_LB(b0 + 1) E_MF(0, 0, 0) E_MF(1, 0, 0) __builtin_amdgcn_sched_barrier(0);
            }
#undef E_MF
#undef E_LA
#undef E_LB
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 6 - the fused main loop of a tile (half engine): the BUILD of quarter q + 1 is issued inside the GEMM over quarter q.
// The kernel was not issue-bound (12 % fewer vector instructions bought 1 %, profiles/r06_b): a workgroup's tile is a chain of phases - index,
// 4 x (build, GEMM), epilogue - that a wave walks one after the other, and with two waves per SIMD the partner hides only part of it.  Here a
// wave's vector work for the next quarter rides in the shadow of its own MFMAs: the planes are double-buffered (unpadded 128-byte rows whose
// 16-byte chunks are XOR-swizzled by row & 7: 2 x 32 KB, conflict-free b64 writes and b128 reads), a thread's share of a quarter - 2 NMT "slots" of
// 4 columns x 16 rows apart - is cut into six pieces per slot, one piece behind each pair of MFMAs (two (m, k-block) groups = six pairs per slot:
// <= 8 vector instructions per pair, at most four of them v_exp / v_rcp), gathers are requested three slots ahead (24 registers instead of the 64 a
// whole quarter needs), and a quarter costs ONE barrier (the next quarter's planes complete + this quarter's planes free) instead of two.
// Only quarter 0 of a tile is built without MFMAs beside it.
#ifndef CMDGEN_E128_RING
#define CMDGEN_E128_RING 2      // slots of gathered rows in flight (2: no spill inside the loop; 3 and 4 spill and measure 2-3 % slower, profiles/r06_c)
#endif
template <int NMT> struct XPipe {
    static constexpr int SL = 2 * NMT;          // slots per quarter
    static constexpr int R = NMT == 1 ? 2 : CMDGEN_E128_RING;                       // ring of slots in flight = how many slots ahead a gather is requested (indexed by the tile-wide slot number)
    gq4 p[R], q[R]; float r[R], d0[R];          // gathered P / Q chunks and the row's (radial, d0)
    int4 rec;                                   // (row, col, r, d0) of the slot after them
    float4 wr, wd;                              // the first layer's radial / d0 weights of this thread's four columns, for the quarter being built
    float v[4], e[4];
};
// (qb, sl): quarter being built and slot inside it - compile-time constants after unrolling (the ring index is the tile-wide slot number mod R)
template <int NMT>
__device__ __forceinline__ void x_rec(XPipe<NMT>& X, const E128Lds& L, const int tid, const int sl, const int ne) {
    const int e = min(16 * (sl % XPipe<NMT>::SL) + (tid >> 4), ne - 1);
    X.rec = *reinterpret_cast<const int4*>(&L.e[e]);
}
template <int NMT>
__device__ __forceinline__ void x_issue(XPipe<NMT>& X, const RowBufs& rb, const int tid, const int qb, const int sl) {
    constexpr int R = XPipe<NMT>::R;
    const unsigned cofs = (unsigned)(qb * KQ + 4 * (tid & 15)) * 4u;
    const int k = (qb * XPipe<NMT>::SL + sl) % R;
    X.p[k] = __builtin_amdgcn_raw_buffer_load_b128(rb.P, (int)(((unsigned)X.rec.x << 10) + cofs), 0, 0);
    X.q[k] = __builtin_amdgcn_raw_buffer_load_b128(rb.Q, (int)(((unsigned)X.rec.y << 10) + cofs), 0, 0);
    X.r[k] = __int_as_float(X.rec.z); X.d0[k] = __int_as_float(X.rec.w);
}
template <int NMT>
__device__ __forceinline__ void x_cols(XPipe<NMT>& X, const E128Lds& L, const int tid, const int q) {
    const int col = q * KQ + 4 * (tid & 15);
    X.wr = *reinterpret_cast<const float4*>(L.wrd + col); X.wd = *reinterpret_cast<const float4*>(L.wrd + H + col);
}
// piece 0..5 of slot sl of quarter qb; wofs: this thread's (swizzled) byte offset inside a 16-row group of a plane
template <int NMT>
__device__ __forceinline__ void x_piece(XPipe<NMT>& X, E128Lds& L, const RowBufs& rb, const int tid, const int qb, const int sl, const int piece, const int ne, const int wofs) {
    constexpr int SL = XPipe<NMT>::SL, R = XPipe<NMT>::R;
    const int k = (qb * SL + sl) % R;
    const float wr[4] = {X.wr.x, X.wr.y, X.wr.z, X.wr.w}, wd[4] = {X.wd.x, X.wd.y, X.wd.z, X.wd.w};
    if (piece == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) X.v[i] = __fmaf_rn(wr[i], X.r[k], __uint_as_float(X.p[k][i]) + __uint_as_float(X.q[k][i]));
    } else if (piece == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { X.v[i] = __fmaf_rn(wd[i], X.d0[k], X.v[i]); X.e[i] = X.v[i] * -1.4426950408889634f; }
    } else if (piece == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) X.e[i] = __builtin_amdgcn_exp2f(X.e[i]);
    } else if (piece == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) X.e[i] = 1.0f + X.e[i];
        X.e[0] = __builtin_amdgcn_rcpf(X.e[0]); X.e[1] = __builtin_amdgcn_rcpf(X.e[1]);
    } else if (piece == 4) {
        X.e[2] = __builtin_amdgcn_rcpf(X.e[2]); X.e[3] = __builtin_amdgcn_rcpf(X.e[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) X.v[i] *= X.e[i];                     // silu_f's operations, in its order
    } else {
        uint32_t a0, a1, b0, b1;
        split2_pair(X.v[0], X.v[1], a0, a1); split2_pair(X.v[2], X.v[3], b0, b1);
        unsigned char* dst = reinterpret_cast<unsigned char*>(L.planes) + (qb & 1) * (4 * PLF) + sl * (16 * KQ * 2) + wofs;
        *reinterpret_cast<uint2*>(dst) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(dst + 2 * PLF) = make_uint2(a1, b1);
        // the ring entry of this slot was consumed in pieces 0 / 1: request the slot R ahead into it, read the record of the one after
        if (sl + R < SL) x_issue<NMT>(X, rb, tid, qb, sl + R); else if (qb < 3) x_issue<NMT>(X, rb, tid, qb + 1, sl + R - SL);
        if (sl + R + 1 < SL || qb < 3) x_rec<NMT>(X, L, tid, (sl + R + 1) % SL, ne);
        if (sl == SL - 1 && qb < 3) x_cols<NMT>(X, L, tid, qb + 1);
    }
}

// the A fragment (16-byte chunk 2 kq + lane / 32 of row 32 m + lane % 32) of piece s in plane buffer b: a0 = this lane's byte offset for kq = 0
__device__ __forceinline__ efrag x_afrag(const E128Lds& L, const int a0, const int b, const int s, const int m, const int kq) {
    return *reinterpret_cast<const efrag*>(reinterpret_cast<const unsigned char*>(L.planes) + b * (4 * PLF) + s * (2 * PLF) + m * (32 * KQ * 2) + (a0 ^ (kq << 5)));
}

template <int NMT>
__device__ __forceinline__ void x_main(E128Lds& L, const int tid, const int lane, const int ne, const RowBufs& rb, const __amdgpu_buffer_rsrc_t rw,
                                       sf32x16 (&acc)[NMT][2], efrag (&bs)[2][2][NPL]) {
    constexpr int SL = XPipe<NMT>::SL;
    XPipe<NMT> X;
    const int rsub = tid >> 4, c4 = tid & 15;
    const int wofs = (rsub * KQ + ((((c4 >> 1) ^ (rsub & 7)) << 3) | ((c4 & 1) << 2))) * 2;
    const int a0 = ((lane & 31) * KQ + ((((lane >> 5) ^ (lane & 7)) & 7) << 3)) * 2;
    // prologue: the first R slots requested, the next one's record read; then quarter 0 is built with nothing beside it
    constexpr int R = XPipe<NMT>::R;
    x_cols<NMT>(X, L, tid, 0);
#pragma unroll
    for (int sl = 0; sl < R; ++sl) { x_rec<NMT>(X, L, tid, sl, ne); x_issue<NMT>(X, rb, tid, 0, sl); }
    x_rec<NMT>(X, L, tid, R % SL, ne);
#pragma unroll
    for (int sl = 0; sl < SL; ++sl)
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) x_piece<NMT>(X, L, rb, tid, 0, sl, pc, ne, wofs);
    lds_barrier();
    // (the four quarters are unrolled: the waits for gathers and weight fragments stay counted across quarter boundaries - behind a loop's back edge
    // the compiler waits for everything in flight - and no piece needs a branch on "is there a next quarter")
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = q & 1;
        __builtin_amdgcn_s_setprio(CMDGEN_E128_GPRIO);
        efrag a[2][NPL];
#pragma unroll
        for (int s = 0; s < NPL; ++s) a[0][s] = x_afrag(L, a0, b, s, 0, 0);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const int qn = (int)(((unsigned)((4 * q + kq + 1) & 15) * KBS) << 4);        // byte offset of the next k-block in the wave's weight tiles (scalar)
            constexpr int NBL = 2 * NPL, BPG = (NBL + NMT - 1) / NMT;
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                const int g = kq * NMT + m, cs = g & 1, nx = cs ^ 1, bc = kq & 1, bn = bc ^ 1;
                const bool more_a = (m + 1 < NMT) || (kq < 3);
                const bool more_b = kq < 3 || q < 3;
                const int nm = (m + 1 < NMT) ? m + 1 : 0, nkq = (m + 1 < NMT) ? kq : kq + 1;
                const int b0 = m * BPG;
                // the slot of quarter q + 1 whose pieces ride behind this group's MFMA pairs (q is a run-time value: the slot index is quarter-relative)
                const int sl = g >> 1, pc0 = 3 * (g & 1);
#define X_PC(I) if (q < 3) x_piece<NMT>(X, L, rb, tid, q + 1, sl, pc0 + (I), ne, wofs);
#define E_MF(N, AI, BI) acc[m][N] = E128_MFMA(a[cs][AI], bs[bc][N][BI], acc[m][N]);
#define E_LA(S) if (more_a) a[nx][S] = x_afrag(L, a0, b, S, nm, nkq);
#define E_LB(I) if ((I) < NBL && (I) >= b0 && (I) < b0 + BPG && more_b) bs[bn][(I) & 1][(I) >> 1] = w_frag(rw, lane, qn + (int)((((unsigned)((I) & 1) * NS + (unsigned)((I) >> 1) * 64u)) << 4));
                E_LA(1) E_LB(b0 + 2) X_PC(0) E_MF(0, 1, 0) E_MF(1, 1, 0) __builtin_amdgcn_sched_barrier(0);
                E_LA(0) E_LB(b0) E_LB(b0 + 3) X_PC(1) E_MF(0, 0, 1) E_MF(1, 0, 1) __builtin_amdgcn_sched_barrier(0);
                E_LB(b0 + 1) X_PC(2) E_MF(0, 0, 0) E_MF(1, 0, 0) __builtin_amdgcn_sched_barrier(0);
#undef X_PC
#undef E_MF
#undef E_LA
#undef E_LB
            }
        }
        __builtin_amdgcn_s_setprio(CMDGEN_E128_VPRIO);
        lds_barrier();                        // quarter q + 1's planes are complete, quarter q's are free
    }
}

// Sum over the 32 lanes of a half wave of 32 values per lane, by value halving: after five exchanges lane l of a half holds the complete
// sum of value l & 31 (31 exchanges instead of 160 for a butterfly on every value).  Round 6: no LDS round trips and no selects around the
// exchange - the first stage is v_permlane16_swap_b32 (rows of 16 lanes: the odd rows of one register against the even rows of the other,
// i.e. exactly "keep your half of the values, hand over the other half") + one add per pair; the other four stages add a DPP-permuted copy
// (row_mirror, row_half_mirror, two quad permutations: the partner differs in the stage's lane bit, and over the five stages the partners
// cover all 32 lanes) and keep one of the two sums by that lane bit.  65 vector instructions per 32 values (93 + 31 ds_swizzle before).
__device__ __forceinline__ float reduce32_over32(float (&v)[32], const int lane) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const u2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 16]), false, false);
        v[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);        // lanes with bit 4 clear: value i of lanes l, l + 16; bit 4 set: value i + 16
    }
#define E_DPP(V, CTRL) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(V), CTRL, 0xf, 0xf, true))
#define E_STAGE(N, BIT, CTRL) {                                                                                   \
        const bool up = (lane & (BIT)) != 0;                                                                      \
        _Pragma("unroll") for (int i = 0; i < (N) / 2; ++i) {                                                     \
            const float lo = v[i] + E_DPP(v[i], CTRL), hi = v[i + (N) / 2] + E_DPP(v[i + (N) / 2], CTRL);         \
            v[i] = up ? hi : lo; } }
    E_STAGE(16, 8, 0x140) E_STAGE(8, 4, 0x141) E_STAGE(4, 2, 0x4e) E_STAGE(2, 1, 0xb1)
#undef E_STAGE
#undef E_DPP
    return v[0];
}

// SiLU(acc / sc) for an accumulator that carries the weight pack's power-of-two scale sc (c1 = -log2(e) / sc): the same five operations as
// silu_f - sc (1 + u) is formed by one fma, and 1 / (sc y) = (1 / y) / sc exactly - and the same bits as silu_f(acc / sc)
__device__ __forceinline__ float silu_scaled(float a, float c1, float sc) {
    const float u = __builtin_amdgcn_exp2f(a * c1);
    return a * __builtin_amdgcn_rcpf(__fmaf_rn(u, sc, sc));
}

// accumulator register r of row tile m -> row of the tile
#define E_ROW(M, R) ((M) * 32 + ((R) & 3) + 8 * ((R) >> 2) + 4 * (lane >> 5))

// ------------------------------------------------------------------------------------------------------------------------------
// One tile of NMT x 32 rows (ne of them listed) after its index phase: K in four build -> GEMM passes, then the epilogue in registers.
struct TileCtx {
    RowBufs rb; __amdgpu_buffer_rsrc_t rw;     // P / Q rows; the wave's two 32-column tiles of the split weight
    float ba0;
    float c1, sc;                   // SiLU of a scaled accumulator: x = acc / sc;  c1 = -log2(e) / sc
    int colw, layer;
};
template <bool COORD, int NMT>
__device__ __forceinline__ void tile_compute(E128Lds& L, const Layout& lay, const Work& w, const Dims& d, const TileCtx& c, const int ne,
                                             efrag (&bs)[2][2][NPL], unsigned long long (&st_)[8], unsigned long long& st_t) {
    int tid = threadIdx.x & 255;            // thread of the tile's four waves (the phase-locked driver runs two tiles per workgroup)
    asm volatile("" : "+v"(tid));           // opaque: per-lane addresses derived from it are recomputed per tile instead of being hoisted out of the tile loop and spilled
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const RowBufs rb = c.rb; const __amdgpu_buffer_rsrc_t rw = c.rw;
    const float ba0 = c.ba0, c1 = c.c1, sc = c.sc;
    const int colw = 64 * wave + (lane & 31);
    const int layer = c.layer;
#if CMDGEN_STAMPS == 6
#define STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_t; st_t = n_; } while (0)
#define STAMPB(i) do {} while (0)
#elif CMDGEN_STAMPS == 9      // second diagnostic mode: work and wait of the index phase and of the epilogue's pieces (tools/e128_stamps.py --mode 9)
#define STAMPB(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_t; st_t = n_; } while (0)
#define STAMP(i) do {} while (0)
#else
#define STAMP(i) do {} while (0)
#define STAMPB(i) do {} while (0)
#endif
    sf32x16 acc[NMT][2];                                                                // start from the bias of the layer (b2 / b7; staged in LDS: not a register held across the tile loop)
    if constexpr (NPL == 2) {
        // half engine (round 6): every accumulator tile starts as ONE MORE MFMA - the rank-1 product (column of 256s) x (row of bias pieces) with
        // C = 0: the B fragment holds, in its k = 0 / k = 1 slots, the two fp16 pieces of bias * scale / 256 of the lane's column (lanes 0-31; zero
        // elsewhere), the A fragment 256 in the same two slots.  8 MFMAs per tile instead of 128 v_mov_b32 per lane.
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 onesu = {lane < 32 ? 0x5c005c00u : 0u, 0u, 0u, 0u};                   // fp16 256.0 at k = 0 and k = 1 of every row
        const efrag ones = __builtin_bit_cast(efrag, onesu);
        const sf32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            uint32_t p0, p1;
            split2_pair(L.colv[colw + 32 * n] * 0.00390625f, 0.f, p0, p1);             // two fp16 pieces of bias * scale / 256 (low halves; the high halves are zero)
            const u4 v = {lane < 32 ? ((p0 & 0xffffu) | (p1 << 16)) : 0u, 0u, 0u, 0u};
            const efrag cb = __builtin_bit_cast(efrag, v);
#pragma unroll
            for (int m = 0; m < NMT; ++m) acc[m][n] = E128_MFMA(ones, cb, zero);
        }
    } else {
        const float bias0 = L.colv[colw], bias1 = L.colv[colw + 32];
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[m][0][r] = bias0; acc[m][1][r] = bias1; }
    }
    // ---------------- four build -> GEMM passes over a quarter of K each.  The first 64 rows of the NEXT quarter are gathered before the GEMM
    // over this one and land during it (32 registers; the half engine's GEMM leaves them - on the three-piece bf16 split they spilled, profiles/r04_c);
    // the other 64 rows are requested at the start of the build and land while the first batch is turned into planes.
    constexpr bool AHEAD = NPL == 2 && CMDGEN_E128_AHEAD != 0;
    if constexpr (FUSED) {
        x_main<NMT>(L, tid, lane, ne, rb, rw, acc, bs);
    } else {
    Gath g0;
    if constexpr (AHEAD) gather_half<NMT>(L, tid, 0, 0, ne, rb, g0);
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        if constexpr (!AHEAD) gather_half<NMT>(L, tid, q, 0, ne, rb, g0);
        if constexpr (NMT > 2) {
            Gath g1;
            gather_half<NMT>(L, tid, q, 1, ne, rb, g1);
            store_half<NMT>(L, tid, q, 0, ne, g0);
            store_half<NMT>(L, tid, q, 1, ne, g1);
        } else {
            store_half<NMT>(L, tid, q, 0, ne, g0);
        }
        STAMP(6);
        lds_barrier();
        STAMP(1);
        if constexpr (AHEAD) if (q < 3) gather_half<NMT>(L, tid, q + 1, 0, ne, rb, g0);
        __builtin_amdgcn_s_setprio(CMDGEN_E128_GPRIO);
        gemm_quarter<NMT>(L.planes, lane, q, rw, acc, bs);
        __builtin_amdgcn_s_setprio(CMDGEN_E128_VPRIO);
        STAMP(7);
        lds_barrier();                                                                  // every wave is done reading the planes
        STAMP(2);
    }
    }
    STAMPB(2);
    // ---------------- epilogue in registers: SiLU, the row dot (attention logit / coord_mlp.4)
    const float hv0 = L.colv[H + colw], hv1 = L.colv[H + colw + 32];
#pragma unroll
    for (int mh = 0; mh < (NMT + 1) / 2; ++mh) {          // two row tiles (32 values per lane) at a time
        float pl[32];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 2 * mh + mm;
                float dotp = 0.f;
                if (m < NMT) {
                    const float v0 = silu_scaled(acc[m < NMT ? m : 0][0][r], c1, sc), v1 = silu_scaled(acc[m < NMT ? m : 0][1][r], c1, sc);
                    acc[m < NMT ? m : 0][0][r] = v0; acc[m < NMT ? m : 0][1][r] = v1;
                    dotp = v0 * hv0 + v1 * hv1;
                }
                pl[mm * 16 + r] = dotp;
            }
        const float o = reduce32_over32(pl, lane);
        // value 16 mm + r = lane & 31
        const int l5 = lane & 31;
        L.part[wave][E_ROW(2 * mh + (l5 >> 4), l5 & 15)] = o;
    }
    STAMPB(3);
    lds_barrier();
    STAMP(3);
    STAMPB(4);
    // every wave forms the gates of all rows for itself (two rows per lane) and keeps them in its own LDS strip: no barrier before their use
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = 64 * u + lane;
        if (e < 32 * NMT) {
            const float s = (L.part[0][e] + L.part[1][e]) + (L.part[2][e] + L.part[3][e]);
            L.gw[wave][e] = COORD ? (d.use_tanh ? tanhf(s) * d.coords_range : s) : (d.attention ? sigmoid_f(s + ba0) : 1.0f);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(5);
    STAMPB(5);
    if constexpr (COORD) {
        // ordered segment sums of the three components: one thread per (segment, component), rows in list order
        const int nseg = L.meta[0];
        for (int i = tid; i < 3 * nseg; i += 256) {
            const int sgi = i / 3, comp = i - 3 * sgi;
            const int rb = L.segstart[sgi], re = L.segstart[sgi + 1];
            float sum = 0.f;
            for (int e = rb; e < re; ++e) sum += L.cd[e][comp] * L.gw[wave][e];     // trans = coord_diff * tanh(phi) * range, summed in list order (egnn_new.py:94-96)
            float* dst = reinterpret_cast<float*>(w.ACC + (size_t)layer * lay.Nm + L.segrow[sgi]) + comp;
            if (sgi == 0 || sgi == nseg - 1) atomicAdd(dst, sum); else *dst = sum;      // a receiver may continue in the neighbouring tiles; ACC is zero before the launch
        }
    } else {
        // gated messages and their ordered segment sum by receiver, in registers.  v_permlane32_swap_b32 on the register pair (column tile 0,
        // column tile 1) of an accumulator row group turns the 32 x 32 layout (lane half = rows +0 / +4) into one where EVERY lane of a
        // register holds the same row (lane = column 64 wave + lane): the tile's rows are then visited in list order by wave-uniform code, a
        // receiver's sum is a chain of 64-lane adds of the gated messages in ascending sender order like the reference's CPU scatter_add_
        // (egnn_new.py:283), and a finished receiver leaves as one 256-byte row segment.  Segment starts are a 128-bit scalar mask.
        const int nseg = L.meta[0];
        const unsigned sm[4] = {(unsigned)__builtin_amdgcn_readfirstlane(L.smask[0]), (unsigned)__builtin_amdgcn_readfirstlane(L.smask[1]),
                                (unsigned)__builtin_amdgcn_readfirstlane(L.smask[2]), (unsigned)__builtin_amdgcn_readfirstlane(L.smask[3])};
        const int segrow_v = L.segrow[lane & 31];                       // receiver of segment (lane & 31); read per finished segment with v_readlane
        float* const aggc = w.agg + 64 * wave + lane;
        float sum = 0.f;
        int sg = 0;                                                      // wave-uniform: the segment being summed
        auto flush = [&]() {
            if (sg < nseg) {
                float* dst = aggc + (size_t)__builtin_amdgcn_readlane(segrow_v, sg) * H;
                if (sg == 0 || sg == nseg - 1) atomicAdd(dst, sum);     // the receiver may continue in the neighbouring tiles
                else *dst = sum;                                         // agg is zero between blocks
            }
            ++sg; sum = 0.f;
        };
        // Round 6: the gate multiply rides in the scan's add (one v_fma per element instead of v_mul + v_add): after the swap a register holds ONE
        // row, whose gate is the same in every lane - a broadcast read of four consecutive rows' gates (one ds_read_b128 per four rows, as the
        // separate gating pass needed), requested one row group ahead so that the scan's scalar branches never wait for it.
        auto gates_of = [&](int mj, float4& gx, float4& gy) {
            const int base = 32 * (mj >> 2) + 8 * (mj & 3);
            gx = *reinterpret_cast<const float4*>(&L.gw[wave][base]);
            gy = *reinterpret_cast<const float4*>(&L.gw[wave][base + 4]);
        };
        float4 gxn, gyn;
        gates_of(0, gxn, gyn);
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 gx = gxn, gy = gyn;
                if (4 * m + j + 1 < 4 * NMT) gates_of(4 * m + j + 1, gxn, gyn);
                float x[4], y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    typedef unsigned u2 __attribute__((ext_vector_type(2)));
                    const u2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[m][0][4 * j + i]), __float_as_uint(acc[m][1][4 * j + i]), false, false);
                    x[i] = __uint_as_float(sw[0]); y[i] = __uint_as_float(sw[1]);     // rows 32 m + 8 j + i and + 4 + i, all 64 columns of the wave
                }
                // the eight rows of this group in list order: one test for "no segment starts here" (the common case: a receiver has ~20 C-alpha /
                // ~60 full-atom edges), else row by row
                const int base8 = 32 * m + 8 * j;
                unsigned bits8 = (sm[base8 >> 5] >> (base8 & 31)) & 0xffu;
                if (base8 == 0) bits8 &= ~1u;                                                      // row 0 opens segment 0: nothing to flush
                const float gxa[4] = {gx.x, gx.y, gx.z, gx.w}, gya[4] = {gy.x, gy.y, gy.z, gy.w};
                if (bits8 == 0u) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum = __fmaf_rn(x[i], gxa[i], sum);
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum = __fmaf_rn(y[i], gya[i], sum);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (bits8 & (1u << i)) flush();
                        sum = __fmaf_rn(x[i], gxa[i], sum);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (bits8 & (16u << i)) flush();
                        sum = __fmaf_rn(y[i], gya[i], sum);
                    }
                }
            }
        flush();                                                         // the tile's last segment when it ends with the tile's last row
    }
#undef STAMP
#undef STAMPB
}

// ------------------------------------------------------------------------------------------------------------------------------
// What both drivers share: the per-kernel constants of a thread, the walk over a workgroup's (or half workgroup's) chunks and tiles, and the
// index phase of a tile.
template <bool COORD>
struct EdgeSrc {
    const float* P; const float* Q; const int* rowp; const int* colp; const float* d0p;
    __device__ __forceinline__ EdgeSrc(const Work& w) : P(COORD ? w.Pc : w.P), Q(COORD ? w.Qc : w.Q), rowp(COORD ? w.crow : w.erow), colp(COORD ? w.ccol : w.ecol), d0p(COORD ? w.cd0 : w.ed0) {}
};

// the k-th chunk of virtual workgroup vb of nb (XCD-aware: workgroups with equal index % 8 share an L2 and get one contiguous range of chunks)
__device__ __forceinline__ int xcd_chunk_v(int vb, int nb, int k, int nch) {
    const int g = vb & 7, wg_in_g = vb >> 3, wgs_in_g = (nb - g + 7) >> 3, per_g = (nch + 7) >> 3;
    const int t = wg_in_g + k * wgs_in_g;
    if (wgs_in_g == 0 || t >= per_g) return -1;
    const int c = g * per_g + t;
    return c < nch ? c : -1;
}

// Tile walk: chunk size CH, nch chunks; a chunk is cut into equal tiles of 32 .. MT rows (trows).  All values wave-uniform.
struct TileWalk {
    int vb, nb, E, CH, nch;          // the list and its cut
    int kc, cbeg, cend, trows, e0;   // the chunk in work and the next tile's first row; cbeg < 0: nothing left
    __device__ __forceinline__ void load_chunk() {
        const int c = xcd_chunk_v(vb, nb, kc, nch);
        if (c < 0) { cbeg = -1; cend = -1; trows = 32; e0 = 0; return; }
        cbeg = c * CH; cend = min(E, cbeg + CH);
        e0 = cbeg;
        split_rest();
    }
    // rows of the next tile: what is left of the chunk cut into equal tiles of at most MT rows (a multiple of 32).  Re-split before EVERY tile,
    // so a chunk never ends in a short tile behind full ones (416 rows -> 128, 96, 96, 96; after a tile was cut at its 33rd receiver the
    // rest is divided again): a receiver with fewer edges than the shortest tile (>= 64 rows while the chunk has them) lies in at most two tiles.
    __device__ __forceinline__ void split_rest() {
        const int rest = cend - e0;
        const int ntile = (rest + MT - 1) / MT;
        trows = ntile > 0 ? ((((rest + ntile - 1) / ntile) + 31) & ~31) : 32;
    }
    __device__ __forceinline__ void init(int vb_, int nb_, int E_, int max_n) {
        vb = vb_; nb = nb_; E = E_;
        CH = (((E + nb - 1) / nb) + 31) & ~31;
        // a chunk holds more rows than any receiver has edges: the receiver's rows then lie in at most two chunks, i.e. its sum has at most two
        // float-atomic partials, which commute - results are reproducible bit for bit.  Dense samples (full-atom pockets: ~60 edges per phar point
        // at the pocket centre) need a full tile for that; C-alpha samples (< 60 nodes each) half of one.  Short lists just use fewer workgroups.
        const int minch = max_n > 128 ? MTL : 64;
        if (CH < minch) CH = minch;
        nch = (E + CH - 1) / CH;
        kc = 0; load_chunk();
    }
    __device__ __forceinline__ bool valid() const { return cbeg >= 0; }
    __device__ __forceinline__ int ne_full() const { return min(trows, cend - e0); }
    // after a tile of ne rows: is there another tile for this walker?  (does not advance)
    __device__ __forceinline__ bool more_after(int ne) const { return e0 + ne < cend || xcd_chunk_v(vb, nb, kc + 1, nch) >= 0; }
    __device__ __forceinline__ void advance(int ne) { e0 += ne; if (e0 >= cend) { ++kc; load_chunk(); } else split_rest(); }
};

// (row, col, d0, level) of a tile's rows, kept one tile ahead by wave 0: two rows per lane
struct RowPref { int nrow[2], ncol[2], nhop[2]; float nd0[2]; int nx_e0; };
template <bool COORD>
__device__ __forceinline__ void pref_fetch(RowPref& pf, const EdgeSrc<COORD>& es, const Work& w, const int lane, const int live_thr, int e0, int ne) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        pf.nrow[u] = -1; pf.ncol[u] = -1; pf.nd0[u] = 0.f; pf.nhop[u] = 255;
        if (64 * u + lane < ne) {
            const int e = e0 + 64 * u + lane;
            pf.nrow[u] = es.rowp[e]; pf.ncol[u] = es.colp[e]; pf.nd0[u] = es.d0p[e];
            if (!COORD && live_thr) pf.nhop[u] = w.ehop[e];
        }
    }
    pf.nx_e0 = e0;
}

// ---------------- index phase of the tile at e0 (wave 0 of the tile's four waves): positions, radial, segments.  Writes L.e / cd / seg /
// segrow / segstart / smask / meta; requests the rows of the tile after it.
template <bool COORD>
__device__ __forceinline__ void index_phase(E128Lds& L, const Layout& lay, const Work& w, const Dims& d, const EdgeSrc<COORD>& es, RowPref& pf, const TileWalk& tw,
                                            const int lane_in, const int layer_in, const int live_thr) {
    // opaque copies: whatever this phase derives from the lane or the layer (lane masks, per-lane addresses, 64-bit row offsets) is formed HERE, per tile,
    // instead of once per kernel - hoisted, such values stay live across the tile's GEMMs and end up in scratch memory, whose reloads wait
    // (s_waitcnt vmcnt) for every store of the previous tile
    int lane = lane_in, layer = layer_in;
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+s"(layer));
    const int e0 = tw.e0, ne_full = tw.ne_full(), cend = tw.cend, trows = tw.trows;
    if (pf.nx_e0 != e0) pref_fetch<COORD>(pf, es, w, lane, live_thr, e0, ne_full);      // first tile of the chunk, or the previous tile was cut short (below)
    int row[2], col[2], hop[2]; float d0[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { row[u] = pf.nrow[u]; col[u] = pf.ncol[u]; hop[u] = pf.nhop[u]; d0[u] = pf.nd0[u]; }
    if (e0 + ne_full < cend) {                   // the rows of the tile after this one (sized as TileWalk::split_rest will size it; a cut tile is fetched again)
        const int rest = cend - e0 - ne_full, ntile = (rest + MT - 1) / MT;
        pref_fetch<COORD>(pf, es, w, lane, live_thr, e0 + ne_full, min(rest, (((rest + ntile - 1) / ntile) + 31) & ~31));
    }
    // segments: runs of equal receivers (the lists are sorted by receiver)
    // (lane - 1's value by DPP wave_shr:1, lane 63's by v_readlane: no lane-id arithmetic kept in registers across the tile)
    const int prev0 = __builtin_amdgcn_update_dpp(row[0], row[0], 0x138, 0xf, 0xf, false);
    const bool s0 = lane < ne_full && (lane == 0 || row[0] != prev0);
    const int last0 = __builtin_amdgcn_readlane(row[0], 63);
    const int prev1 = __builtin_amdgcn_update_dpp(row[1], row[1], 0x138, 0xf, 0xf, false);
    const bool s1 = 64 + lane < ne_full && row[1] != (lane == 0 ? last0 : prev1);
    const unsigned long long m0 = __ballot(s0), m1 = __ballot(s1);
    const unsigned long long below = (2ull << lane) - 1ull;                        // lanes <= this one
    const int n0 = __popcll(m0);
    const int sg0 = __popcll(m0 & below) - 1, sg1 = n0 + __popcll(m1 & below) - 1;
    int ne = ne_full, ns = n0 + __popcll(m1);
    if (!COORD && ns > 32) {
        // the message kernel's epilogue keeps 32 receivers per tile: cut the tile where the 33rd begins (the next tile starts there)
        ne = __popcll(__ballot(lane < ne_full && sg0 < 32)) + __popcll(__ballot(64 + lane < ne_full && sg1 < 32));
        ns = 32;
    }
    bool live = true;
    if (!COORD && live_thr) live = (__ballot(lane < ne && hop[0] <= live_thr) | __ballot(64 + lane < ne && hop[1] <= live_thr)) != 0ull;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = 64 * u + lane;
        float r = 0.f;
        if (t < ne && live) {
            if (COORD) {
                const float4 pi = pos_mat(lay, w, row[u], layer), pj = pos_mat(lay, w, col[u], layer);
                float cx = pi.x - pj.x, cy = pi.y - pj.y, cz = pi.z - pj.z;
                r = cx * cx + cy * cy + cz * cz;
                const float den = sqrtf(r + 1e-8f) + d.norm_constant;              // coord2diff, egnn_new.py:265-271
                L.cd[t][0] = cx / den; L.cd[t][1] = cy / den; L.cd[t][2] = cz / den;
            } else {
                // block 0: the radial IS the d0 of the graph pass (same dist2, same operands); later blocks: lazily updated positions
                r = layer == 0 ? d0[u] : dist2(pos_lazy(lay, w, d, row[u], layer), pos_lazy(lay, w, d, col[u], layer));
            }
        }
        EdgeRec er; er.row = row[u]; er.col = col[u]; er.r = r; er.d0 = d0[u];
        L.e[t] = er;
    }
    L.seg[lane] = (unsigned char)(lane < ne ? sg0 : 255);
    L.seg[64 + lane] = (unsigned char)(64 + lane < ne ? sg1 : 255);
    if (s0 && lane < ne) { L.segrow[sg0] = row[0]; L.segstart[sg0] = lane; }
    if (s1 && 64 + lane < ne) { L.segrow[sg1] = row[1]; L.segstart[sg1] = 64 + lane; }
    if (lane == 0) {
        L.meta[0] = ns; L.meta[1] = live ? 1 : 0; L.meta[2] = ne; L.segstart[ns] = ne;
        const unsigned long long k0 = m0 | (ne < 64 ? 1ull << ne : 0ull), k1 = m1 | (ne >= 64 && ne < 128 ? 1ull << (ne - 64) : 0ull);
        L.smask[0] = (int)(unsigned)k0; L.smask[1] = (int)(unsigned)(k0 >> 32); L.smask[2] = (int)(unsigned)k1; L.smask[3] = (int)(unsigned)(k1 >> 32);
    }
}

template <bool COORD>
__device__ __forceinline__ void tile_dispatch(E128Lds& L, const Layout& lay, const Work& w, const Dims& d, const TileCtx& tc, const int ne, efrag (&bs)[2][2][NPL],
                                              unsigned long long (&st_)[8], unsigned long long& st_t) {
    switch ((ne + 31) >> 5) {
        case 4: if constexpr (MT >= 128) tile_compute<COORD, 4>(L, lay, w, d, tc, ne, bs, st_, st_t); break;
        case 3: if constexpr (MT >= 96) tile_compute<COORD, 3>(L, lay, w, d, tc, ne, bs, st_, st_t); break;
        case 2: tile_compute<COORD, 2>(L, lay, w, d, tc, ne, bs, st_, st_t); break;
        default: tile_compute<COORD, 1>(L, lay, w, d, tc, ne, bs, st_, st_t); break;
    }
}

template <bool COORD>
__device__ __forceinline__ TileCtx make_ctx(const LayerW& lw, const EdgeSrc<COORD>& es, const int wave, const int lane, const int layer, const int n_rows) {
    const WPack& W = COORD ? lw.W7 : lw.W2;
    const float* bvec = COORD ? lw.b7 : lw.b2;
    const float* hvec = COORD ? lw.w5 : lw.wa;                                   // the row dot's weight vector
    TileCtx tc;
    tc.colw = 64 * wave + (lane & 31);
    tc.rb = row_bufs(es.P, es.Q, n_rows); tc.layer = layer;
    const float sc = NPL == 3 ? 1.0f : W.wh_scale;                               // the accumulators carry the weight pack's scale
    tc.sc = sc; tc.c1 = -1.4426950408889634f * (NPL == 3 ? 1.0f : W.wh_inv);
    (void)bvec; (void)hvec;
    tc.ba0 = COORD ? 0.f : lw.ba[0];
    tc.rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<efrag*>(reinterpret_cast<const efrag*>(NPL == 3 ? W.ws : W.wh) + (size_t)(2 * wave) * NS), 0, (int)(2u * NS * 16u), 0x00020000);
    return tc;
}

// ------------------------------------------------------------------------------------------------------------------------------
// The driver: two free-running 256-thread workgroups per CU, one chunk walk each.  (Round 5's phase-locked 512-thread driver, k_edge128pp, lost its
// A/B by 4-6 % and left the build in round 6: profiles/r06_removed_experiments.patch.)
template <bool COORD>
__global__ __launch_bounds__(256, CMDGEN_E128_WPS) void k_edge128(Layout lay, Work w, Dims d, LayerW lw, int layer, int live_thr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char e128_lds[];        // (dynamic: the fused form's two plane buffers bring E128Lds to 77 KB)
    E128Lds& L = *reinterpret_cast<E128Lds*>(e128_lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const EdgeSrc<COORD> es(w);
    L.wrd[tid] = (COORD ? lw.wr_c : lw.wr_e)[tid]; L.wrd[H + tid] = (COORD ? lw.wd_c : lw.wd_e)[tid];       // visible after the first tile's barrier
    L.colv[tid] = (COORD ? lw.b7 : lw.b2)[tid] * (NPL == 3 ? 1.0f : (COORD ? lw.W7 : lw.W2).wh_scale); L.colv[H + tid] = (COORD ? lw.w5 : lw.wa)[tid];
    const TileCtx tc = make_ctx<COORD>(lw, es, wave, lane, layer, lay.N);
    TileWalk tw; tw.init((int)blockIdx.x, (int)gridDim.x, w.totals[COORD ? 1 : 0], lay.max_n);
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0;       // diagnostic builds (-DCMDGEN_STAMPS=6): summed phase cycles
#if CMDGEN_STAMPS == 6 || CMDGEN_STAMPS == 9
    st_t = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_t; int st_tiles = 0;
#endif
#if CMDGEN_STAMPS == 6
#define STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_t; st_t = n_; } while (0)
#define STAMPB(i) do {} while (0)
#elif CMDGEN_STAMPS == 9
#define STAMPB(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_t; st_t = n_; } while (0)
#define STAMP(i) do {} while (0)
#else
#define STAMP(i) do {} while (0)
#define STAMPB(i) do {} while (0)
#endif
    RowPref pf; pf.nx_e0 = -1;
#pragma unroll
    for (int u = 0; u < 2; ++u) { pf.nrow[u] = -1; pf.ncol[u] = -1; pf.nhop[u] = 255; pf.nd0[u] = 0.f; }
    while (tw.valid()) {
        efrag bs[2][2][NPL];
#pragma unroll
        for (int i = 0; i < 2 * NPL; ++i) bs[0][i & 1][i >> 1] = w_frag(tc.rw, lane, (int)(((unsigned)(i & 1) * NS + (unsigned)(i >> 1) * 64u) << 4));     // k-block 0, in flight during the index phase
        if (wave == 0) index_phase<COORD>(L, lay, w, d, es, pf, tw, lane, layer, live_thr);
        STAMPB(0);
        lds_barrier();
        STAMP(0);
        STAMPB(1);
        const int ne = L.meta[2];
        if (!L.meta[1]) {                                                                   // dead tile (see edge_msg_body, kernels_egnn.hip)
            if (tid == 0) atomicAdd(&w.counters[6], (unsigned long long)ne);
        } else {
            tile_dispatch<COORD>(L, lay, w, d, tc, ne, bs, st_, st_t);
        }
        STAMPB(6);
        lds_barrier();                                                                      // the next index phase rewrites e / seg / meta
        STAMP(4);
        STAMPB(7);
        tw.advance(ne);
#if CMDGEN_STAMPS == 6 || CMDGEN_STAMPS == 9
        ++st_tiles;
#endif
    }
#if CMDGEN_STAMPS == 6 || CMDGEN_STAMPS == 9
    if (lane == 0 && (blockIdx.x & 3) == 0 && st_tiles > 0 && COORD == (CMDGEN_STAMP_COORD != 0)) {
        for (int i = 0; i < 8; ++i) atomicAdd(&w.dbg[wave * 8 + i], st_[i]);
        atomicAdd(&w.dbg[32 + wave], __builtin_amdgcn_s_memtime() - st_begin);
        atomicAdd(&w.dbg[40], 1ull);
        if (wave == 0) atomicAdd(&w.dbg[41], (unsigned long long)st_tiles);
    }
#endif
}
#undef STAMP
#undef STAMPB
