// cmdgen_node_planes.h - one node tile of 64 or 32 rows with the A operand as producer-side planes (node_planes_tile): the body of the four
// plane node kernels of kernels_node64.hip -
//   k_node64   64 rows, four waves (two 32-column tiles each), one workgroup per CU: ring of 16 k-blocks, fp32 h tile in LDS
//   k_node32p  32 rows, four waves, two workgroups per CU
//   k_node64e  64 rows, EIGHT waves (NCT = 1: one 32-column tile each), one workgroup per CU: two plane images, stores under the next GEMM's MFMAs
//   k_node64d  64 rows, four waves, LEAN: ring of four, no fp32 h tile, buffer addressing - two workgroups per CU
// (which one runs: make_launch, cmdgen_api.hip; why: kernels_node64.hip and profiles/r06_n_node64e.txt).
// Included once per matrix engine by kernels_node64.hip (N64_NPL = 3: three bf16 pieces per operand, six MFMAs per product; 2: two fp16
// pieces, three MFMAs - the "half" engine of cmdgen_split.h; the eight-wave and the lean tile exist on the half engine only) inside a namespace
// of its own.  No include guard on purpose.

#ifndef CMDGEN_N64_EXP
#define CMDGEN_N64_EXP 0      // timing experiments only (1: agg * rcp(nf) instead of agg / nf; 2: no zero stores to agg - wrong results)
#endif
#ifndef N64E_RING
#define N64E_RING 8      // ring depth of the eight-wave tile (two waves per SIMD: 256 registers each; 16 spills)
#endif
#define NPLD 264            // 16-bit elements per plane row: 256 + 8 (row stride 528 B: conflict-free ds_read_b128)
// Depth of the weight ring (k-blocks of fragments in registers).  4: three blocks ahead.  16 (half engine, 64-row tiles): a GEMM's WHOLE weight
// stream (16 k-blocks x 2 tiles x 2 pieces = 256 registers per lane; one wave per SIMD has 512) is requested while the previous GEMM runs, i.e.
// BEFORE that GEMM's results are stored: the stores (64 KB per tile and output, bound by the chip's HBM write rate - every workgroup reaches the
// same phase together) drain under the next GEMM instead of in front of its weight loads (in-order memory queue).  profiles/r05_s
template <int NROWS, int NCT> struct N64Depth { static constexpr int v = (N64_NPL == 2 && NROWS == 64) ? (NCT == 1 ? N64E_RING : 16) : 4; };
constexpr int NPL = N64_NPL;
constexpr unsigned KBS = 64u * NPL;          // 16-byte units per k-block of a 32-column tile in the packed split weight
#if N64_NPL == 3
typedef sbf16x8 nfrag;
#define N64_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#else
typedef sf16x8 nfrag;
#define N64_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#endif

__device__ __forceinline__ void n64_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NRING, int NCT> struct N64Ring { nfrag b[NRING][NCT][NPL]; };   // k-blocks x the wave's NCT 32-column tiles x NPL pieces
template <int NCT> struct N64Tiles { const nfrag* p[NCT]; };                   // WAVE-UNIFORM pointers to k-block 0 of the wave's tiles of one GEMM

// acc[m][n] += A(planes) x W_n^T over K = 256 (16 k-blocks) for the wave's NCT 32-column tiles (2: four waves per tile; 1: eight waves, half
// engine, 64 rows).  planes: the planes of the tile; cur.p[n] / nxt.p[n]: tile n of this GEMM / the next one (the lane's 16 bytes at
// [lane + 64 piece]).  On entry the ring holds k-blocks 0 .. NRING - 2 of this GEMM; on exit those of the next.
// side: work of the caller issued in the shadow of the MFMAs, one call per k-block with the block's index as a type (eight-wave tile: the
// stores of the PREVIOUS projection's results - its epilogue then costs no phase of its own); the GEMM is fully unrolled for it.
template <int I> struct N64Idx { static constexpr int v = I; };
struct N64NoSide { template <int I> __device__ __forceinline__ void operator()(N64Idx<I>) const {} };
template <int NMT, int NRING, int NCT, class SIDE = N64NoSide>
__device__ __forceinline__ void n64_gemm(const unsigned short* planes, const N64Tiles<NCT>& cur, const N64Tiles<NCT>& nxt,
                                         sf32x16 (&acc)[NMT][NCT], N64Ring<NRING, NCT>& ring, const SIDE& side = SIDE()) {
    constexpr int KB16 = 16, NPE = NMT * 32 * NPLD;
    const int lane = threadIdx.x & 63;
    const unsigned short* ap = planes + (lane & 31) * NPLD + (lane >> 5) * 8;
    nfrag a[2][NMT][NPL];                          // [set][m][piece]
    // one 16-byte load each: the weight fragment (tile n, piece s) of k-block KB into ring set SET / the A fragment (rows 32 m.., piece s)
#define NG_LB(SET, KB, N, S) ring.b[SET][N][S] = (N == 0 ? q0_ : q1_)[lane + (S) * 64];
    // (past the k-range the A reads fetch the row's pad / the next row: in bounds, unused)
#define NG_LA(SET, KB, M, S) a[SET][M][S] = *reinterpret_cast<const nfrag*>(ap + (S) * NPE + (M) * 32 * NPLD + (KB) * 16);
#define NG_LOADA(SET, KB) _Pragma("unroll") for (int m_ = 0; m_ < NMT; ++m_) _Pragma("unroll") for (int s_ = 0; s_ < NPL; ++s_) \
        a[SET][m_][s_] = *reinterpret_cast<const nfrag*>(ap + s_ * NPE + m_ * 32 * NPLD + (KB) * 16);
#define NG_MF(M, N, AS, AI, BS, BI) acc[M][N] = N64_MFMA(a[AS][M][AI], ring.b[BS][N][BI], acc[M][N]);
    // one group = one load and the two MFMAs of one row half (pieces AI x BI), pinned: the loads issue in the shadow of the MFMAs
    // instead of in a burst between k-blocks (which left the matrix pipe idle ~100 cycles per block).  Small terms first.
#define NG_GRP(LOAD, M, AS, BS, AI, BI) LOAD NG_MF(M, 0, AS, AI, BS, BI) NG_MF(M, 1, AS, AI, BS, BI) __builtin_amdgcn_sched_barrier(0);
#define NG_GRP1(LOAD, M, AS, BS, AI, BI) LOAD NG_MF(M, 0, AS, AI, BS, BI) __builtin_amdgcn_sched_barrier(0);
#define NG_HEAD(I) constexpr int AS_ = (I) & 1, AN_ = ((I) + 1) & 1, BS_ = (I) & (NRING - 1), BN_ = ((I) + NRING - 1) & (NRING - 1);  \
        const int ka_ = kb + (I) + 1, kq_ = kb + (I) + NRING - 1;                                             \
        const bool in_ = kq_ < KB16; const unsigned ko_ = (unsigned)(in_ ? kq_ : kq_ - KB16) * KBS;           \
        const nfrag* q0_ = (in_ ? cur.p[0] : nxt.p[0]) + ko_; const nfrag* q1_ = (in_ ? cur.p[NCT - 1] : nxt.p[NCT - 1]) + ko_;
#if N64_NPL == 3
#define NG_BLOCK(I) { NG_HEAD(I)                                                                              \
        if constexpr (NMT == 2) {                                                                             \
        NG_GRP(NG_LA(AN_, ka_, 0, 2), 0, AS_, BS_, 2, 0) NG_GRP(NG_LA(AN_, ka_, 1, 2), 1, AS_, BS_, 2, 0)       \
        NG_GRP(NG_LA(AN_, ka_, 0, 1), 0, AS_, BS_, 1, 1) NG_GRP(NG_LA(AN_, ka_, 1, 1), 1, AS_, BS_, 1, 1)       \
        NG_GRP(NG_LA(AN_, ka_, 0, 0), 0, AS_, BS_, 0, 2) NG_GRP(NG_LA(AN_, ka_, 1, 0), 1, AS_, BS_, 0, 2)       \
        NG_GRP(NG_LB(BN_, kq_, 0, 0), 0, AS_, BS_, 1, 0) NG_GRP(NG_LB(BN_, kq_, 0, 1), 1, AS_, BS_, 1, 0)       \
        NG_GRP(NG_LB(BN_, kq_, 0, 2), 0, AS_, BS_, 0, 1) NG_GRP(NG_LB(BN_, kq_, 1, 0), 1, AS_, BS_, 0, 1)       \
        NG_GRP(NG_LB(BN_, kq_, 1, 1), 0, AS_, BS_, 0, 0) NG_GRP(NG_LB(BN_, kq_, 1, 2), 1, AS_, BS_, 0, 0)       \
        } else {    /* 32-row tile: twelve MFMAs, nine loads */                                               \
        NG_GRP(NG_LA(AN_, ka_, 0, 2) NG_LB(BN_, kq_, 0, 0), 0, AS_, BS_, 2, 0) NG_GRP(NG_LA(AN_, ka_, 0, 1) NG_LB(BN_, kq_, 0, 1), 0, AS_, BS_, 1, 1) \
        NG_GRP(NG_LA(AN_, ka_, 0, 0) NG_LB(BN_, kq_, 0, 2), 0, AS_, BS_, 0, 2) NG_GRP(NG_LB(BN_, kq_, 1, 0), 0, AS_, BS_, 1, 0) \
        NG_GRP(NG_LB(BN_, kq_, 1, 1), 0, AS_, BS_, 0, 1) NG_GRP(NG_LB(BN_, kq_, 1, 2), 0, AS_, BS_, 0, 0) } }
#else
#define NG_BLOCK(I) { NG_HEAD(I)                                                                              \
        if constexpr (NCT == 1) {           /* half engine, 64 rows, eight waves: six MFMAs, six loads */         \
        static_assert(NCT == 2 || NMT == 2, "one column tile per wave: 64-row tiles");                        \
        NG_GRP1(NG_LA(AN_, ka_, 0, 1), 0, AS_, BS_, 1, 0) NG_GRP1(NG_LA(AN_, ka_, 1, 1), 1, AS_, BS_, 1, 0)     \
        NG_GRP1(NG_LA(AN_, ka_, 0, 0) NG_LB(BN_, kq_, 0, 0), 0, AS_, BS_, 0, 1) NG_GRP1(NG_LA(AN_, ka_, 1, 0) NG_LB(BN_, kq_, 0, 1), 1, AS_, BS_, 0, 1) \
        NG_GRP1(NG_SIDE(I), 0, AS_, BS_, 0, 0) NG_GRP1(, 1, AS_, BS_, 0, 0)                                    \
        } else if constexpr (NMT == 2) {    /* half engine, 64 rows: twelve MFMAs (a1 b0, a0 b1, a0 b0), eight loads */ \
        NG_GRP(NG_LA(AN_, ka_, 0, 1), 0, AS_, BS_, 1, 0) NG_GRP(NG_LA(AN_, ka_, 1, 1), 1, AS_, BS_, 1, 0)       \
        NG_GRP(NG_LA(AN_, ka_, 0, 0) NG_LB(BN_, kq_, 0, 0), 0, AS_, BS_, 0, 1) NG_GRP(NG_LA(AN_, ka_, 1, 0) NG_LB(BN_, kq_, 0, 1), 1, AS_, BS_, 0, 1) \
        NG_GRP(NG_LB(BN_, kq_, 1, 0), 0, AS_, BS_, 0, 0) NG_GRP(NG_LB(BN_, kq_, 1, 1), 1, AS_, BS_, 0, 0)       \
        } else {                            /* half engine, 32 rows: six MFMAs, six loads */                  \
        NG_GRP(NG_LA(AN_, ka_, 0, 1) NG_LB(BN_, kq_, 0, 0), 0, AS_, BS_, 1, 0)                                  \
        NG_GRP(NG_LA(AN_, ka_, 0, 0) NG_LB(BN_, kq_, 0, 1) NG_LB(BN_, kq_, 1, 0), 0, AS_, BS_, 0, 1)            \
        NG_GRP(NG_LB(BN_, kq_, 1, 1), 0, AS_, BS_, 0, 0) } }
#endif
    NG_LOADA(0, 0)
#define NG_SIDE(I) side(N64Idx<(I)>{});
    if constexpr (NRING == 16 || !std::is_same<SIDE, N64NoSide>::value) {
      constexpr int kb = 0;
      NG_BLOCK(0) NG_BLOCK(1) NG_BLOCK(2) NG_BLOCK(3) NG_BLOCK(4) NG_BLOCK(5) NG_BLOCK(6) NG_BLOCK(7)
      NG_BLOCK(8) NG_BLOCK(9) NG_BLOCK(10) NG_BLOCK(11) NG_BLOCK(12) NG_BLOCK(13) NG_BLOCK(14) NG_BLOCK(15)
    } else if constexpr (NRING == 8) {
#undef NG_SIDE
#define NG_SIDE(I)
#pragma unroll 1
      for (int kb = 0; kb < KB16; kb += NRING) { NG_BLOCK(0) NG_BLOCK(1) NG_BLOCK(2) NG_BLOCK(3) NG_BLOCK(4) NG_BLOCK(5) NG_BLOCK(6) NG_BLOCK(7) }
    } else {
      static_assert(NRING == 4, "ring depths: 4, 8, 16");
#pragma unroll 1
      for (int kb = 0; kb < KB16; kb += NRING) { NG_BLOCK(0) NG_BLOCK(1) NG_BLOCK(2) NG_BLOCK(3) }
    }
#undef NG_LB
#undef NG_LA
#undef NG_LOADA
#undef NG_MF
#undef NG_GRP
#undef NG_GRP1
#undef NG_SIDE
#undef NG_BLOCK
#undef NG_HEAD
}

// two values of the SAME column and two rows (an accumulator register pair) -> the three planes: one packed conversion per piece,
// low half to row ra, high half to row rb
__device__ __forceinline__ void n64_split_store2(unsigned short* planes, int NPE, int off_a, int off_b, float va, float vb) {
#if N64_NPL == 3
    {
        uint32_t p0, p1, p2;
        split3_pair(va, vb, p0, p1, p2);
        planes[off_a] = (unsigned short)p0;            planes[off_b] = (unsigned short)(p0 >> 16);
        planes[NPE + off_a] = (unsigned short)p1;      planes[NPE + off_b] = (unsigned short)(p1 >> 16);
        planes[2 * NPE + off_a] = (unsigned short)p2;  planes[2 * NPE + off_b] = (unsigned short)(p2 >> 16);
    }
#else
    {
        uint32_t p0, p1;
        split2_pair(va, vb, p0, p1);
        planes[off_a] = (unsigned short)p0;            planes[off_b] = (unsigned short)(p0 >> 16);
        planes[NPE + off_a] = (unsigned short)p1;      planes[NPE + off_b] = (unsigned short)(p1 >> 16);
    }
#endif
}

// a 32-column tile of a packed split weight ([nt][K/16][NPL pieces][64 lanes] x 16 bytes) at k-block kb0: wave-uniform pointer
__device__ __forceinline__ const nfrag* n64_tile(const WPack& W, int kb16_total, int nt, int kb0) {
    return reinterpret_cast<const nfrag*>(NPL == 3 ? W.ws : W.wh) + ((size_t)nt * kb16_total + kb0) * KBS;
}
// four consecutive k of one row -> the planes
__device__ __forceinline__ void n64_store4(unsigned short* planes, int NPE, int off, const float4& v) {
    if constexpr (NPL == 3) split_store4(planes, NPE, off, v); else split_store4_half(planes, NPE, off, v);
}
// the power of two a weight pack's accumulators carry (1 on the bf16 split) and its inverse
__device__ __forceinline__ float n64_scale(const WPack& W) { return NPL == 3 ? 1.0f : W.wh_scale; }
__device__ __forceinline__ float n64_inv(const WPack& W) { return NPL == 3 ? 1.0f : W.wh_inv; }
// x / d with the reciprocal r = 1 / d and one correction step (three operations instead of the ten of the IEEE sequence; the quotient is the
// correctly rounded one except in rare half-ulp ties)
__device__ __forceinline__ float n64_div(float x, float d, float r) {
    const float q = x * r;
    return __fmaf_rn(__fmaf_rn(-q, d, x), r, q);
}
// SiLU(a / sc) for an accumulator carrying the scale sc (c1 = -log2(e) / sc): five operations, the bits of silu_f(a / sc)
__device__ __forceinline__ float n64_silu_scaled(float a, float c1, float sc) {
    const float u = __builtin_amdgcn_exp2f(a * c1);
    return a * __builtin_amdgcn_rcpf(__fmaf_rn(u, sc, sc));
}

__device__ __forceinline__ float4 n64_node_pos(const Layout& lay, const Work& w, const Dims& d, int n, int layer) {
    const float4 p = (layer == 1) ? w.X0[n] : w.XL[(size_t)(layer - 1) * lay.Nm + n];
    const float4 a = w.ACC[(size_t)(layer - 1) * lay.Nm + n];
    const float dv = agg_div(w, d, n);
    return make_float4(p.x + a.x / dv, p.y + a.y / dv, p.z + a.z / dv, 0.f);
}

#define N64_ZERO(ACC) _Pragma("unroll") for (int m = 0; m < NMT; ++m) _Pragma("unroll") for (int n = 0; n < NCT; ++n) _Pragma("unroll") for (int r = 0; r < 16; ++r) ACC[m][n][r] = 0.0f;
// accumulator layout: register r of tile (m, n) -> row = 32 m + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), col = 32 NCT wave + 32 n + (lane & 31)
#define N64_ROW(M, R) ((M) * 32 + ((R) & 3) + 8 * ((R) >> 2) + 4 * (lane >> 5))

// One NROWS-row tile (64, or 32: one accumulator row per wave) of rows row0 .. min(row0 + NROWS, row_end) - 1.
// FULL: the tile has all its rows - every `row < nvalid` test folds away, so the row loads of a phase are plain back-to-back loads in flight
// together and the row stores carry no exec-mask branches (with the tests, hipcc wraps each load in its own branch and even waits inside the
// sequence: the tile-in and agg hand-over phases took 12k cycles each; profiles/r05_q).  Only a layout's last tile takes the general path.
// NCT: 32-column tiles per wave - 2: four waves (256 threads); 1: eight waves (512 threads), each with half the columns and half the epilogue work.
// LEAN: the 64-row four-wave tile sized for TWO workgroups per CU - ring of four k-blocks, no fp32 h tile in LDS (68 KB, <= 256 registers).
template <int NROWS, bool FULL, int NCT, bool LEAN>
__device__ __forceinline__ void node_planes_tile_body(unsigned short* planes, const Layout& lay, const Work& w, const Dims& d, const LayerW& lw,
                                                 const LayerW& lw_next, const int layer, const int has_next_arg, const int row0, const int row_end) {
    constexpr int H = 256, LPR = H / 4, NMT = NROWS / 32, NPE = NROWS * NPLD;
    constexpr int NW = 8 / NCT, NPASS = NROWS / NW;        // waves = rows per pass of the row-wise phases (64 threads per row)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int has_next = has_next_arg & 1, live_thr = (has_next_arg >> 1) & 0x1fffffff;       // bits 1..29: only tiles with a node within that many hops of a moving node (see below)
    const bool skip_pc = ((has_next_arg >> 30) & 1) != 0;                     // not the last GCL of its block (inv_sublayers > 1): no P_c | Q_c
#if CMDGEN_STAMPS == 5      // diagnostic build: per-phase cycle stamps into w.dbg ([wave][phase] sums, [32 + wave] lifetime, [40] waves)
    unsigned long long nst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nst_t = __builtin_amdgcn_s_memtime();
    const unsigned long long nst_begin = nst_t;
#define NSTAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); nst_[i] += n_ - nst_t; nst_t = n_; } while (0)
#else
#define NSTAMP(i) do {} while (0)
#endif
    const int nvalid = FULL ? NROWS : min(NROWS, row_end - row0);
    // half engine: the fp32 h tile stays in LDS behind the planes (64 KB; the residual reads it there instead of fetching h again)
    // eight-wave tile: a SECOND plane image instead (agg / nf is split into it while the tile comes in, so the two halves of the first product
    // run back to back with no hand-over phase between them; later it takes h_new while other waves still read T from the first image)
    constexpr bool TWO = NCT == 1;
    constexpr bool HLDS = NPL == 2 && !TWO && !LEAN;
    float* const hf = reinterpret_cast<float*>(planes + NPL * NPE + 64);
    unsigned short* const planesB = TWO ? planes + NPL * NPE + 64 : planes;
    const bool want_pc = row0 < lay.Nm;
    const int c4 = tid % LPR, rsub = tid / LPR;
    if (live_thr && !want_pc && w.need_qc) {
        // the new h of a pocket node is still read only if the node sends along a coordinate edge: a tile without one only restores
        // "agg is zero between blocks" (kernels_egnn.hip, node_tile_body)
        const int r = lane & (NROWS - 1);
        if (__ballot(r < nvalid && w.need_qc[row0 + r] <= live_thr) == 0ull) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int rr = pass * NW + rsub;
                if (rr < nvalid) reinterpret_cast<float4*>(w.agg + (size_t)(row0 + rr) * H)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (tid == 0) atomicAdd(&w.counters[7], (unsigned long long)nvalid);
            return;
        }
    }
    // the chain's weight tiles: this wave's columns 32 NCT wave .. = tiles NCT wave (, NCT wave + 1) of every [H out] matrix
    N64Tiles<NCT> t3a, t3b, t4;
#pragma unroll
    for (int n = 0; n < NCT; ++n) { t3a.p[n] = n64_tile(lw.W3, 32, NCT * wave + n, 0); t3b.p[n] = n64_tile(lw.W3, 32, NCT * wave + n, 16); t4.p[n] = n64_tile(lw.W4, 16, NCT * wave + n, 0); }
    // projections: jobs 0..3 = P_c, Q_c, P', Q' (bit j of `jobs` set: the job runs); Wpq rows 0..H-1 -> P (tiles 0..7), H.. -> Q (8..15)
    // Q_c only where a row of the tile sends along a coordinate edge of this evaluation (flags of the graph pass, kernels_egnn.hip)
    const bool want_qc = want_pc || !w.need_qc || __ballot((lane & (NROWS - 1)) < nvalid && w.need_qc[row0 + (lane & (NROWS - 1))] <= 1) != 0ull;
    const unsigned jobs = (want_pc && !skip_pc ? 1u : 0u) | (want_qc && !skip_pc ? 2u : 0u) | (has_next ? 12u : 0u);
    auto job_tiles = [&](int j) { N64Tiles<NCT> t;
        _Pragma("unroll") for (int n = 0; n < NCT; ++n) t.p[n] = n64_tile(j < 2 ? lw.Wpq_c : lw_next.Wpq_e, 16, (j & 1) * 8 + NCT * wave + n, 0);
        return t; };
    const int job0 = jobs ? __builtin_ctz(jobs) : 1;             // (no job at all: the W4 product's look-ahead reads Q_c's first blocks, unused)
    constexpr int NRING = LEAN ? 4 : N64Depth<NROWS, NCT>::v;
    N64Ring<NRING, NCT> ring;
    const int colw = 32 * NCT * wave + (lane & 31);
    const float sc3 = n64_scale(lw.W3), c13 = -1.4426950408889634f * n64_inv(lw.W3), inv4 = n64_inv(lw.W4);       // the accumulators carry their weight pack's scale
    float b3c[NCT], b4c[NCT];
#pragma unroll
    for (int n = 0; n < NCT; ++n) { b3c[n] = lw.b3[colw + 32 * n] * sc3; b4c[n] = lw.b4[colw + 32 * n]; }
    if (layer >= 1 && tid < NROWS) {                                           // materialise the coordinates entering this block
        const int n = row0 + tid;
        if (tid < nvalid && n < lay.Nm) w.XL[(size_t)layer * lay.Nm + n] = n64_node_pos(lay, w, d, n, layer);
    }
    // ---- h: rows in flight, then split once per element into the planes (four consecutive k per thread and row)
    float4 av[NPASS];
    auto agg_load = [&]() {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int r = pass * NW + rsub;
            av[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nvalid) av[pass] = reinterpret_cast<const float4*>(w.agg + (size_t)(row0 + r) * H)[c4];
        }
    };
    {
        float4 hv[NPASS];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int r = pass * NW + rsub;
            hv[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nvalid) hv[pass] = reinterpret_cast<const float4*>(w.h + (size_t)(row0 + r) * H)[c4];
        }
        if constexpr (TWO) agg_load();             // two images: agg is split before the first product - in flight together with h
        // the first GEMM's weight fragments: requested behind the tile's own loads (vmcnt retires in order)
#pragma unroll
        for (int kb = 0; kb < NRING - 1; ++kb)
#pragma unroll
            for (int n = 0; n < NCT; ++n)
#pragma unroll
                for (int s_ = 0; s_ < NPL; ++s_) ring.b[kb][n][s_] = t3a.p[n][(unsigned)kb * KBS + lane + s_ * 64];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            n64_store4(planes, NPE, (pass * NW + rsub) * NPLD + 4 * c4, hv[pass]);
            if constexpr (HLDS) *reinterpret_cast<float4*>(hf + (pass * NW + rsub) * H + 4 * c4) = hv[pass];
        }
    }
    // agg: requested now, consumed after the h-part of the first product
    if constexpr (!TWO && !LEAN) agg_load();       // (LEAN: 64 registers it cannot hold across the GEMM - loaded in the hand-over phase, beside the partner workgroup's work)
    auto agg_pass = [&](auto pidx) {          // one pass of: agg / nf -> planes (second image), agg <- 0 ("agg is zero between blocks")
        constexpr int pass = decltype(pidx)::v;
        const int r = pass * NW + rsub;
#if CMDGEN_N64_EXP != 2
        if (r < nvalid) reinterpret_cast<float4*>(w.agg + (size_t)(row0 + r) * H)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        float4 v = av[pass];
        const float dv = r < nvalid ? agg_div(w, d, row0 + r) : 1.0f;
        { const float rv = __builtin_amdgcn_rcpf(dv); v.x = n64_div(v.x, dv, rv); v.y = n64_div(v.y, dv, rv); v.z = n64_div(v.z, dv, rv); v.w = n64_div(v.w, dv, rv); }
        n64_store4(planesB, NPE, r * NPLD + 4 * c4, v);
    };
    n64_lds_barrier();
    NSTAMP(0);
    sf32x16 acc[NMT][NCT];
    N64_ZERO(acc)
    // the residual's h in the accumulator layout (two images: no fp32 tile in LDS; fetched under the agg part's MFMAs, two values per k-block, by
    // buffer loads - the row's offset a scalar operand, rows past the tile's valid ones read as 0)
    float hold[NMT][NCT][16];
    if constexpr (TWO) {
        // agg is split under the h part's MFMAs (one pass per two k-blocks), one barrier before the agg part reads it
        static_assert(!TWO || NPASS == 8, "eight passes over sixteen k-blocks");
        n64_gemm<NMT, NRING, NCT>(planes, t3a, t3b, acc, ring, [&](auto idx) { if constexpr (decltype(idx)::v % 2 == 0) agg_pass(N64Idx<decltype(idx)::v / 2>{}); });
        NSTAMP(6);
        n64_lds_barrier();
        NSTAMP(7);
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.h) + (size_t)row0 * H, 0, nvalid * H * 4, 0x00020000);
        const int hoff = ((4 * (lane >> 5)) * H + colw) * 4;
        n64_gemm<NMT, NRING, NCT>(planesB, t3b, t4, acc, ring, [&](auto idx) {
            constexpr int r = decltype(idx)::v;
#pragma unroll
            for (int m = 0; m < NMT; ++m)
                hold[m][0][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, hoff, (m * 32 + (r & 3) + 8 * (r >> 2)) * H * 4, 0));
        });
    } else {
        n64_gemm<NMT, NRING, NCT>(planes, t3a, t3b, acc, ring);                                 // h part of [h | agg]
        NSTAMP(6);
        n64_lds_barrier();                                                     // every wave is done reading h
        if constexpr (LEAN) agg_load();
        agg_pass(N64Idx<0>{}); agg_pass(N64Idx<1>{}); agg_pass(N64Idx<2>{}); agg_pass(N64Idx<3>{}); agg_pass(N64Idx<4>{}); agg_pass(N64Idx<5>{}); agg_pass(N64Idx<6>{}); agg_pass(N64Idx<7>{});
        if constexpr (NPASS == 16) { agg_pass(N64Idx<8>{}); agg_pass(N64Idx<9>{}); agg_pass(N64Idx<10>{}); agg_pass(N64Idx<11>{}); agg_pass(N64Idx<12>{}); agg_pass(N64Idx<13>{}); agg_pass(N64Idx<14>{}); agg_pass(N64Idx<15>{}); }
        static_assert(NPASS == 8 || NPASS == 16, "row passes of the tile");
        n64_lds_barrier();
        NSTAMP(7);
        n64_gemm<NMT, NRING, NCT>(planesB, t3b, t4, acc, ring);                                 // agg part
    }
    NSTAMP(1);
    n64_lds_barrier();                                                         // every wave is done reading agg
    // ---- T = SiLU(pre3): from the accumulators straight into the planes (register pairs r, r + 1 = two rows of one column)
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int n = 0; n < NCT; ++n)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float bb = b3c[n];
                const int col = colw + 32 * n;
                n64_split_store2(planes, NPE, N64_ROW(m, r) * NPLD + col, N64_ROW(m, r + 1) * NPLD + col, n64_silu_scaled(acc[m][n][r] + bb, c13, sc3), n64_silu_scaled(acc[m][n][r + 1] + bb, c13, sc3));
            }
    // the residual's h, in the accumulator layout, requested now (L2) and consumed after the W4 product (two images: fetched above)
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int n = 0; n < NCT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = N64_ROW(m, r);
                if constexpr (!HLDS && !TWO && !LEAN) hold[m][n][r] = row < nvalid ? w.h[(size_t)(row0 + row) * H + colw + 32 * n] : 0.f;
            }
    n64_lds_barrier();
    NSTAMP(2);
    N64_ZERO(acc)
    {
        const N64Tiles<NCT> nxt = job_tiles(job0);
        n64_gemm<NMT, NRING, NCT>(planes, t4, nxt, acc, ring);
    }
    NSTAMP(3);
    // the tile's rows of h as a buffer: the row's offset is a compile-time scalar operand of every access, the lane's part one register - no
    // per-row address registers (the eight-wave and the lean tile have 256 registers) - and rows past the valid ones fall outside its range
    __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(w.h + (size_t)row0 * H, 0, nvalid * H * 4, 0x00020000);
    const int lean_off = ((4 * (lane >> 5)) * H + colw) * 4;
    if constexpr (LEAN) {           // the residual's h: fetched here (not held across the W4 product: registers), in flight during the barrier
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int n = 0; n < NCT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    hold[m][n][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, lean_off + 32 * n * 4, (m * 32 + (r & 3) + 8 * (r >> 2)) * H * 4, 0));
    }
    if constexpr (!TWO) n64_lds_barrier();                                     // every wave is done reading T (two images: h_new goes to the other one)
    // ---- h_new = h + (acc + b4): to global from the accumulators, and split into the planes for the projections
    // (eight waves: the rows go out under the first projection's MFMAs - see the projections - or after them when the tile has no projection)
    sf32x16 accp[NMT][1];
    bool have_prev = false; float biasp = 0.f, invp = 1.f;
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int n = 0; n < NCT; ++n)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float bb = b4c[n];
                const int col = colw + 32 * n, ra = N64_ROW(m, r), rb = N64_ROW(m, r + 1);
                const float ha = (HLDS ? hf[ra * H + col] : hold[m][n][r]) + __fmaf_rn(acc[m][n][r], inv4, bb), hb = (HLDS ? hf[rb * H + col] : hold[m][n][r + 1]) + __fmaf_rn(acc[m][n][r + 1], inv4, bb);       // residual (egnn_new.py:57); inv4: a power of two, exact
                if constexpr (NCT == 1) { accp[m][0][r] = ha; accp[m][0][r + 1] = hb; }
                else if constexpr (LEAN) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ha), rp, lean_off + 32 * n * 4, (m * 32 + (r & 3) + 8 * (r >> 2)) * H * 4, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hb), rp, lean_off + 32 * n * 4, (m * 32 + ((r + 1) & 3) + 8 * ((r + 1) >> 2)) * H * 4, 0);
                } else {
                    if (ra < nvalid) w.h[(size_t)(row0 + ra) * H + col] = ha;
                    if (rb < nvalid) w.h[(size_t)(row0 + rb) * H + col] = hb;
                }
                n64_split_store2(planesB, NPE, ra * NPLD + col, rb * NPLD + col, ra < nvalid ? ha : 0.f, rb < nvalid ? hb : 0.f);
            }
    n64_lds_barrier();
    NSTAMP(4);
    // ---- projections, K = 256, A = h_new: one rolled loop over the jobs
    if constexpr (NCT == 1) {
        // eight waves: a projection's results leave under the NEXT projection's MFMAs (two stores per k-block: n64_gemm's side work) instead of in a
        // phase of their own between two GEMMs; only the last job's are stored after the loop
        // (buffer stores: the row's offset is a compile-time scalar operand, the lane's part one register for the whole loop - no per-row
        // address registers - and rows past the tile's valid ones fall outside the descriptor's range, dropped by the hardware)
        have_prev = true; biasp = -0.0f; invp = 1.0f;            // h_new itself is the first "previous result": x * 1 + (-0) = x, bit for bit
        const int voff = lean_off;
        auto store_prev = [&](auto idx) {
            constexpr int r = decltype(idx)::v;
#pragma unroll
            for (int m = 0; m < NMT; ++m)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __fmaf_rn(accp[m][0][r], invp, biasp)), rp, voff, (m * 32 + (r & 3) + 8 * (r >> 2)) * H * 4, 0);
        };
#pragma unroll 1
        for (unsigned rest = jobs; rest != 0u; rest &= rest - 1u) {
            const int j = __builtin_ctz(rest);
            const unsigned after = rest & (rest - 1u);
            const int jn = after ? __builtin_ctz(after) : j;
            const N64Tiles<NCT> tc = job_tiles(j), tn = job_tiles(jn);
            float* out = j == 0 ? w.Pc : j == 1 ? w.Qc : j == 2 ? w.P : w.Q;
            const float* bv = j == 0 ? lw.b6 : lw_next.b1;
            const float bias0 = (j == 0 || j == 2) ? bv[colw] : 0.f;
            const float invj = n64_inv(j < 2 ? lw.Wpq_c : lw_next.Wpq_e);
            N64_ZERO(acc)
            if (have_prev) n64_gemm<NMT, NRING, NCT>(planesB, tc, tn, acc, ring, store_prev);
            else n64_gemm<NMT, NRING, NCT>(planesB, tc, tn, acc, ring);
#pragma unroll
            for (int m = 0; m < NMT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) accp[m][0][r] = acc[m][0][r];
            rp = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)row0 * H, 0, nvalid * H * 4, 0x00020000);
            have_prev = true; biasp = bias0; invp = invj;
        }
        if (have_prev) {
#pragma unroll
            for (int m = 0; m < NMT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __fmaf_rn(accp[m][0][r], invp, biasp)), rp, voff, (m * 32 + (r & 3) + 8 * (r >> 2)) * H * 4, 0);
        }
    } else {
#pragma unroll 1
    for (unsigned rest = jobs; rest != 0u; rest &= rest - 1u) {
        const int j = __builtin_ctz(rest);
        const unsigned after = rest & (rest - 1u);
        const int jn = after ? __builtin_ctz(after) : j;                       // (last job: re-reads its own first blocks)
        const N64Tiles<NCT> tc = job_tiles(j), tn = job_tiles(jn);
        float* __restrict__ out = j == 0 ? w.Pc : j == 1 ? w.Qc : j == 2 ? w.P : w.Q;
        const float* bv = j == 0 ? lw.b6 : lw_next.b1;
        float biasv[NCT];                                                       // (in flight during the GEMM)
#pragma unroll
        for (int n = 0; n < NCT; ++n) biasv[n] = (j == 0 || j == 2) ? bv[colw + 32 * n] : 0.f;
        const float invj = n64_inv(j < 2 ? lw.Wpq_c : lw_next.Wpq_e);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)row0 * H, 0, nvalid * H * 4, 0x00020000);
        N64_ZERO(acc)
        n64_gemm<NMT, NRING, NCT>(planes, tc, tn, acc, ring);
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int n = 0; n < NCT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = N64_ROW(m, r);
                    if constexpr (LEAN) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __fmaf_rn(acc[m][n][r], invj, biasv[n])), ro, lean_off + 32 * n * 4, (m * 32 + (r & 3) + 8 * (r >> 2)) * H * 4, 0);
                    else if (row < nvalid) out[(size_t)(row0 + row) * H + colw + 32 * n] = __fmaf_rn(acc[m][n][r], invj, biasv[n]);
                }
    }
    }
    NSTAMP(5);
#if CMDGEN_STAMPS == 5
    if (lane == 0 && wave < 4) {                   // (the eight-wave tile reports its first four waves)
        for (int i = 0; i < 8; ++i) atomicAdd(&w.dbg[wave * 8 + i], nst_[i]);
        atomicAdd(&w.dbg[32 + wave], __builtin_amdgcn_s_memtime() - nst_begin);
        atomicAdd(&w.dbg[40], 1ull);
    }
#endif
#undef NSTAMP
}

template <int NROWS, int NCT = 2, bool LEAN = false>
__device__ __forceinline__ void node_planes_tile(unsigned short* planes, const Layout& lay, const Work& w, const Dims& d, const LayerW& lw,
                                                 const LayerW& lw_next, const int layer, const int has_next_arg, const int row0, const int row_end) {
    if (row_end - row0 >= NROWS) node_planes_tile_body<NROWS, true, NCT, LEAN>(planes, lay, w, d, lw, lw_next, layer, has_next_arg, row0, row_end);
    else node_planes_tile_body<NROWS, false, NCT, LEAN>(planes, lay, w, d, lw, lw_next, layer, has_next_arg, row0, row_end);
}


