// kernels_node64.hip - k_node for LARGE batches: 64-row tiles, both operands' images in LDS, one workgroup per CU.
//
// Why.  A node tile streams every weight of the block's GEMM chain (GCL.node_model, egnn_new.py:48-58, and the first-layer
// factorisation of the two edge MLPs): 7 H^2 x 6 B = 2.75 MB on the split engine, whatever its height.  With 32-row tiles
// (kernels_egnn.hip, k_node<256,32>) 256 C-alpha pockets are 472 tiles = 1.3 GB of weight traffic per launch, full-atom pockets
// 8.4 GB - 15.6 and 17.8 TB/s in the measured launch times, close to the streaming rate the chip's L2s deliver (17-22 TB/s,
// tools/l2_stream_test.cpp).  A 64-row tile halves the bytes per row.  MEASURED (profiles/r03_m_node64.txt): no gain - the 32-row
// kernel was not waiting for its stream after all; both variants spend ~1100 cycles per k-block where the MFMAs take 768, on the
// VALU work of four waves that each split the same A fragments.  Kept opt-in (CMDGEN_NODE64=1) as the starting point of a
// producer-side-planes version.  It needs what round 2's 64-row variant did not have: BOTH fp32 images in LDS (h kept for the residual,
// agg -> T -> h_new: 2 x 66 KB, one workgroup per CU) and, with a single wave per SIMD, the register split of the A fragments
// placed BETWEEN the MFMAs by hand (hipcc puts it in front of or behind a k-block's MFMAs: matrix pipe and VALU in series).
//
// Tile: 64 rows x 256 columns, 4 waves x 64 columns (2 x 2 accumulator tiles of 32 x 32), v_mfma_f32_32x32x16_bf16, six bf16
// products per fp32 product (cmdgen_split.h), weight fragments three k-blocks ahead in a ring of four register sets carried from
// one GEMM of the chain into the next.
#include "cmdgen_dev.h"
#include <hip/hip_ext.h>

#define NLDA 260            // floats per LDS row: 256 + 4 (conflict-free ds_read_b128)
#define NROWS 64
#define NRING 4

__device__ __forceinline__ void n64_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct N64Ring { sbf16x8 b[NRING][2][3]; };     // k-blocks x two 32-column tiles x three pieces

// acc[m][n] += A(lds fp32 image, 64 rows, first k at `ap`) x W_n^T over K = 256 (16 k-blocks) for the wave's two 32-column tiles.
// cur[n] / nxt[n]: WAVE-UNIFORM pointers to k-block 0 of tile n of this GEMM / the next one; the lane's 16 bytes at [lane + 64 piece].
// On entry the ring holds k-blocks 0, 1, 2 of this GEMM in sets 0, 1, 2; on exit those of the next.  Per k-block: 24 MFMAs, and
// between them, pinned by sched_barriers, the eight pair-splits (11 VALU operations each) of the NEXT block's two A fragments.
__device__ __forceinline__ void n64_gemm(const float* ap, const sbf16x8* const (&cur)[2], const sbf16x8* const (&nxt)[2],
                                         sf32x16 (&acc)[2][2], N64Ring& ring) {
    constexpr int KB16 = 16;
    const int lane = threadIdx.x & 63;
    float4 raw[2][2][2];                             // [set][m][half of the 8 k-values]
    uint32_t pa[2][2][4], pb[2][2][4], pc[2][2][4];  // [set][m][pair]: the three bf16 pieces, as packed pairs
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
#define NG_FRAG(P, S, M) __builtin_bit_cast(sbf16x8, (u4v){P[S][M][0], P[S][M][1], P[S][M][2], P[S][M][3]})
#define NG_LOADB(SET, KB) { const bool in_ = (KB) < KB16; _Pragma("unroll") for (int n = 0; n < 2; ++n) {                       \
        const sbf16x8* q_ = in_ ? cur[n] + (unsigned)(KB) * 192u : nxt[n] + (unsigned)((KB) - KB16) * 192u;                       \
        _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) ring.b[SET][n][s_] = q_[lane + s_ * 64]; } }
    // A reads run two blocks ahead; past the k-range they fetch the row's pad / the next row (in bounds, unused)
#define NG_LOADA(SET, KB) _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                         \
        raw[SET][m][0] = *reinterpret_cast<const float4*>(ap + m * 32 * NLDA + (KB) * 16);                                         \
        raw[SET][m][1] = *reinterpret_cast<const float4*>(ap + m * 32 * NLDA + (KB) * 16 + 4); }
#define NG_SP(DST, SET, M, J) { const float x_ = (J) == 0 ? raw[SET][M][0].x : (J) == 1 ? raw[SET][M][0].z : (J) == 2 ? raw[SET][M][1].x : raw[SET][M][1].z;   \
                                const float y_ = (J) == 0 ? raw[SET][M][0].y : (J) == 1 ? raw[SET][M][0].w : (J) == 2 ? raw[SET][M][1].y : raw[SET][M][1].w;   \
                                split3_pair(x_, y_, pa[DST][M][J], pb[DST][M][J], pc[DST][M][J]); }
#define NG_SB() __builtin_amdgcn_sched_barrier(0);
#define NG_MF(M, N, AP, AS, BS, BI) acc[M][N] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(NG_FRAG(AP, AS, M), ring.b[BS][N][BI], acc[M][N], 0, 0, 0);
    // three MFMAs (one piece pairing over three of the four accumulator tiles ...), then one pair-split of the next block; small terms first
#define NG_TRIPLE(AS, BS, X0, X1, X2) X0 X1 X2 NG_SB()
#define NG_BODY(AS, BS)                                                                                                                                      \
        NG_MF(0, 0, pc, AS, BS, 0) NG_MF(0, 1, pc, AS, BS, 0) NG_MF(1, 0, pc, AS, BS, 0) NG_SB() NG_SP((AS) ^ 1, (AS) ^ 1, 0, 0) NG_SB()                    \
        NG_MF(1, 1, pc, AS, BS, 0) NG_MF(0, 0, pb, AS, BS, 1) NG_MF(0, 1, pb, AS, BS, 1) NG_SB() NG_SP((AS) ^ 1, (AS) ^ 1, 0, 1) NG_SB()                    \
        NG_MF(1, 0, pb, AS, BS, 1) NG_MF(1, 1, pb, AS, BS, 1) NG_MF(0, 0, pa, AS, BS, 2) NG_SB() NG_SP((AS) ^ 1, (AS) ^ 1, 0, 2) NG_SB()                    \
        NG_MF(0, 1, pa, AS, BS, 2) NG_MF(1, 0, pa, AS, BS, 2) NG_MF(1, 1, pa, AS, BS, 2) NG_SB() NG_SP((AS) ^ 1, (AS) ^ 1, 0, 3) NG_SB()                    \
        NG_MF(0, 0, pb, AS, BS, 0) NG_MF(0, 1, pb, AS, BS, 0) NG_MF(1, 0, pb, AS, BS, 0) NG_SB() NG_SP((AS) ^ 1, (AS) ^ 1, 1, 0) NG_SB()                    \
        NG_MF(1, 1, pb, AS, BS, 0) NG_MF(0, 0, pa, AS, BS, 1) NG_MF(0, 1, pa, AS, BS, 1) NG_SB() NG_SP((AS) ^ 1, (AS) ^ 1, 1, 1) NG_SB()                    \
        NG_MF(1, 0, pa, AS, BS, 1) NG_MF(1, 1, pa, AS, BS, 1) NG_MF(0, 0, pa, AS, BS, 0) NG_SB() NG_SP((AS) ^ 1, (AS) ^ 1, 1, 2) NG_SB()                    \
        NG_MF(0, 1, pa, AS, BS, 0) NG_MF(1, 0, pa, AS, BS, 0) NG_MF(1, 1, pa, AS, BS, 0) NG_SB() NG_SP((AS) ^ 1, (AS) ^ 1, 1, 3) NG_SB()
#define NG_BLOCK(I) {                                                                                         \
        NG_LOADB(((I) + NRING - 1) & (NRING - 1), kb + (I) + NRING - 1)                                       \
        NG_SB()                                                                                               \
        NG_BODY((I) & 1, (I) & (NRING - 1))                                                                   \
        NG_LOADA((I) & 1, kb + (I) + 2)                                                                       \
        NG_SB() }
    NG_LOADA(0, 0) NG_LOADA(1, 1)
    NG_SP(0, 0, 0, 0) NG_SP(0, 0, 0, 1) NG_SP(0, 0, 0, 2) NG_SP(0, 0, 0, 3) NG_SP(0, 0, 1, 0) NG_SP(0, 0, 1, 1) NG_SP(0, 0, 1, 2) NG_SP(0, 0, 1, 3)
#pragma unroll 1
    for (int kb = 0; kb < KB16; kb += NRING) { NG_BLOCK(0) NG_BLOCK(1) NG_BLOCK(2) NG_BLOCK(3) }
#undef NG_FRAG
#undef NG_LOADB
#undef NG_LOADA
#undef NG_SP
#undef NG_SB
#undef NG_MF
#undef NG_TRIPLE
#undef NG_BODY
#undef NG_BLOCK
}

// a 32-column tile of a packed split weight ([nt][K/16][3 pieces][64 lanes] x 16 bytes) at k-block kb0: wave-uniform pointer
__device__ __forceinline__ const sbf16x8* n64_tile(const void* ws, int kb16_total, int nt, int kb0) {
    return reinterpret_cast<const sbf16x8*>(ws) + ((size_t)nt * kb16_total + kb0) * 192;
}

__device__ __forceinline__ float4 n64_node_pos(const Layout& lay, const Work& w, const Dims& d, int n, int layer) {
    const float4 p = (layer == 1) ? w.X0[n] : w.XL[(size_t)(layer - 1) * lay.Nm + n];
    const float4 a = w.ACC[(size_t)(layer - 1) * lay.Nm + n];
    return make_float4(p.x + a.x / d.norm_factor, p.y + a.y / d.norm_factor, p.z + a.z / d.norm_factor, 0.f);
}

#define N64_FOREACH(ACC, BODY) _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < 2; ++n) _Pragma("unroll") for (int r = 0; r < 16; ++r) { \
        const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); const int col = 64 * wave + 32 * n + (lane & 31); const float v = ACC[m][n][r]; BODY }

__global__ __launch_bounds__(256, 1) void k_node64(Layout lay, Work w, Dims d, LayerW lw, LayerW lw_next, int layer, int has_next) {
    constexpr int H = 256, LPR = H / 4;
    __shared__ __attribute__((aligned(16))) float bufs[2 * NROWS * NLDA + 64];      // + the A prefetch's overshoot past the last row
    float* buf0 = bufs;                          // h (kept for the residual)
    float* buf1 = bufs + NROWS * NLDA;           // agg / nf  ->  T  ->  h_new
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#if CMDGEN_STAMPS == 5      // diagnostic build: per-phase cycle stamps into w.dbg ([wave][phase] sums, [32 + wave] lifetime, [40] waves)
    unsigned long long nst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nst_t = __builtin_amdgcn_s_memtime();
    const unsigned long long nst_begin = nst_t;
#define NSTAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); nst_[i] += n_ - nst_t; nst_t = n_; } while (0)
#else
#define NSTAMP(i) do {} while (0)
#endif
    const int row0 = (int)blockIdx.x * NROWS;
    const int nvalid = min(NROWS, lay.N - row0);
    const bool want_pc = row0 < lay.Nm;
    const int c4 = tid % LPR, rsub = tid / LPR;
    // the chain's weight tiles: this wave's columns 64 wave .. 64 wave + 63 = tiles 2 wave, 2 wave + 1 of every [H out] matrix
    const sbf16x8* const t3a[2] = {n64_tile(lw.W3.ws, 32, 2 * wave, 0), n64_tile(lw.W3.ws, 32, 2 * wave + 1, 0)};
    const sbf16x8* const t3b[2] = {n64_tile(lw.W3.ws, 32, 2 * wave, 16), n64_tile(lw.W3.ws, 32, 2 * wave + 1, 16)};
    const sbf16x8* const t4[2] = {n64_tile(lw.W4.ws, 16, 2 * wave, 0), n64_tile(lw.W4.ws, 16, 2 * wave + 1, 0)};
    // projections: jobs 0..3 = P_c, Q_c, P', Q' (bit j of `jobs` set: the job runs); Wpq rows 0..H-1 -> P (tiles 0..7), H.. -> Q (8..15)
    const unsigned jobs = (want_pc ? 1u : 0u) | 2u | (has_next ? 12u : 0u);
    auto job_tile = [&](int j, int n) { return n64_tile(j < 2 ? lw.Wpq_c.ws : lw_next.Wpq_e.ws, 16, (j & 1) * 8 + 2 * wave + n, 0); };
    const int job0 = __builtin_ctz(jobs);
    N64Ring ring;
    const int colw = 64 * wave + (lane & 31);
    const float b3c0 = lw.b3[colw], b3c1 = lw.b3[colw + 32], b4c0 = lw.b4[colw], b4c1 = lw.b4[colw + 32];
    if (layer >= 1 && tid < NROWS) {                                           // materialise the coordinates entering this block
        const int n = row0 + tid;
        if (tid < nvalid && n < lay.Nm) w.XL[(size_t)layer * lay.Nm + n] = n64_node_pos(lay, w, d, n, layer);
    }
    // both images in two batches of 32 rows: all loads of a batch in flight, then its LDS writes; agg is zeroed where it was read
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
        float4 hv[8], av[8];
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int r = bt * 32 + pass * 4 + rsub;
            hv[pass] = make_float4(0.f, 0.f, 0.f, 0.f); av[pass] = hv[pass];
            if (r < nvalid) {
                hv[pass] = reinterpret_cast<const float4*>(w.h + (size_t)(row0 + r) * H)[c4];
                av[pass] = reinterpret_cast<const float4*>(w.agg + (size_t)(row0 + r) * H)[c4];
            }
        }
        if (bt == 1) {          // the first GEMM's weight fragments: requested behind the tile's own loads (vmcnt retires in order)
#pragma unroll
            for (int kb = 0; kb < NRING - 1; ++kb)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int s_ = 0; s_ < 3; ++s_) ring.b[kb][n][s_] = t3a[n][(unsigned)kb * 192u + lane + s_ * 64];
        }
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int r = bt * 32 + pass * 4 + rsub;
            if (r < nvalid) reinterpret_cast<float4*>(w.agg + (size_t)(row0 + r) * H)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);   // agg is zero between blocks
            float4 v = av[pass];
            v.x /= d.norm_factor; v.y /= d.norm_factor; v.z /= d.norm_factor; v.w /= d.norm_factor;
            *reinterpret_cast<float4*>(buf0 + r * NLDA + 4 * c4) = hv[pass];
            *reinterpret_cast<float4*>(buf1 + r * NLDA + 4 * c4) = v;
        }
    }
    n64_lds_barrier();
    NSTAMP(0);
    const float* a0 = buf0 + (lane & 31) * NLDA + (lane >> 5) * 8;             // this lane's A row / k-slot (32x32x16: 8 k per lane)
    const float* a1 = buf1 + (lane & 31) * NLDA + (lane >> 5) * 8;
    sf32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    n64_gemm(a0, t3a, t3b, acc, ring);                                         // h part of [h | agg]
    n64_gemm(a1, t3b, t4, acc, ring);                                          // agg part
    NSTAMP(1);
    n64_lds_barrier();                                                         // every wave is done reading agg
    N64_FOREACH(acc, buf1[row * NLDA + col] = silu_f(v + (n == 0 ? b3c0 : b3c1));)
    n64_lds_barrier();
    NSTAMP(2);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    {
        const sbf16x8* const nxt[2] = {job_tile(job0, 0), job_tile(job0, 1)};
        n64_gemm(a1, t4, nxt, acc, ring);
    }
    NSTAMP(3);
    n64_lds_barrier();                                                         // every wave is done reading T
    N64_FOREACH(acc, buf1[row * NLDA + col] = row < nvalid ? buf0[row * NLDA + col] + (v + (n == 0 ? b4c0 : b4c1)) : 0.f;)   // residual (egnn_new.py:57)
    n64_lds_barrier();
#pragma unroll
    for (int pass = 0; pass < NROWS / 4; ++pass) {                             // h_new leaves as whole 1 KiB rows
        const int r = pass * 4 + rsub;
        if (r < nvalid) reinterpret_cast<float4*>(w.h + (size_t)(row0 + r) * H)[c4] = *reinterpret_cast<const float4*>(buf1 + r * NLDA + 4 * c4);
    }
    NSTAMP(4);
    // projections, K = 256, A = h_new: one rolled loop over the jobs
#pragma unroll 1
    for (unsigned rest = jobs; rest != 0u; rest &= rest - 1u) {
        const int j = __builtin_ctz(rest);
        const unsigned after = rest & (rest - 1u);
        const int jn = after ? __builtin_ctz(after) : j;                       // (last job: re-reads its own first blocks)
        const sbf16x8* const tc[2] = {job_tile(j, 0), job_tile(j, 1)};
        const sbf16x8* const tn[2] = {job_tile(jn, 0), job_tile(jn, 1)};
        float* __restrict__ out = j == 0 ? w.Pc : j == 1 ? w.Qc : j == 2 ? w.P : w.Q;
        const float* bv = j == 0 ? lw.b6 : lw_next.b1;
        const float bias0 = (j == 0 || j == 2) ? bv[colw] : 0.f, bias1 = (j == 0 || j == 2) ? bv[colw + 32] : 0.f;       // (in flight during the GEMM)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
        n64_gemm(a1, tc, tn, acc, ring);
        N64_FOREACH(acc, if (row < nvalid) out[(size_t)(row0 + row) * H + col] = v + (n == 0 ? bias0 : bias1);)
    }
    NSTAMP(5);
#if CMDGEN_STAMPS == 5
    if (lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&w.dbg[wave * 8 + i], nst_[i]);
        atomicAdd(&w.dbg[32 + wave], __builtin_amdgcn_s_memtime() - nst_begin);
        atomicAdd(&w.dbg[40], 1ull);
    }
#endif
#undef NSTAMP
}

// launcher: true when the 64-row kernel took the launch (H = 256, split engine, sampler)
bool cmdgen_launch_node64(const EvalLaunch& a, int l, hipStream_t s) {
    if (a.d.H != 256 || !a.split || a.save || !a.node64 || !a.layers[l].W3.ws) return false;
    const int nt = (a.lay.N + NROWS - 1) / NROWS;
    const int has_next = l + 1 < a.d.L;
    if (a.pe_start) hipExtLaunchKernelGGL(k_node64, dim3(nt), dim3(256), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, a.layers[l],
                                          a.layers[has_next ? l + 1 : l], l, has_next);
    else hipLaunchKernelGGL(k_node64, dim3(nt), dim3(256), 0, s, a.lay, a.w, a.d, a.layers[l], a.layers[has_next ? l + 1 : l], l, has_next);
    return true;
}
