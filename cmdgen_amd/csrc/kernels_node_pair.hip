// kernels_node_pair.hip - k_node for SMALL batches: two workgroups share a 32-row node tile by output columns.
//
// Why.  With 64 pockets GCL.node_model + the projections (egnn_new.py:48-58 and the first-layer factorisation of the two
// edge MLPs) run as 236 workgroups of 16 rows, one per CU, and each of them streams EVERY weight of the block's GEMM chain
// for its 16 rows: 7 H^2 x 6 B = 2.75 MB on the split engine.  A CU pulls ~95 GB/s from its XCD's L2, so the launch is that
// stream (31 us; the matrix work behind it is 8 us; profiles/r03_b).  Here a PAIR of workgroups owns 32 rows and every
// GEMM of the chain is cut so that a workgroup touches half of its weight: 1.375 MB per CU for the same rows per CU.
//
//   GEMM1  pre3 = [h | agg/nf] W3^T + b3        cut by OUTPUT columns: half q computes columns [128 q, 128 q + 128), K = 512
//   T = SiLU(pre3)                              each half holds T[:, its 128 columns] - exactly a K-half of the next product
//   GEMM2  part_q = T[:, K-half q] W4[:, K-half q]^T     cut by K: all 256 output columns, K = 128: no exchange in front of it
//   exchange: the halves swap their partial sums (32 x 256 fp32 = 32 KB each way) through L2
//   h_new = h + ((part_0 + part_1) + b4)        formed identically (same bits) by both halves; each stores half the columns
//   P_c | Q_c | P' | Q' = h_new W^T (+ bias)    cut by output columns again, K = 256
//
// The exchange is the only inter-workgroup step: partial sums leave as L2-write-through (sc1) stores, one 128-byte line per
// store instruction, every storing wave drains (s_waitcnt vmcnt(0)), the workgroup's barrier, one lane's agent-scope atomic
// add on the pair's flag; the partner polls that flag with sc1 loads (one lane, s_sleep, bounded), the workgroup's barrier,
// then sc1 loads of the 32 KB (MI355X guide, inter-workgroup visibility, "valid forms", third row).  The grid is at most one
// workgroup per CU (all co-resident), the two halves of a pair sit 8 block ids apart (the same XCD under round-robin
// placement: speed only); a wait that gives up is counted in counters[5] and reported by cmdgen_chain_status.
// Everything runs on the split-bf16 engine (cmdgen_split.h): 32x32x16 MFMA, register split of the fp32 LDS image, weight
// fragments three k-blocks ahead in a ring of four register sets that is carried from one GEMM of the chain into the next.
#include "cmdgen_dev.h"
#include <hip/hip_ext.h>

#define PLDA 260            // floats per LDS row: 256 + 4 (conflict-free ds_read_b128)
#define PROWS 32

__device__ __forceinline__ void pair_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define PRING 16                                // ring of weight "tile-blocks" (one 32-column tile x one 16-k block x three pieces = 3 KB per wave)
struct PairRing { sbf16x8 b[PRING][3]; };

// one GEMM of the chain: acc[n] += A(lds fp32, 32 rows, row stride PLDA, first k at `ap`) x W_n^T for NT 32-column weight tiles over
// 16 / NT k-blocks of 16 - every GEMM of the chain is exactly PRING tile-blocks, numbered t = kb * NT + n and kept in ring slot t.
// cur[n] / nxt[n]: WAVE-UNIFORM pointers to k-block 0 of tile n of this GEMM / of the next one (NTN tiles); the lane's 16 bytes
// are at [lane + 64 piece].  On entry the ring holds this GEMM's tile-blocks 0 .. 14; while tile-block t is multiplied, slot
// t - 1 (free since the previous step) is refilled: with this GEMM's tile-block 15 at t = 0, with the NEXT GEMM's tile-block
// t - 1 afterwards.  So fragments run 15 tile-blocks = 45 KB per wave ahead: a workgroup alone on its CU needs that much in flight
// to draw what the L2 delivers (three k-blocks ahead: 41 GB/s per CU; seven: 70; profiles/r03_k).
// Code shape: straight-line, everything about the ring known at compile time (a branch between MFMAs puts every sched_barrier
// out of effect); three instantiations, a few call sites.
template <int NT, int NTN>
__device__ __forceinline__ void pair_gemm(const float* ap, const sbf16x8* const (&cur)[2], const sbf16x8* const (&nxt)[2],
                                          sf32x16 (&acc)[NT], PairRing& ring) {
    constexpr int KB16 = PRING / NT;
    const int lane = threadIdx.x & 63;
    float4 raw[2][2];
    uint32_t pa[2][4], pb[2][4], pc[2][4];           // the three bf16 pieces of this and the next k-block's A fragment, as packed pairs
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
#define PG_FRAG(P, S) __builtin_bit_cast(sbf16x8, (u4v){P[S][0], P[S][1], P[S][2], P[S][3]})
    // refill behind tile-block T (T = 0: this GEMM's last tile-block into slot 15; else the next GEMM's tile-block T - 1 into slot T - 1)
#define PG_REFILL(T) { constexpr int t_ = (T);                                                                              \
        const sbf16x8* q_ = t_ == 0 ? cur[(PRING - 1) % NT] + (unsigned)((PRING - 1) / NT) * 192u                          \
                                    : nxt[(t_ - 1) % NTN] + (unsigned)((t_ - 1) / NTN) * 192u;                             \
        _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) ring.b[(t_ + PRING - 1) % PRING][s_] = q_[lane + s_ * 64]; }
    // A reads run two blocks ahead; past the GEMM's k-range they fetch the row's pad / the next row (in bounds, unused)
#define PG_LOADA(SET, KB) { raw[SET][0] = *reinterpret_cast<const float4*>(ap + (KB) * 16); raw[SET][1] = *reinterpret_cast<const float4*>(ap + (KB) * 16 + 4); }
    // one pair of the next block's fragment (11 VALU operations), pinned between two MFMAs: hipcc otherwise puts the whole split in
    // front of the block's MFMAs (VALU then matrix pipe in series)
#define PG_SP(DST, SET, J) { const float x_ = (J) == 0 ? raw[SET][0].x : (J) == 1 ? raw[SET][0].z : (J) == 2 ? raw[SET][1].x : raw[SET][1].z;      \
                             const float y_ = (J) == 0 ? raw[SET][0].y : (J) == 1 ? raw[SET][0].w : (J) == 2 ? raw[SET][1].y : raw[SET][1].w;      \
                             split3_pair(x_, y_, pa[DST][J], pb[DST][J], pc[DST][J]); }
#define PG_SB() __builtin_amdgcn_sched_barrier(0);
#define PG_MF(N, AP, AS, KB, BI) acc[N] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PG_FRAG(AP, AS), ring.b[(KB) * NT + (N)][BI], acc[N], 0, 0, 0);
    // small terms first; the next block's four pairs are split in the gaps
#define PG_BODY(AS, KB)                                                                                       \
    if constexpr (NT == 1) {                                                                                  \
        PG_MF(0, pc, AS, KB, 0) PG_SB() PG_SP((AS) ^ 1, (AS) ^ 1, 0) PG_SB()                                  \
        PG_MF(0, pb, AS, KB, 1) PG_SB() PG_SP((AS) ^ 1, (AS) ^ 1, 1) PG_SB()                                  \
        PG_MF(0, pa, AS, KB, 2) PG_SB() PG_SP((AS) ^ 1, (AS) ^ 1, 2) PG_SB()                                  \
        PG_MF(0, pb, AS, KB, 0) PG_SB() PG_SP((AS) ^ 1, (AS) ^ 1, 3) PG_SB()                                  \
        PG_MF(0, pa, AS, KB, 1) PG_SB()                                                                       \
        PG_MF(0, pa, AS, KB, 0) PG_SB()                                                                       \
    } else {                                                                                                  \
        PG_MF(0, pc, AS, KB, 0) PG_MF(1, pc, AS, KB, 0) PG_SB() PG_SP((AS) ^ 1, (AS) ^ 1, 0) PG_SB()          \
        PG_MF(0, pb, AS, KB, 1) PG_MF(1, pb, AS, KB, 1) PG_SB() PG_SP((AS) ^ 1, (AS) ^ 1, 1) PG_SB()          \
        PG_MF(0, pa, AS, KB, 2) PG_MF(1, pa, AS, KB, 2) PG_SB() PG_SP((AS) ^ 1, (AS) ^ 1, 2) PG_SB()          \
        PG_MF(0, pb, AS, KB, 0) PG_MF(1, pb, AS, KB, 0) PG_SB() PG_SP((AS) ^ 1, (AS) ^ 1, 3) PG_SB()          \
        PG_MF(0, pa, AS, KB, 1) PG_MF(1, pa, AS, KB, 1) PG_SB()                                               \
        PG_MF(0, pa, AS, KB, 0) PG_MF(1, pa, AS, KB, 0) PG_SB()                                               \
    }
    // k-block KB: its MFMAs first, then the refills behind its tile-blocks (slot t - 1: the previous k-block's last tile, and
    // with two tiles this k-block's first - both consumed by now)
#define PG_BLOCK(KB) if constexpr ((KB) < KB16) {                                                             \
        PG_BODY((KB) & 1, KB)                                                                                 \
        if constexpr (NT == 1) { PG_REFILL(KB) } else { PG_REFILL(2 * (KB)) PG_REFILL(2 * (KB) + 1) }         \
        PG_LOADA((KB) & 1, (KB) + 2)                                                                          \
        PG_SB() }
    PG_LOADA(0, 0) PG_LOADA(1, 1)
    PG_SP(0, 0, 0) PG_SP(0, 0, 1) PG_SP(0, 0, 2) PG_SP(0, 0, 3)
    PG_BLOCK(0) PG_BLOCK(1) PG_BLOCK(2) PG_BLOCK(3) PG_BLOCK(4) PG_BLOCK(5) PG_BLOCK(6) PG_BLOCK(7)
    PG_BLOCK(8) PG_BLOCK(9) PG_BLOCK(10) PG_BLOCK(11) PG_BLOCK(12) PG_BLOCK(13) PG_BLOCK(14) PG_BLOCK(15)
#undef PG_FRAG
#undef PG_REFILL
#undef PG_LOADA
#undef PG_SP
#undef PG_SB
#undef PG_MF
#undef PG_BODY
#undef PG_BLOCK
}

// a 32-column tile of a packed split weight (cmdgen_split.h: [nt][K/16][3 pieces][64 lanes] x 16 bytes) at k-block kb0: a
// wave-uniform pointer (lane 0's 16 bytes)
__device__ __forceinline__ const sbf16x8* pair_tile(const void* ws, int kb16_total, int nt, int kb0) {
    return reinterpret_cast<const sbf16x8*>(ws) + ((size_t)nt * kb16_total + kb0) * 192;
}

__device__ __forceinline__ float4 pair_node_pos(const Layout& lay, const Work& w, const Dims& d, int n, int layer) {
    const float4 p = (layer == 1) ? w.X0[n] : w.XL[(size_t)(layer - 1) * lay.Nm + n];
    const float4 a = w.ACC[(size_t)(layer - 1) * lay.Nm + n];
    return make_float4(p.x + a.x / d.norm_factor, p.y + a.y / d.norm_factor, p.z + a.z / d.norm_factor, 0.f);
}

__global__ __launch_bounds__(256, 1) void k_node_pair(Layout lay, Work w, Dims d, LayerW lw, LayerW lw_next, int layer, int has_next,
                                                       int npairs, float* __restrict__ scratch, int* __restrict__ flags) {
    constexpr int H = 256, LPR = H / 4;
    __shared__ __attribute__((aligned(16))) float bufs[2 * PROWS * PLDA + 64];      // + the A prefetch's overshoot past the last row
    float* buf0 = bufs;                          // h (kept for the residual)
    float* buf1 = bufs + PROWS * PLDA;           // agg / nf  ->  T (this half's columns)  ->  h_new
    const int bid = (int)blockIdx.x;
    const int pair = (bid >> 4) * 8 + (bid & 7), half = (bid >> 3) & 1;       // halves of a pair: block ids 8 apart
    if (pair >= npairs) return;                                               // (both halves of a padding pair leave)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;      // (wave: a scalar, so that the weight pointers are)
#if CMDGEN_STAMPS == 4      // diagnostic build: per-phase cycle stamps into w.dbg ([wave][phase] sums, [32 + wave] lifetime, [40] waves)
    unsigned long long pst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pst_t = __builtin_amdgcn_s_memtime();
    const unsigned long long pst_begin = pst_t;
#define PSTAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pst_[i] += n_ - pst_t; pst_t = n_; } while (0)
#else
#define PSTAMP(i) do {} while (0)
#endif
    const int row0 = pair * PROWS;
    const int nvalid = min(PROWS, lay.N - row0);
    const bool want_pc = row0 < lay.Nm;
    const int c4 = tid % LPR, rsub = tid / LPR;
    // ---- the chain's weight tiles.  W3: [H out][2H in] (32 k-blocks); W4, Wpq_*: K = 256 (16 k-blocks); Wpq rows 0..H-1 -> P, H.. -> Q
    const int ntq = 4 * half + wave;                                           // this wave's 32-column tile of an N-split product
    const sbf16x8* const t3a[2] = {pair_tile(lw.W3.ws, 32, ntq, 0), nullptr};
    const sbf16x8* const t3b[2] = {pair_tile(lw.W3.ws, 32, ntq, 16), nullptr};
    const sbf16x8* const t4[2] = {pair_tile(lw.W4.ws, 16, 2 * wave, 8 * half), pair_tile(lw.W4.ws, 16, 2 * wave + 1, 8 * half)};
    // the projections are jobs 0..3 = P_c, Q_c, P', Q' (bit j of `jobs` set: the job runs); job j's tile / output / bias by arithmetic on j
    const unsigned jobs = (want_pc ? 1u : 0u) | 2u | (has_next ? 12u : 0u);
    auto job_tile = [&](int j) { return pair_tile(j < 2 ? lw.Wpq_c.ws : lw_next.Wpq_e.ws, 16, (j & 1) * 8 + ntq, 0); };
    const int job0 = __builtin_ctz(jobs);
    PairRing ring;
    // per-column vectors of this lane's columns, fetched now
    const int colq = 128 * half + 32 * wave + (lane & 31);                     // N-split products
    const float b3c = lw.b3[colq];
    const int col2 = 64 * wave + (lane & 31);                                  // K-split product (GEMM2): columns col2, col2 + 32
    const float b4c0 = lw.b4[col2], b4c1 = lw.b4[col2 + 32];
    if (half == 0 && layer >= 1 && tid < PROWS) {                              // materialise the coordinates entering this block
        const int n = row0 + tid;
        if (tid < nvalid && n < lay.Nm) w.XL[(size_t)layer * lay.Nm + n] = pair_node_pos(lay, w, d, n, layer);
    }
    {   // both images, all loads in flight, then the LDS writes
        float4 hv[PROWS / 4], av[PROWS / 4];
#pragma unroll
        for (int pass = 0; pass < PROWS / 4; ++pass) {
            const int r = pass * 4 + rsub;
            hv[pass] = make_float4(0.f, 0.f, 0.f, 0.f); av[pass] = hv[pass];
            if (r < nvalid) {
                hv[pass] = reinterpret_cast<const float4*>(w.h + (size_t)(row0 + r) * H)[c4];
                av[pass] = reinterpret_cast<const float4*>(w.agg + (size_t)(row0 + r) * H)[c4];
            }
        }
        // the first GEMM's weight fragments: requested behind the tile's own loads (vmcnt retires in order: the tile is waited
        // for first and must not queue behind 21 KB of weights per wave), in flight during the LDS writes and the barrier
#pragma unroll
        for (int kb = 0; kb < PRING - 1; ++kb)
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) ring.b[kb][s_] = t3a[0][(unsigned)kb * 192u + lane + s_ * 64];
#pragma unroll
        for (int pass = 0; pass < PROWS / 4; ++pass) {
            const int r = pass * 4 + rsub;
            float4 v = av[pass];
            v.x /= d.norm_factor; v.y /= d.norm_factor; v.z /= d.norm_factor; v.w /= d.norm_factor;
            *reinterpret_cast<float4*>(buf0 + r * PLDA + 4 * c4) = hv[pass];
            *reinterpret_cast<float4*>(buf1 + r * PLDA + 4 * c4) = v;
        }
    }
    pair_lds_barrier();
    PSTAMP(0);
    const float* a0 = buf0 + (lane & 31) * PLDA + (lane >> 5) * 8;             // this lane's A row / k-slot (32x32x16: 8 k per lane)
    const float* a1 = buf1 + (lane & 31) * PLDA + (lane >> 5) * 8;
    // ---- GEMM1 (N-split): pre3[:, colq]
    sf32x16 acc1[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[0][r] = 0.0f;
    pair_gemm<1, 1>(a0, t3a, t3b, acc1, ring);
    pair_gemm<1, 2>(a1, t3b, t4, acc1, ring);
    PSTAMP(1);
    pair_lds_barrier();                                                        // every wave is done reading agg
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        buf1[row * PLDA + colq] = silu_f(acc1[0][r] + b3c);                    // T, this half's 128 columns
    }
    pair_lds_barrier();
    PSTAMP(2);
    // ---- GEMM2 (K-split): partial sums over this half's 128 k-values, all 256 columns
    sf32x16 acc2[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[n][r] = 0.0f;
    {
        const sbf16x8* const nxt[2] = {job_tile(job0), nullptr};
        pair_gemm<2, 1>(a1 + 128 * half, t4, nxt, acc2, ring);
    }
    PSTAMP(3);
    // ---- exchange of the partial sums
    float* mine = scratch + ((size_t)pair * 2 + half) * (PROWS * H);
    const float* theirs = scratch + ((size_t)pair * 2 + (half ^ 1)) * (PROWS * H);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            __hip_atomic_store(mine + row * H + col2 + 32 * n, acc2[n][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // global_store sc1
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PSTAMP(4);
    if (tid == 0) {
        __hip_atomic_fetch_add(flags + pair * 2 + half, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int it = 0;
        while (__hip_atomic_load(flags + pair * 2 + (half ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 1) {
            __builtin_amdgcn_s_sleep(1);
            if (++it > (1 << 22)) { atomicAdd(&w.counters[5], 1ull); break; }
        }
        __hip_atomic_store(flags + pair * 2 + (half ^ 1), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch (this workgroup is its only reader)
    }
    __syncthreads();
    PSTAMP(5);
    float other[2][16];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            other[n][r] = __hip_atomic_load(theirs + row * H + col2 + 32 * n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load sc1
        }
    // ---- h_new = h + ((part_0 + part_1) + b4): the same expression, the same operand order in both halves
    const bool i_store = (wave >> 1) == half;            // this half stores columns [128 half, 128 half + 128)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = col2 + 32 * n;
            const float p0 = half == 0 ? acc2[n][r] : other[n][r], p1 = half == 0 ? other[n][r] : acc2[n][r];
            const float hn = buf0[row * PLDA + col] + ((p0 + p1) + (n == 0 ? b4c0 : b4c1));                 // residual (egnn_new.py:57)
            buf1[row * PLDA + col] = row < nvalid ? hn : 0.f;
            if (i_store && row < nvalid) {
                w.h[(size_t)(row0 + row) * H + col] = hn;
                w.agg[(size_t)(row0 + row) * H + col] = 0.f;                   // agg is zero between blocks (the partner has read it: its flag says so)
            }
        }
    pair_lds_barrier();
    PSTAMP(6);
    // ---- projections (N-split), K = 256, A = h_new: one rolled loop over the jobs
#pragma unroll 1
    for (unsigned rest = jobs; rest != 0u; rest &= rest - 1u) {
        const int j = __builtin_ctz(rest);
        const unsigned after = rest & (rest - 1u);
        const sbf16x8* const tc[2] = {job_tile(j), nullptr};
        const sbf16x8* const tn[2] = {job_tile(after ? __builtin_ctz(after) : j), nullptr};       // (last job: re-reads its own first blocks)
        float* __restrict__ out = j == 0 ? w.Pc : j == 1 ? w.Qc : j == 2 ? w.P : w.Q;
        const float bias = j == 0 ? lw.b6[colq] : j == 2 ? lw_next.b1[colq] : 0.f;                 // (in flight during the GEMM)
        sf32x16 accp[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) accp[0][r] = 0.0f;
        pair_gemm<1, 1>(a1, tc, tn, accp, ring);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < nvalid) out[(size_t)(row0 + row) * H + colq] = accp[0][r] + bias;
        }
    }
    PSTAMP(7);
#if CMDGEN_STAMPS == 4
    if (lane == 0) {
        for (int i = 0; i < 8; ++i) atomicAdd(&w.dbg[wave * 8 + i], pst_[i]);
        atomicAdd(&w.dbg[32 + wave], __builtin_amdgcn_s_memtime() - pst_begin);
        atomicAdd(&w.dbg[40], 1ull);
    }
#endif
#undef PSTAMP
}

// launcher: true when the pair kernel took the launch (H = 256, split engine, sampler, at most one workgroup per CU)
bool cmdgen_launch_node_pair(const EvalLaunch& a, int l, hipStream_t s) {
    if (a.d.H != 256 || !a.split || a.save || !a.node_pair || !a.w.pair_scratch) return false;
    const int npairs = (a.lay.N + PROWS - 1) / PROWS;
    const int grid = 16 * ((npairs + 7) / 8);
    if (2 * npairs > a.n_cus || npairs > a.pair_cap) return false;
    if (!a.layers[l].W3.ws) return false;
    const int has_next = l + 1 < a.d.L;
    if (a.pe_start) hipExtLaunchKernelGGL(k_node_pair, dim3(grid), dim3(256), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, a.layers[l],
                                          a.layers[has_next ? l + 1 : l], l, has_next, npairs, a.w.pair_scratch, a.w.pair_flags);
    else hipLaunchKernelGGL(k_node_pair, dim3(grid), dim3(256), 0, s, a.lay, a.w, a.d, a.layers[l], a.layers[has_next ? l + 1 : l], l, has_next,
                            npairs, a.w.pair_scratch, a.w.pair_flags);
    return true;
}
