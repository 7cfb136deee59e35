// cmdgen_egnn_common.h - what the translation units of the evaluation's tile kernels share (kernels_egnn.hip: readout + the launch
// sequence; kernels_egnn_graph.hip: radius graph + k_embed; kernels_egnn_msg.hip / _node.hip / _coord.hip: one kernel family each):
// the LDS row pad, the LDS-only barrier, the accumulator / projection helpers, positions, the edge-tile builders, the tile walk, the
// matrix-engine selectors of the edge kernels, and the per-family launch entry points.  One family per translation unit keeps an A/B
// rebuild of one kernel under a minute (round 4: 1 m 50 s for the single file).
#pragma once
#include "cmdgen_dev.h"
#include <hip/hip_ext.h>

#define LDA(H) ((H) + 4)

bool cmdgen_launch_node64(const EvalLaunch& a, int l, hipStream_t s);         // kernels_node64.hip: k_node for large batches
bool cmdgen_launch_node16w(const EvalLaunch& a, int l, hipStream_t s);        // kernels_node16w.hip: 16-row tiles on eight waves (small batches)
bool cmdgen_launch_msg128(const EvalLaunch& a, int l, hipStream_t s);         // kernels_edge128.hip: the edge kernels for long lists (128-row tiles)
bool cmdgen_launch_coord128(const EvalLaunch& a, int l, hipStream_t s);

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0), i.e.
// drains every outstanding global store / atomic of the wave (1-3 us each time); the barriers of
// the tile kernels only hand LDS tiles between phases, so stores and atomics stay in flight.
// Loads whose values are needed are still waited for by the compiler's own counted s_waitcnt.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// per-family launch entry points (each picks the instantiation for a.d.H and the launch's tile rows)
void cmdgen_launch_edge_count(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket, hipStream_t s);                 // kernels_egnn_graph.hip
void cmdgen_launch_edge_write(const EvalLaunch& a, hipStream_t s);
void cmdgen_launch_embed_tiles(const EvalLaunch& a, int mt, const float* xp, const float* xq, const float* t, const float4* coef, ChainState* chain, hipStream_t s);
void cmdgen_launch_write_embed_tiles(const EvalLaunch& a, int mt, const float* xp, const float* xq, const float* t, const float4* coef, ChainState* chain, hipStream_t s);   // H = 256
void cmdgen_launch_msg_tiles(const EvalLaunch& a, int l, hipStream_t s);                                                         // kernels_egnn_msg.hip
void cmdgen_launch_node_tiles(const EvalLaunch& a, int l, hipStream_t s);                                                        // kernels_egnn_node.hip
void cmdgen_launch_coord_tiles(const EvalLaunch& a, int l, hipStream_t s);                                                       // kernels_egnn_coord.hip

// ------------------------------------------------------------------------------------
// shared pieces of the tile kernels
// ------------------------------------------------------------------------------------
// out[row][col] = acc + bias (bias may be null) for rows < nvalid
template <int H, int MT>
__device__ __forceinline__ void store_acc_rows(const TileAcc<MT>& acc, int wave, float* __restrict__ out,
                                               int row0, int nvalid, const ColVec<MT>* bias) {
    acc_foreach_n<MT>(acc, wave, [&](int row, int col, int n, float v) {
        if (row < nvalid) out[(size_t)(row0 + row) * H + col] = v + (bias ? bias->v[n] : 0.f);
    });
}

// P|Q = BUF x Wpq^T for an MT-row tile already resident in LDS: two passes of H columns.
// `carry` holds the first fragments of the first pass; `after` is the GEMM that follows this call
// (its first fragments are fetched by the last iteration here).
template <int H, int MT, bool SP>
__device__ __forceinline__ void tile_project_pq(const float* buf, const WPack& Wpq,
                                                const ColVec<MT>& bias_p, float* __restrict__ Pout,
                                                float* __restrict__ Qout, int row0, int nvalid,
                                                bool want_p, typename Eng<MT, SP>::Carry& carry, const typename Eng<MT, SP>::Frag after,
                                                bool want_q = true) {
    typedef Eng<MT, SP> G;
    const int wave = threadIdx.x >> 6;
    const typename G::Frag fp = G::frag(Wpq, H / 8, 0, wave), fq = G::frag(Wpq, H / 8, 0, H / 64 + wave);
    TileAcc<MT> acc;
    if (want_p) {
        acc_zero<MT>(acc);
        G::template gemm<H / 8>(buf, LDA(H), fp, want_q ? fq : after, acc, carry);
        store_acc_rows<H, MT>(acc, wave, Pout, row0, nvalid, &bias_p);
    }
    if (want_q) {
        acc_zero<MT>(acc);
        G::template gemm<H / 8>(buf, LDA(H), fq, after, acc, carry);
        store_acc_rows<H, MT>(acc, wave, Qout, row0, nvalid, nullptr);
    }
}

// ------------------------------------------------------------------------------------
// positions: pocket rows never move in conditional mode (Nm = Nl; in joint mode Nm = N and every row
// moves); moving rows of block l are
// X[l] = X[l-1] + ACC[l-1] / normalization_factor, materialised by k_node(l) and formed on
// the fly (same expression, same bits) by k_edge_msg(l), which runs before it.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float4 node_pos(const Layout& lay, const Work& w, const Dims& d, int n,
                                           int layer, bool lazy) {
    if (n >= lay.Nm) return w.XP[n - lay.Nl];
    if (layer == 0) return w.X0[n];
    if (!lazy) return w.XL[(size_t)layer * lay.Nm + n];
    const float4 p = (layer == 1) ? w.X0[n] : w.XL[(size_t)(layer - 1) * lay.Nm + n];
    const float4 a = w.ACC[(size_t)(layer - 1) * lay.Nm + n];
    const float dv = agg_div(w, d, n);
    return make_float4(p.x + a.x / dv, p.y + a.y / dv, p.z + a.z / dv, 0.f);
}

// A-tile generation shared by the two edge kernels:
//   a1[e][:] = SiLU(P[row_e] + Q[col_e] + w_r * radial_e + w_d * d0_e)     (b folded into P)
// which equals SiLU(W1 [h_row | h_col | radial | d0] + b1) of egnn_new.py:33-36 / :89-93.
template <int H, int MT>
__device__ __forceinline__ void build_edge_tile(float* buf, const int* s_row, const int* s_col,
                                                const float* s_r, const float* s_d0, int ne,
                                                const float* __restrict__ P, const float* __restrict__ Q,
                                                const float4& wr4, const float4& wd4,      // this thread's four columns of w_r, w_d
                                                float* __restrict__ pre_out = nullptr, float* __restrict__ act_out = nullptr,
                                                const float* s_emb = nullptr, const float* s_we = nullptr) {
    // s_emb / s_we (sin_embedding, LDS): the tile's [MT][24] sinusoid features and the [24][H] feature columns of the first layer
    constexpr int LPR = H / 4;                  // lanes per row (float4 each) -> 4 rows per pass
    const int ltid = threadIdx.x % H;
    const int c4 = ltid % LPR, rsub = ltid / LPR;
#pragma unroll 8
    for (int pass = 0; pass < MT / 4; ++pass) {
        const int e = pass * 4 + rsub;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < ne) {
            const float4 p = reinterpret_cast<const float4*>(P + (size_t)s_row[e] * H)[c4];
            const float4 q = reinterpret_cast<const float4*>(Q + (size_t)s_col[e] * H)[c4];
            const float r = s_r[e], d0 = s_d0[e];
            float4 pre;
            if (s_emb) {
                pre = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
                for (int k = 0; k < 24; ++k) {
                    const float f = s_emb[e * 24 + k];
                    const float4 wk = *reinterpret_cast<const float4*>(s_we + k * H + 4 * c4);
                    pre.x = fmaf(f, wk.x, pre.x); pre.y = fmaf(f, wk.y, pre.y); pre.z = fmaf(f, wk.z, pre.z); pre.w = fmaf(f, wk.w, pre.w);
                }
            } else
            pre = make_float4(p.x + q.x + wr4.x * r + wd4.x * d0, p.y + q.y + wr4.y * r + wd4.y * d0,
                                           p.z + q.z + wr4.z * r + wd4.z * d0, p.w + q.w + wr4.w * r + wd4.w * d0);
            a.x = silu_f(pre.x); a.y = silu_f(pre.y); a.z = silu_f(pre.z); a.w = silu_f(pre.w);
            if (pre_out) {                                 // training: rows of the tile in the compact list's order
                reinterpret_cast<float4*>(pre_out + (size_t)e * H)[c4] = pre;
                if (act_out) reinterpret_cast<float4*>(act_out + (size_t)e * H)[c4] = a;
            }
        }
        *reinterpret_cast<float4*>(buf + e * LDA(H) + 4 * c4) = a;
    }
}

// squared distance rounded like the reference's coord2diff (torch.sum(coord_diff ** 2, 1): three products, two additions, no fma).  The sinusoid
// features multiply sqrt(r) by up to 2 pi 1024 / 15: one ulp of r moves the argument by 1e-4 rad, so the features use THIS rounding, not dist2's
__device__ __forceinline__ float sumsq_ref(const float4& a, const float4& b) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
// sin_embedding (egnn_new.py:249-260, :144-146, :196-197): per edge [sin(f_k sqrt(r + 1e-8)), k < 6 | cos(...) | the same of d0]
template <int H, int MT>
__device__ __forceinline__ void sin_features(float* s_emb, const float* s_r, const float* s_d0, int ne, const Dims& d) {
    for (int idx = threadIdx.x; idx < ne * 24; idx += H) {
        const int e = idx / 24, k = idx - e * 24, kk = k < 12 ? k : k - 12;
        const float x = sqrtf((k < 12 ? s_r[e] : s_d0[e]) + 1e-8f) * d.sin_freq[kk < 6 ? kk : kk - 6];
        s_emb[idx] = kk < 6 ? sinf(x) : cosf(x);
    }
}

// Half-K tile build for the plane variant of the split engine (H = 256): columns [128 half, 128 half + 128) of
//   SiLU(P[row_e] + Q[col_e] + w_r radial_e + w_d d0_e)
// split into three bf16 pieces ONCE, by the thread that computes them, and written as three planes [MT][136] - so the
// GEMM that follows carries no conversion work (cmdgen_split.h, tile_gemm_planes).  w_r / w_d come from LDS (s_wr, s_wd).
template <int MT>
__device__ __forceinline__ void build_edge_half(unsigned short* planes, int half, const int* s_row, const int* s_col,
                                                const float* s_r, const float* s_d0, int ne,
                                                const float* __restrict__ P, const float* __restrict__ Q,
                                                const float* s_wr, const float* s_wd,
                                                float* __restrict__ pre_out = nullptr, float* __restrict__ act_out = nullptr) {
    // pre_out / act_out (training forward): the tile's first row of the stored pre-activations / activations
    constexpr int H = 256, PLDA = SPLIT_PLANE_LDA(H / 2), PE = MT * PLDA;
    const int c4 = threadIdx.x & 31, rsub = threadIdx.x >> 5;          // 32 lanes x 16 bytes = one half row, 8 rows per pass
    const int col = half * (H / 2) + 4 * c4;
    const float4 wr4 = *reinterpret_cast<const float4*>(s_wr + col), wd4 = *reinterpret_cast<const float4*>(s_wd + col);
    float4 p[MT / 8], q[MT / 8];
#pragma unroll
    for (int pass = 0; pass < MT / 8; ++pass) {
        const int e = pass * 8 + rsub;
        p[pass] = make_float4(0.f, 0.f, 0.f, 0.f); q[pass] = p[pass];
        if (e < ne) {
            p[pass] = *reinterpret_cast<const float4*>(P + (size_t)s_row[e] * H + col);
            q[pass] = *reinterpret_cast<const float4*>(Q + (size_t)s_col[e] * H + col);
        }
    }
#pragma unroll
    for (int pass = 0; pass < MT / 8; ++pass) {
        const int e = pass * 8 + rsub;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < ne) {
            const float r = s_r[e], d0 = s_d0[e];
            const float4 pre = make_float4(p[pass].x + q[pass].x + wr4.x * r + wd4.x * d0, p[pass].y + q[pass].y + wr4.y * r + wd4.y * d0,
                                           p[pass].z + q[pass].z + wr4.z * r + wd4.z * d0, p[pass].w + q[pass].w + wr4.w * r + wd4.w * d0);
            a = make_float4(silu_f(pre.x), silu_f(pre.y), silu_f(pre.z), silu_f(pre.w));
            if (pre_out) {
                *reinterpret_cast<float4*>(pre_out + (size_t)e * H + col) = pre;
                if (act_out) *reinterpret_cast<float4*>(act_out + (size_t)e * H + col) = a;
            }
        }
        split_store4(planes, PE, e * PLDA + 4 * c4, a);
    }
}

// Full-K tile build for 32-row tiles (cmdgen_split.h, tile_gemm_planes_swz32): all 256 columns of the tile at once - 16 gathered rows
// per thread in flight together, ONE round trip per tile - into the swizzled, unpadded plane image.  Thread -> columns 4 c4 .. 4 c4 + 3
// (c4 = tid % 64) of rows pass * 4 + tid / 64; wr4 / wd4: the thread's four radial / d0 weights (fixed columns: registers, no LDS).
template <int NPC>
__device__ __forceinline__ void build_edge_full32(unsigned short* planes, const int* s_row, const int* s_col, const float* s_r, const float* s_d0,
                                                  int ne, const float* __restrict__ P, const float* __restrict__ Q, const float4& wr4, const float4& wd4,
                                                  float* __restrict__ pre_out = nullptr, float* __restrict__ act_out = nullptr) {
    // pre_out / act_out (training forward): the tile's first row of the stored pre-activations / activations (whole 1 KB rows per wave)
    constexpr int H = 256, MT = 32;
    const int c4 = threadIdx.x & 63, rsub = threadIdx.x >> 6;
    float4 p[MT / 4], q[MT / 4];
#pragma unroll
    for (int pass = 0; pass < MT / 4; ++pass) {
        const int e = pass * 4 + rsub;
        p[pass] = make_float4(0.f, 0.f, 0.f, 0.f); q[pass] = p[pass];
        if (e < ne) {
            p[pass] = reinterpret_cast<const float4*>(P + (size_t)s_row[e] * H)[c4];
            q[pass] = reinterpret_cast<const float4*>(Q + (size_t)s_col[e] * H)[c4];
        }
    }
#pragma unroll
    for (int pass = 0; pass < MT / 4; ++pass) {
        const int e = pass * 4 + rsub;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < ne) {
            const float r = s_r[e], d0 = s_d0[e];
            const float4 pre = make_float4(p[pass].x + q[pass].x + wr4.x * r + wd4.x * d0, p[pass].y + q[pass].y + wr4.y * r + wd4.y * d0,
                                           p[pass].z + q[pass].z + wr4.z * r + wd4.z * d0, p[pass].w + q[pass].w + wr4.w * r + wd4.w * d0);
            a = make_float4(silu_f(pre.x), silu_f(pre.y), silu_f(pre.z), silu_f(pre.w));
            if (pre_out) {
                reinterpret_cast<float4*>(pre_out + (size_t)e * H)[c4] = pre;
                if (act_out) reinterpret_cast<float4*>(act_out + (size_t)e * H)[c4] = a;
            }
        }
        if constexpr (NPC == 3) split_store4_swz(planes, e, c4, a); else split_store4_swz_half(planes, e, c4, a);
    }
}

// Training save hook: the LDS tile holds PRE-activations.  They and their SiLU leave for HBM as whole rows (16 bytes per
// lane - scattered 4-byte stores straight from the accumulators cost several times the bandwidth), and the tile is left
// holding SiLU(pre) as the sampler's epilogue would have written it.  pre_out / act_out (may be null: the consumer
// recomputes it) point at the tile's first row.
template <int H, int MT>
__device__ __forceinline__ void save_rows_silu(float* buf, int nvalid, float* __restrict__ pre_out, float* __restrict__ act_out) {
    constexpr int LPR = H / 4;
    const int c4 = threadIdx.x % LPR, rsub = threadIdx.x / LPR;
#pragma unroll
    for (int pass = 0; pass < MT / 4; ++pass) {
        const int r = pass * 4 + rsub;
        float4* cell = reinterpret_cast<float4*>(buf + r * LDA(H) + 4 * c4);
        const float4 p = *cell;
        const float4 a = make_float4(silu_f(p.x), silu_f(p.y), silu_f(p.z), silu_f(p.w));
        if (r < nvalid) {
            reinterpret_cast<float4*>(pre_out + (size_t)r * H)[c4] = p;
            if (act_out) reinterpret_cast<float4*>(act_out + (size_t)r * H)[c4] = a;
        }
        *cell = a;
    }
}

// per-row dot product of the LDS tile with a weight vector: H/MT threads per row
template <int H, int MT>
__device__ __forceinline__ float tile_row_dot(const float* buf, const float* wv, int& r_out, bool& lead) {
    constexpr int TPR = H / MT;                 // threads per row (4, 8 or 16 at H=256)
    constexpr int CPT = H / TPR;                // columns per thread (= MT)
    const int ltid = threadIdx.x % H;
    const int r = ltid / TPR, q = ltid % TPR;
    float s = 0.f;
    const float4* mrow = reinterpret_cast<const float4*>(buf + r * LDA(H) + q * CPT);
    const float4* w4 = reinterpret_cast<const float4*>(wv + q * CPT);
#pragma unroll
    for (int k = 0; k < CPT / 4; ++k) {
        const float4 m = mrow[k], a = w4[k];
        s += m.x * a.x + m.y * a.y + m.z * a.z + m.w * a.w;
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o);
    r_out = r; lead = (q == 0);
    return s;
}

// XCD-aware tile walk for the persistent-style edge kernels (cdna guide T1): workgroups are dealt
// round-robin over the 8 XCDs (blockIdx % 8 names the XCD group), each XCD has its own 4 MB L2.
// Giving every XCD group one contiguous range of tiles keeps the P/Q rows it gathers (edges are sorted
// by sample and receiver) inside that L2 instead of spreading every sample over all eight.
// Placement only affects speed, never results.  Returns the k-th tile of this workgroup or -1.
__device__ __forceinline__ int xcd_tile(int k, int ntiles) {
    const int vb = (int)blockIdx.x, nb = (int)gridDim.x;
    const int g = vb & 7;
    const int wg_in_g = vb >> 3;
    const int wgs_in_g = (nb - g + 7) >> 3;                 // workgroups whose blockIdx % 8 == g
    const int per_g = (ntiles + 7) >> 3;                    // tiles per XCD group (last group may be short)
    const int t = wg_in_g + k * wgs_in_g;
    if (wgs_in_g == 0 || t >= per_g) return -1;
    const int tile = g * per_g + t;
    return tile < ntiles ? tile : -1;
}

// the matrix engine of the full-K 32-row plane tiles (FK = pieces per operand): fragments, carry, GEMM, and the inverse of the power of
// two the accumulators carry (half engine: WPack::wh_scale)
template <int FK> struct EngFK;
template <> struct EngFK<3> {
    typedef SFragPtr Frag; typedef SCarry Carry;
    static __device__ __forceinline__ Frag frag(const WPack& W, int, int, int cg) { return sfrag_ptr(W.ws, 16, 0, cg); }
    static __device__ __forceinline__ void prefetch(const Frag& f, Carry& c) { split_prefetch(f, c); }
    static __device__ __forceinline__ void gemm(const unsigned short* planes, const Frag f, sf32x16 (&acc)[1][2], Carry& c) { tile_gemm_planes_swz32(planes, f, f, acc, c); }
    static __device__ __forceinline__ float inv(const WPack&) { return 1.0f; }
};
template <> struct EngFK<2> {
    typedef HFragPtr Frag; typedef HCarry Carry;
    static __device__ __forceinline__ Frag frag(const WPack& W, int, int, int cg) { return hfrag_ptr(W.wh, 16, 0, cg); }
    static __device__ __forceinline__ void prefetch(const Frag& f, Carry& c) { half_prefetch(f, c); }
    static __device__ __forceinline__ void gemm(const unsigned short* planes, const Frag f, sf32x16 (&acc)[1][2], Carry& c) { tile_gemm_planes_swz32_half(planes, f, f, acc, c); }
    static __device__ __forceinline__ float inv(const WPack& W) { return W.wh_inv; }
};
template <int MT, bool SP, int FK> struct EdgeEng { typedef Eng<MT, SP> G; };
template <int MT, bool SP> struct EdgeEng<MT, SP, 3> { typedef EngFK<3> G; };
template <int MT, bool SP> struct EdgeEng<MT, SP, 2> { typedef EngFK<2> G; };

// ------------------------------------------------------------------------------------
// LDS of an edge-tile workgroup, shared by the two edge bodies.
// ------------------------------------------------------------------------------------
// FK (full-K planes, 32-row tiles of the sampler on the split engine): the A tile is three unpadded [32][256] bf16 planes (48 KB) and the
// radial / d0 weights live in registers - 51.6 KB in all, three workgroups per CU.
template <int H, int MT, int FK = 0> struct EdgeLds {       // FK: 0, or the number of full-K planes (3: bf16 split, 2: half engine)
    float buf[FK ? (FK * MT * H / 2 > MT * LDA(H) ? FK * MT * H / 2 : MT * LDA(H)) : MT * LDA(H)];   // A tile (fp32 image or bf16 planes), then the epilogue's m tile
    int s_row[MT], s_col[MT];
    float s_r[MT], s_d0[MT], s_att[MT];
    float s_cd[MT][3], s_tr[MT][3];         // coordinate body only
    float s_vec[H];                         // att_mlp / coord_mlp.4 weight: read by every tile's row dot (LDS broadcast, not 16 L1 round trips)
    float s_wrd[FK ? 4 : 2 * H];            // radial / d0 weight columns (half-K plane variant)
    int s_live[2];                          // last block of a conditional evaluation: does the tile hold a receiver whose h is still read?
};

// tiles of >= 32 rows run on the split-bf16 engine when the launch asks for it (the training forward: only its two edge
// kernels, and only when the step re-packed split weights for them - save_split); 16-row tiles are always fp32 MFMA
// (there the L2 weight stream, not the matrix rate, binds)
#define MT_DISPATCH(mt, FN, ...) do { const bool sp_ = a.split && (!a.save || a.save_split);                                                 \
        if ((mt) >= 64) { if (sp_) FN<H, 64, true>(__VA_ARGS__); else FN<H, 64, false>(__VA_ARGS__); }                       \
        else if ((mt) == 32) { if (sp_) FN<H, 32, true>(__VA_ARGS__); else FN<H, 32, false>(__VA_ARGS__); }                 \
        else FN<H, 16, false>(__VA_ARGS__); } while (0)

