// kernels_egnn.hip - one EGNNDynamics.forward (dynamics.py:75-139) as gfx950 kernels.
//
// Launch sequence of one evaluation (host side: cmdgen_api.hip, launch_evaluation):
//   k_edge_count   radius graph, pass 1: degrees per receiver      (dynamics.py:141-147)
//   k_edge_write   pass 2: compact (row, col, d0) lists sorted like torch.where
//   k_embed        encoders + time column + embedding + P/Q of block 0
//   per block l:   k_edge_msg   GCL.edge_model + segment sum          (egnn_new.py:31-52)
//                  k_node       GCL.node_model, then P/Q for the coord MLP and block l+1
//                  k_edge_coord EquivariantUpdate.coord_model          (egnn_new.py:87-104)
//   k_readout      embedding_out, decoders, velocity, NaN flag       (dynamics.py:110-139)
//
// Tiling: a workgroup owns MT rows (edges or nodes; MT = 64, 32 or 16, chosen per launch so
// that even a 64-pocket batch spreads over all 256 CUs) and all H output columns; wave w owns
// columns [64w, 64w+64), so the block has H/64 waves.  The A operand (rows x H) lives in LDS
// with a 4-float row pad (conflict-free ds_read_b128); the B operand (weights) streams from
// L2 in MFMA fragment order (cmdgen_dev.h).
#include "cmdgen_dev.h"
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

// ------------------------------------------------------------------------------------
// Radius graph.  One workgroup per sample; a wave scans the candidate senders of one
// receiver at a time, so neighbours come out in ascending sender order and ballots give
// both the degree and the compaction offsets.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int flat_node(int i, int nl, int pb, int qb, int Nl) {
    return i < nl ? pb + i : Nl + qb + (i - nl);
}

__global__ void k_edge_count(Layout lay, Work w, Dims d, const float* __restrict__ xh_phar,
                             const float* __restrict__ xh_pocket) {
    extern __shared__ float4 spos[];            // [max_n] positions, then int sdeg[max_n]
    int* sdeg = reinterpret_cast<int*>(spos + lay.max_n);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nl = lay.num_phar[b], np = lay.num_pocket[b], n = nl + np;
    const int pb = lay.phar_base[b], qb = lay.pocket_base[b];
    const int ldp = 3 + d.P, ldq = 3 + d.R;
    for (int i = tid; i < n; i += blockDim.x) {
        float4 p;
        if (i < nl) {
            const float* s = xh_phar + (size_t)(pb + i) * ldp;
            p = make_float4(s[0], s[1], s[2], 0.f);
            w.X0[pb + i] = p;
            for (int l = 0; l < d.L; ++l) w.ACC[(size_t)l * lay.Nm + pb + i] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const float* s = xh_pocket + (size_t)(qb + i - nl) * ldq;
            p = make_float4(s[0], s[1], s[2], 0.f);
            w.XP[qb + i - nl] = p;
            if (d.joint) {                  // joint mode: pocket nodes move as well (dynamics.py:105-107)
                const int n = lay.Nl + qb + i - nl;
                w.X0[n] = p;
                for (int l = 0; l < d.L; ++l) w.ACC[(size_t)l * lay.Nm + n] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        spos[i] = p;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    for (int i = wave; i < n; i += nwaves) {
        const float4 pi = spos[i];
        int cnt = 0, self = 0;
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            bool ok = false;
            if (j < n) {
                const float r2 = dist2(pi, spos[j]);
                ok = (d.cutoff2 < 0.f) || (r2 <= d.cutoff2);
            }
            const unsigned long long m = __ballot(ok);
            cnt += __popcll(m);
            if (i >= j0 && i < j0 + 64) self = (int)((m >> (i - j0)) & 1ull);
        }
        if (lane == 0) { sdeg[i] = cnt | (self << 30); w.degL[pb + qb + i] = cnt | (self << 30); }   // bit 30: the self loop exists
    }
    __syncthreads();
    if (wave == 0) {
        int e = 0, eph = 0, ens = 0, ensq = 0;
        for (int i = lane; i < n; i += 64) {
            const int dg = sdeg[i] & 0x3fffffff; e += dg;
            if (i < nl) { eph += dg; ens += dg - ((sdeg[i] >> 30) & 1); }
            else ensq += dg - ((sdeg[i] >> 30) & 1);
        }
        for (int o = 32; o > 0; o >>= 1) {
            e += __shfl_xor(e, o); eph += __shfl_xor(eph, o); ens += __shfl_xor(ens, o); ensq += __shfl_xor(ensq, o);
        }
        if (lane == 0) { w.pocketE[b] = e; w.pocketEph[b] = eph; w.pocketEns[b] = ens; w.pocketEnsQ[b] = ensq; }
    }
    if (b == 0 && tid == 0) {
        atomicAdd(&w.counters[0], 1ull);                       // evaluations
        atomicAdd(&w.counters[3], (unsigned long long)lay.N);  // nodes
    }
}

// (a device function: it is also the first B workgroups of k_write_embed)
__device__ __forceinline__ void edge_write_body(const Layout& lay, const Work& w, const Dims& d, const int b) {
    extern __shared__ float4 spos[];
    int* soff = reinterpret_cast<int*>(spos + lay.max_n);
    int* sdg = soff + lay.max_n;                 // the sample's degree words (k_edge_count), read many times below
    int* shop = sdg + lay.max_n;                 // hop levels (see below)
    __shared__ int s_base[5];
    __shared__ int s_red[6][16];
    const int tid = threadIdx.x;
    const int nl = lay.num_phar[b], np = lay.num_pocket[b], n = nl + np;
    const int pb = lay.phar_base[b], qb = lay.pocket_base[b];
    const int lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    for (int i = tid; i < n; i += blockDim.x) {
        spos[i] = i < nl ? w.X0[pb + i] : w.XP[qb + i - nl];
        sdg[i] = w.degL[pb + qb + i];
        // aggregation_method 'mean' (egnn_new.py:288-292): every segment sum of a node is divided by its edge count, self loop included
        if (d.agg_mean) w.adiv[flat_node(i, nl, pb, qb, lay.Nl)] = fmaxf((float)(sdg[i] & 0x3fffffff), 1.0f);
    }
    // The compact list is ordered like torch.where on the N x N adjacency of the flat node
    // numbering (dynamics.py:146): all phar receivers first (sample by sample), then all pocket
    // receivers.  So the phar-receiver edges - the only ones the coordinate update needs - are
    // the first Ec entries of the same list.
    int e = 0, eph = 0, ephall = 0, ens = 0, ensall = 0, ensq = 0;
    for (int k = tid; k < lay.B; k += blockDim.x) {
        const int pe = w.pocketE[k], pp = w.pocketEph[k], pn = w.pocketEns[k];
        ephall += pp; ensall += pn;
        if (k < b) { e += pe; eph += pp; ens += pn; ensq += w.pocketEnsQ[k]; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        e += __shfl_xor(e, o); eph += __shfl_xor(eph, o); ephall += __shfl_xor(ephall, o); ens += __shfl_xor(ens, o);
        ensall += __shfl_xor(ensall, o); ensq += __shfl_xor(ensq, o);
    }
    if (lane == 0) {
        s_red[0][wave] = e; s_red[1][wave] = eph; s_red[2][wave] = ephall; s_red[3][wave] = ens;
        s_red[4][wave] = ensall; s_red[5][wave] = ensq;
    }
    __syncthreads();
    if (tid == 0) {
        int te = 0, tp = 0, ta = 0, tn = 0, tna = 0, tq = 0;
        for (int k = 0; k < nwaves; ++k) {
            te += s_red[0][k]; tp += s_red[1][k]; ta += s_red[2][k]; tn += s_red[3][k]; tna += s_red[4][k]; tq += s_red[5][k];
        }
        s_base[0] = tp;                     // phar-receiver section: edges of earlier samples' phar rows
        s_base[1] = ta + (te - tp);         // pocket-receiver section starts after ALL phar-receiver edges
        s_base[2] = ta;
        s_base[3] = tn;                     // coordinate list (phar receivers, self loops dropped)
        s_base[4] = tna + tq;               // joint mode: its pocket-receiver section, same sectioning as the full list
    }
    // exclusive scan of the degrees inside the sample (wave 0, 64 at a time)
    if (wave == 0) {
        int carry = 0;
        for (int c = 0; c < n; c += 64) {
            const int i = c + lane;
            const int v = i < n ? (sdg[i] & 0x3fffffff) : 0;
            int s = v;
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(s, o); if (lane >= o) s += t; }
            if (i < n) soff[i] = carry + s - v;
            carry += __shfl(s, 63);
        }
    }
    __syncthreads();
    const int eph_b = w.pocketEph[b];
    // The coordinate update needs the phar-receiver edges WITHOUT the self loops: their coord_diff is
    // exactly (x_i - x_i)/(...) = 0, so they add exactly 0 to the sum (egnn_new.py:91, :265-271).
    // Offset of receiver i in that list = (edges before it) - (self loops before it); inside a sample the
    // phar rows come first, so the number of earlier rows is i.
    for (int i = wave; i < n; i += nwaves) {
        const float4 pi = spos[i];
        const int gi = flat_node(i, nl, pb, qb, lay.Nl);
        int off = i < nl ? s_base[0] + soff[i] : s_base[1] + (soff[i] - eph_b);
        int coff = 0;
        const bool moving = i < nl || d.joint;           // receivers whose coordinates are updated
        if (moving) {
            int selfs = 0;
            for (int k = (i < nl ? 0 : nl) + lane; k < i; k += 64) selfs += (sdg[k] >> 30) & 1;
            for (int o = 32; o > 0; o >>= 1) selfs += __shfl_xor(selfs, o);
            coff = i < nl ? s_base[3] + soff[i] - selfs : s_base[4] + (soff[i] - eph_b) - selfs;
        }
        bool feeds = moving;                             // does node i send along an edge of the coordinate list?  (dist2 is symmetric:
                                                         // i is a sender of a moving receiver j exactly when j is listed here as i's neighbour)
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            bool ok = false; float r2 = 0.f;
            if (j < n) {
                r2 = dist2(pi, spos[j]);
                ok = (d.cutoff2 < 0.f) || (r2 <= d.cutoff2);
            }
            const unsigned long long m = __ballot(ok);
            if (!moving) feeds = feeds || __ballot(ok && j < nl) != 0ull;
            if (ok) {
                const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
                const int gj = flat_node(j, nl, pb, qb, lay.Nl);
                w.erow[pos] = gi; w.ecol[pos] = gj; w.ed0[pos] = r2;
            }
            off += __popcll(m);
            if (moving) {
                const unsigned long long mc = __ballot(ok && j != i);
                if (ok && j != i) {
                    const int cpos = coff + __popcll(mc & ((1ull << lane) - 1ull));
                    w.crow[cpos] = gi; w.ccol[cpos] = flat_node(j, nl, pb, qb, lay.Nl); w.cd0[cpos] = r2;
                }
                coff += __popcll(mc);
            }
        }
        // Hop distance from the moving nodes along the graph's edges: 0 = moves, 1 = sends along a coordinate edge (its Q_c row is read), and
        // below the levels 2 .. L; 255 = none of those.  k_node64 computes the Q_c rows of a tile only if it holds a node of level <= 1, and
        // block l of a conditional evaluation whose pocket output nobody asks for only needs the nodes of level <= L - l (see edge_msg_body).
        if (lane == 0 && w.need_qc) { const int lvl = moving ? 0 : feeds ? 1 : 255; shop[i] = lvl; w.need_qc[gi] = lvl; }
    }
    if (w.need_qc) {
        // levels 2 .. hop_levels: a node not reached yet joins level k when one of its neighbours is at level k - 1 (positions still in LDS)
        __shared__ int s_any[2];
        __shared__ int s_near;
        if (tid == 0) { s_any[0] = 0; s_any[1] = 0; s_near = 0; }
        __syncthreads();                                                        // levels 0 / 1 / 255 of every node are in shop (LDS)
        // Where at least half of the sample already sits at level <= 1 (the phar points are inside the pocket) the sweep below would reach
        // everybody within a level or two and buy nothing: call the rest level 2 - conservative (a node is never skipped while it is needed),
        // and the ~3 us the sweep costs a 59-node sample stay off the critical path of k_write_embed
        if (w.hop_levels > 1 && !d.joint) {
            int near = 0, ones = 0;                                              // per wave: nodes at level <= 1 / exactly 1 (the sweep's first frontier)
            for (int i0 = 0; i0 < n; i0 += (int)blockDim.x) {
                const int i = i0 + tid;
                const int lv = i < n ? shop[i] : 255;
                near += __popcll(__ballot(lv <= 1)); ones += __popcll(__ballot(lv == 1));
            }
            if (lane == 0 && near) atomicAdd(&s_near, near);
            if (lane == 0 && ones) s_any[1] = 1;                                 // (level 2 looks at s_any[(2 - 1) & 1] below)
            __syncthreads();
            if (2 * s_near >= n) {
                for (int i = tid; i < n; i += blockDim.x) if (shop[i] == 255) { shop[i] = 2; w.need_qc[flat_node(i, nl, pb, qb, lay.Nl)] = 2; }
                __syncthreads();
            }
        }
        // (an empty frontier ends the sweep before it starts: in a drifted chain no pocket node is within reach of a phar point)
        for (int level = 2; level <= w.hop_levels && !d.joint && 2 * s_near < n && s_any[(level - 1) & 1]; ++level) {
            for (int i = wave; i < n; i += nwaves) {
                if (shop[i] != 255) continue;                                   // wave-uniform
                const float4 pi = spos[i];
                bool hit = false;
                for (int j0 = 0; j0 < n && !hit; j0 += 64) {
                    const int j = j0 + lane;
                    const bool ok = j < n && shop[j] == level - 1 && ((d.cutoff2 < 0.f) || dist2(pi, spos[j]) <= d.cutoff2);
                    hit = __ballot(ok) != 0ull;
                }
                if (hit && lane == 0) { shop[i] = level; w.need_qc[flat_node(i, nl, pb, qb, lay.Nl)] = level; s_any[level & 1] = 1; }
            }
            __syncthreads();
            if (!s_any[level & 1]) break;                                       // nothing joined: nothing will
            if (tid == 0) s_any[(level + 1) & 1] = 0;
            __syncthreads();
        }
        // the receiver's level beside every listed edge: k_edge_msg fetches it with the tile's (row, col, d0) one tile ahead - no dependent load
        if (w.ehop) {
            __syncthreads();
            for (int i = wave; i < n; i += nwaves) {
                const int off = i < nl ? s_base[0] + soff[i] : s_base[1] + (soff[i] - eph_b);
                const int dg = sdg[i] & 0x3fffffff, lvl = shop[i];
                for (int e = lane; e < dg; e += 64) w.ehop[off + e] = lvl;
            }
        }
    }
    if (b == 0 && tid == 0) *w.nan_flag = 0;     // after every reader of the previous evaluation's flag, before k_readout sets it
    if (b == lay.B - 1 && tid == 0) {
        const int E = s_base[1] + (w.pocketE[b] - eph_b);
        const int Ec = d.joint ? s_base[4] + w.pocketEnsQ[b] : s_base[3] + w.pocketEns[b];
        w.totals[0] = E; w.totals[1] = Ec;
        atomicAdd(&w.counters[1], (unsigned long long)E);
        atomicAdd(&w.counters[2], (unsigned long long)Ec);
    }
}
__global__ void k_edge_write(Layout lay, Work w, Dims d) { edge_write_body(lay, w, d, blockIdx.x); }

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
// k_embed: h0 = embedding([encoder(features) | t]) for an MT-node tile, then P/Q of block 0.
// Encoders are tiny (8->16->32, R->2R->32): plain FMA loops through LDS.
// ------------------------------------------------------------------------------------
template <int H, int MT, bool SP>
__device__ __forceinline__ void embed_body(const Layout& lay, const Work& w, const Dims& d, const SmallW& sw, const LayerW& lw0,
                                           const float* __restrict__ xh_phar,
                                           const float* __restrict__ xh_pocket,
                                           const float* __restrict__ t_arr,
                                           const float4* __restrict__ coef, const ChainState* chain, const TrainSave& sv,
                                           const PocketCache& pc, const int blk, const int part = 2) {
    // part: 2 = the whole tile; 0 / 1 = one workgroup of a PAIR that shares a full-path tile inside a chain: both run the
    // encoders and the embedding (cheap), 0 writes h and projects P, 1 projects Q - the two 16-row projection passes were 40 %
    // of the tile's critical path when one workgroup ran them back to back (profiles/r02_b_step_fusion.txt, cycle stamps)
    if (pc.c != nullptr && blk * MT >= lay.Nl) {
        // A tile of pocket rows inside a conditional chain: nothing but the time feature has changed since the chain
        // started (SURVEY section 7 "Static structure"), so h, P and Q are one fused multiply-add per element from the
        // cache built at the chain's start (cmdgen_sample_chain) - no encoder, no embedding, no GEMM.
        const int row0 = blk * MT, nvalid = min(MT, lay.N - row0);
        const float t = coef[chain->step].w;
        constexpr int LPR = H / 4;
        const int c4 = threadIdx.x % LPR, rsub = threadIdx.x / LPR;
        const float4 dh = reinterpret_cast<const float4*>(pc.dh)[c4], dP = reinterpret_cast<const float4*>(pc.dP)[c4],
                     dQ = reinterpret_cast<const float4*>(pc.dQ)[c4];
        auto axpy = [&](const float4& a, const float4& b) { return make_float4(fmaf(t, b.x, a.x), fmaf(t, b.y, a.y), fmaf(t, b.z, a.z), fmaf(t, b.w, a.w)); };
#pragma unroll 4
        for (int r = rsub; r < nvalid; r += 4) {
            const size_t q = (size_t)(row0 + r - lay.Nl) * LPR + c4, o = (size_t)(row0 + r) * LPR + c4;
            reinterpret_cast<float4*>(w.h)[o] = axpy(reinterpret_cast<const float4*>(pc.c)[q], dh);
            reinterpret_cast<float4*>(w.P)[o] = axpy(reinterpret_cast<const float4*>(pc.P0)[q], dP);
            reinterpret_cast<float4*>(w.Q)[o] = axpy(reinterpret_cast<const float4*>(pc.Q0)[q], dQ);
        }
        return;
    }
#if CMDGEN_STAMPS == 3
    unsigned long long est_[8] = {0,0,0,0,0,0,0,0}, est_t = __builtin_amdgcn_s_memtime(); const unsigned long long est_b = est_t;
#define ESTAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); est_[i] += n_ - est_t; est_t = n_; } while (0)
#else
#define ESTAMP(i) do {} while (0)
#endif
    __shared__ __attribute__((aligned(16))) float buf[MT * LDA(H)];
    __shared__ float s_in[MT][CMDGEN_MAX_SMALL];
    __shared__ float s_h1[MT][CMDGEN_MAX_SMALL];
    __shared__ float s_h2[MT][CMDGEN_MAX_SMALL + 1];
    const int tid = threadIdx.x, nthr = H;
    const int row0 = blk * MT;
    const int nvalid = min(MT, lay.N - row0);
    const int ldp = 3 + d.P, ldq = 3 + d.R;
    const int Fmax = max(d.P, d.R), F1max = 2 * Fmax;
    typedef Eng<MT, SP> G;
    typename G::Carry carry;                           // weight fragments of the projection, in flight during the encoders
    const typename G::Frag f0 = G::frag(lw0.Wpq_e, H / 8, 0, (part == 1 ? H / 64 : 0) + (tid >> 6));
    G::prefetch(f0, carry);
    const ColVec<MT> b1v = col_load<MT>(lw0.b1, tid >> 6);       // needed by the projection's epilogue four phases later
    const float t_chain = t_arr ? 0.f : coef[chain->step].w;     // two dependent loads: issued now, needed three phases later
    // The eight encoder tensors (2.8k floats at the shipped sizes) are copied into LDS first, sixteen loads per thread in
    // flight at a time: the FMA loops below then read them at LDS latency.  Read from global inside those loops they
    // cost one dependent L2 round trip per unrolled batch (3 passes x up to 10 batches - most of this kernel's time).
    extern __shared__ float s_enc[];
    const int seg_n[8] = {2 * d.P * d.P, 2 * d.P, d.J * 2 * d.P, d.J, 2 * d.R * d.R, 2 * d.R, d.J * 2 * d.R, d.J};
    const float* const seg_p[8] = {sw.pe0_w, sw.pe0_b, sw.pe2_w, sw.pe2_b, sw.re0_w, sw.re0_b, sw.re2_w, sw.re2_b};
    int seg_o[9];
    seg_o[0] = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) seg_o[q + 1] = seg_o[q] + seg_n[q];
    // the tile's input features: requested first (registers), written to LDS after the encoder tensors, so that both sets of
    // loads are in flight together (written where they were loaded, the second set waited for the first: two round trips)
    constexpr int NIN = (MT * CMDGEN_MAX_SMALL + H - 1) / H;            // upper bound of (row, feature) pairs per thread
    float vin[NIN];
#pragma unroll
    for (int q = 0; q < NIN; ++q) {
        const int idx = tid + q * nthr;
        vin[q] = 0.f;
        if (idx < MT * Fmax) {
            const int r = idx / Fmax, k = idx - r * Fmax, n = row0 + r;
            if (r < nvalid) {
                if (n < lay.Nl) { if (k < d.P) vin[q] = xh_phar[(size_t)n * ldp + 3 + k]; }
                else if (k < d.R) vin[q] = xh_pocket[(size_t)(n - lay.Nl) * ldq + 3 + k];
            }
        }
    }
    if (sw.enc_pack) {          // sampler: the eight tensors lie contiguous in one device buffer (cmdgen_finalize_weights)
        for (int i = tid; i < seg_o[8]; i += nthr) s_enc[i] = sw.enc_pack[i];
    } else
    for (int base = 0; base < seg_o[8]; base += 16 * nthr) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int gi = base + q * nthr + tid;
            v[q] = 0.f;
            if (gi < seg_o[8]) {
                int sg = 0;
#pragma unroll
                for (int u = 1; u < 8; ++u) sg += gi >= seg_o[u];
                v[q] = seg_p[sg][gi - seg_o[sg]];
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int gi = base + q * nthr + tid;
            if (gi < seg_o[8]) s_enc[gi] = v[q];
        }
    }
    const float *pe0w = s_enc + seg_o[0], *pe0b = s_enc + seg_o[1], *pe2w = s_enc + seg_o[2], *pe2b = s_enc + seg_o[3];
    const float *re0w = s_enc + seg_o[4], *re0b = s_enc + seg_o[5], *re2w = s_enc + seg_o[6], *re2b = s_enc + seg_o[7];
#pragma unroll
    for (int q = 0; q < NIN; ++q) {                       // one (row, feature) pair per thread and slot
        const int idx = tid + q * nthr;
        if (idx < MT * Fmax) { const int r = idx / Fmax; s_in[r][idx - r * Fmax] = vin[q]; }
    }
    lds_barrier();
    ESTAMP(0);
    // encoder layer 0 + SiLU: thread -> (row r, output o)
    for (int idx = tid; idx < MT * F1max; idx += nthr) {
        const int r = idx / F1max, o = idx - r * F1max;
        const int n = row0 + r;
        if (r >= nvalid) continue;
        const bool ph = n < lay.Nl;
        const int F = ph ? d.P : d.R;
        if (o >= 2 * F) continue;
        const float* W = (ph ? pe0w : re0w) + o * F;
        float s = (ph ? pe0b : re0b)[o];
#pragma unroll 4
        for (int k = 0; k < F; ++k) s = fmaf(s_in[r][k], W[k], s);
        const float act = silu_f(s);
        s_h1[r][o] = act;
        if (sv.enc1_l) {                                   // training: layer-0 pre-activation and activation
            if (ph) { sv.enc1_l[(size_t)n * 2 * F + o] = s; sv.enca_l[(size_t)n * 2 * F + o] = act; }
            else { sv.enc1_p[(size_t)(n - lay.Nl) * 2 * F + o] = s; sv.enca_p[(size_t)(n - lay.Nl) * 2 * F + o] = act; }
        }
    }
    lds_barrier();
    ESTAMP(1);
    // encoder layer 2 -> joint space, then the time column (dynamics.py:92-99)
    for (int idx = tid; idx < MT * d.dyn; idx += nthr) {
        const int r = idx / d.dyn, j = idx - r * d.dyn;
        const int n = row0 + r;
        float s = 0.f;
        if (r < nvalid) {
            if (j < d.J) {
                const bool ph = n < lay.Nl;
                const int F2 = 2 * (ph ? d.P : d.R);
                const float* W = (ph ? pe2w : re2w) + j * F2;
                s = (ph ? pe2b : re2b)[j];
#pragma unroll 4
                for (int k = 0; k < F2; ++k) s = fmaf(s_h1[r][k], W[k], s);
            } else {
                s = t_arr ? t_arr[lay.node_sample[n]] : t_chain;
            }
            if (sv.hdyn) sv.hdyn[(size_t)n * d.dyn + j] = s;
        }
        s_h2[r][j] = s;
    }
    lds_barrier();
    ESTAMP(2);
    {   // embedding dyn -> H: one output column per thread, weights transposed [dyn][H] (coalesced)
        const int c = tid;
        const float bc = sw.emb_b[c];
        float accr[MT];
#pragma unroll
        for (int r = 0; r < MT; ++r) accr[r] = bc;
        for (int k0 = 0; k0 < d.dyn; k0 += 16) {       // sixteen weight loads in flight, then their FMAs (k ascending as before)
            float wk[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) wk[j] = k0 + j < d.dyn ? sw.emb_wT[(size_t)(k0 + j) * H + c] : 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (k0 + j < d.dyn) {
#pragma unroll
                    for (int r = 0; r < MT; ++r) accr[r] = fmaf(s_h2[r][k0 + j], wk[j], accr[r]);
                }
        }
#pragma unroll
        for (int r = 0; r < MT; ++r) {
            const float s = r < nvalid ? accr[r] : 0.f;
            buf[r * LDA(H) + c] = s;
            if (r < nvalid && part != 1) {
                w.h[(size_t)(row0 + r) * H + c] = s;
                if (sv.h) sv.h[(size_t)(row0 + r) * H + c] = s;      // h entering block 0
            }
        }
    }
    lds_barrier();
    ESTAMP(3);
    tile_project_pq<H, MT, SP>(buf, lw0.Wpq_e, b1v, w.P, w.Q, row0, nvalid, part != 1, carry, f0, part != 0);
    ESTAMP(4);
#if CMDGEN_STAMPS == 3
    if ((threadIdx.x & 63) == 0) { const int wv = threadIdx.x >> 6; for (int i = 0; i < 5; ++i) atomicAdd(&w.dbg[wv * 8 + i], est_[i]);
        atomicAdd(&w.dbg[32 + wv], __builtin_amdgcn_s_memtime() - est_b); atomicAdd(&w.dbg[40], 1ull); }
#endif
#undef ESTAMP
}
template <int H, int MT, bool SP>
__global__ __launch_bounds__(H) void k_embed(Layout lay, Work w, Dims d, SmallW sw, LayerW lw0, const float* __restrict__ xh_phar,
                                             const float* __restrict__ xh_pocket, const float* __restrict__ t_arr,
                                             const float4* __restrict__ coef, const ChainState* chain, TrainSave sv, PocketCache pc) {
    embed_body<H, MT, SP>(lay, w, d, sw, lw0, xh_phar, xh_pocket, t_arr, coef, chain, sv, pc, (int)blockIdx.x);
}
// Pass 2 of the radius graph (one workgroup per sample, reads positions and degrees) and k_embed (node tiles, reads features and
// the time) do not depend on each other and are both latency chains of a few workgroups per CU: ONE launch runs them side by
// side - workgroups 0 .. B-1 write the edge lists, the rest are embedding tiles - instead of two dependent launches (a fork /
// join on two streams inside the replayed graph cost more than it hid, profiles/r02_b_step_fusion.txt).  H = 256 only: both
// bodies are written for 256 threads.
template <int MT, bool SP>
__global__ __launch_bounds__(256) void k_write_embed(Layout lay, Work w, Dims d, SmallW sw, LayerW lw0, const float* __restrict__ xh_phar,
                                                     const float* __restrict__ xh_pocket, const float* __restrict__ t_arr,
                                                     const float4* __restrict__ coef, const ChainState* chain, PocketCache pc, int npair) {
    // workgroups: [0, B) edge lists | [B, B + 2 npair) pairs over the first npair tiles (the full-path tiles of a chain) |
    // the rest: one workgroup per remaining tile
    const int i = (int)blockIdx.x - lay.B;
    if (i < 0) edge_write_body(lay, w, d, (int)blockIdx.x);
    else if (i < 2 * npair) embed_body<256, MT, SP>(lay, w, d, sw, lw0, xh_phar, xh_pocket, t_arr, coef, chain, TrainSave{}, pc, i >> 1, i & 1);
    else embed_body<256, MT, SP>(lay, w, d, sw, lw0, xh_phar, xh_pocket, t_arr, coef, chain, TrainSave{}, pc, i - npair, 2);
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
                reinterpret_cast<float4*>(act_out + (size_t)e * H)[c4] = a;
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
                *reinterpret_cast<float4*>(act_out + (size_t)e * H + col) = a;
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
                reinterpret_cast<float4*>(act_out + (size_t)e * H)[c4] = a;
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

// ------------------------------------------------------------------------------------
// edge_msg_body / k_edge_msg: GCL.edge_model + attention gate + segment sum by receiver for MT-edge tiles
// of the compact list.  Persistent-style grid: tiles are taken round-robin until the
// device-side edge count is exhausted, so the launch geometry is static (graph-capturable).
// live_thr: 0 = every tile; else only tiles with a receiver within that many hops of a moving node (dead work).
// ------------------------------------------------------------------------------------
template <int H, int MT, bool SAVE, bool SP, int FK = 0>
__device__ __forceinline__ void edge_msg_body(EdgeLds<H, MT, FK>& L, const Layout& lay, const Work& w, const Dims& d, const LayerW& lw,
                                              const int layer, const int ablate, const TrainSave& sv, const int live_thr) {
    float* buf = L.buf; int* s_row = L.s_row; int* s_col = L.s_col;
    float* s_r = L.s_r; float* s_d0 = L.s_d0; float* s_att = L.s_att; float* s_wa = L.s_vec; float* s_wrd = L.s_wrd;
    const int tid = threadIdx.x, wave = tid >> 6;
    s_wa[tid] = lw.wa[tid];                                    // visible after the first tile's barriers
    constexpr bool PL = SP && H == 256 && MT >= 32;            // plane variant: the producer splits (build_edge_half)
    static_assert(!FK || (PL && MT == 32 && (!SAVE || FK == 2)), "full-K planes: 32-row tiles on the split engine (training forward: the half engine only)");
    if constexpr (PL && !FK) { s_wrd[tid] = lw.wr_e[tid]; s_wrd[H + tid] = lw.wd_e[tid]; }
    const ColVec<MT> b2v = col_load<MT>(lw.b2, wave);          // per-column bias and the gate's bias: once per workgroup
    const float ba0 = lw.ba[0];
    const float4 wr4 = reinterpret_cast<const float4*>(lw.wr_e)[tid % (H / 4)], wd4 = reinterpret_cast<const float4*>(lw.wd_e)[tid % (H / 4)];
    extern __shared__ float s_dyn[];          // sin_embedding only (launched with (24 H + 24 MT) floats): the [24][H] feature columns of edge_mlp.0, then the tile's features
    if constexpr (!FK && !(SP && H == 256 && MT >= 32))      // (the plane variants never see sin_embedding: such a handle runs on the fp32 instruction)
        if (d.sin) for (int i = tid; i < 24 * H; i += H) s_dyn[i] = lw.we_e[i];          // (the first tile's barrier covers it)
    typedef typename EdgeEng<MT, SP, FK>::G G;
    float inv2 = 1.0f;                                         // the half engine's accumulators carry the weight pack's power-of-two scale
    if constexpr (FK != 0) inv2 = SAVE ? lw.W2.wh_dev[1] : G::inv(lw.W2);      // (training: the pack and its scale are re-made on the device every step)
    const typename G::Frag fw = G::frag(lw.W2, H / 8, 0, wave);
    typename G::Carry carry;
    G::prefetch(fw, carry);     // before the edge count is known: the first fragments fly beside that load and the index / position
                                // / gather round trips of the first tile (a workgroup that finds no tile has read 12-24 KB for nothing);
                                // refilled for the next tile by each GEMM's last iteration
    const int E = w.totals[0];
    const int ntiles = (E + MT - 1) / MT;
#if CMDGEN_STAMPS == 1
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_t;
#define STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_t; st_t = n_; } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
    // The (row, col, d0) triple of a tile is requested one tile ahead (three registers): it arrives during the previous tile's
    // build / GEMM, so a tile's first phase starts with the position loads instead of two dependent round trips.
    int nx_row = -1, nx_col = -1, nx_hop = 255; float nx_d0 = 0.f;
    {
        const int t0 = xcd_tile(0, ntiles);
        if (t0 >= 0 && tid < MT && t0 * MT + tid < E) {
            nx_row = w.erow[t0 * MT + tid]; nx_col = w.ecol[t0 * MT + tid]; nx_d0 = w.ed0[t0 * MT + tid];
            if (live_thr) nx_hop = w.ehop[t0 * MT + tid];
        }
    }
    for (int k = 0, tile; (tile = xcd_tile(k, ntiles)) >= 0; ++k) {
        const int e0 = tile * MT;
        const int ne = min(MT, E - e0);
        if (tid < MT) {
            const int row = nx_row, col = nx_col, hop = nx_hop; const float d0 = nx_d0;       // -1 / -1 / 255 / 0 beyond the list's end
            nx_row = -1; nx_col = -1; nx_d0 = 0.f; nx_hop = 255;
            const int tn = xcd_tile(k + 1, ntiles);
            if (tn >= 0 && tn * MT + tid < E) {
                nx_row = w.erow[tn * MT + tid]; nx_col = w.ecol[tn * MT + tid]; nx_d0 = w.ed0[tn * MT + tid];
                if (live_thr) nx_hop = w.ehop[tn * MT + tid];
            }
            if (live_thr) {
                // A conditional evaluation whose pocket output nobody reads: after the last block only the moving nodes' h and the Q_c rows of
                // the coordinate senders are used, so block l needs the new h of the nodes within L - l hops of a moving node only (levels of
                // the graph pass; live_thr = L - l) - a tile without such a receiver is dead work
                const bool live = hop <= live_thr;
                const unsigned long long any = __ballot(live);
                if (tid == 0) L.s_live[k & 1] = any != 0ull;
            }
            float r = 0.f;
            if (tid < ne) {
                // block 0 sees the input positions: its radial IS the d0 the graph pass stored (same dist2, same operands, same
                // bits) - no position round trip; later blocks form the lazily updated positions (node_pos)
                r = ((ablate & 1) || layer == 0) ? d0 : dist2(node_pos(lay, w, d, row, layer, true), node_pos(lay, w, d, col, layer, true));
            }
            float d0f = d0;
            if constexpr (!FK && !(SP && H == 256 && MT >= 32))
            if (d.sin && tid < ne) {                        // the features' distances in the reference's rounding (sumsq_ref)
                d0f = sumsq_ref(node_pos(lay, w, d, row, 0, true), node_pos(lay, w, d, col, 0, true));
                r = layer == 0 ? d0f : sumsq_ref(node_pos(lay, w, d, row, layer, true), node_pos(lay, w, d, col, layer, true));
            }
            s_row[tid] = row; s_col[tid] = col; s_r[tid] = r; s_d0[tid] = d0f;
        }
        lds_barrier();
        STAMP(0);
        if (live_thr && !L.s_live[k & 1]) {                        // (s_live is double-buffered: no thread is two tiles behind)
            if (tid == 0) atomicAdd(&w.counters[6], (unsigned long long)ne);
            continue;
        }
        TileAcc<MT> acc;
        acc_zero<MT>(acc);
        if constexpr (FK) {
            unsigned short* planes = reinterpret_cast<unsigned short*>(buf);
            if (!(ablate & 2)) build_edge_full32<FK>(planes, s_row, s_col, s_r, s_d0, ne, w.P, w.Q, wr4, wd4,
                                                     SAVE ? sv.pre1 + ((size_t)layer * sv.ecap + e0) * H : nullptr,
                                                     SAVE ? sv.act1 + ((size_t)layer * sv.ecap + e0) * H : nullptr);
            lds_barrier();
            STAMP(1);
            if (!(ablate & 4)) G::gemm(planes, fw, acc.a, carry);
        } else if constexpr (PL) {
            // two half-K passes: build columns [0,128) as bf16 planes -> GEMM over k 0..127 -> build [128,256) -> GEMM over the rest
            unsigned short* planes = reinterpret_cast<unsigned short*>(buf);
            constexpr int PLDA = SPLIT_PLANE_LDA(H / 2), PE = MT * PLDA;
            const typename G::Frag fw1 = G::frag(lw.W2, H / 8, H / 16, wave);
            float* pre1_o = SAVE ? sv.pre1 + ((size_t)layer * sv.ecap + e0) * H : nullptr;
            float* act1_o = SAVE ? sv.act1 + ((size_t)layer * sv.ecap + e0) * H : nullptr;
            if (!(ablate & 2)) build_edge_half<MT>(planes, 0, s_row, s_col, s_r, s_d0, ne, w.P, w.Q, s_wrd, s_wrd + H, pre1_o, act1_o);
            lds_barrier();
            if (!(ablate & 4)) tile_gemm_planes<MT, H / 32>(planes, PE, PLDA, fw, fw1, acc.a, carry);
            lds_barrier();
            if (!(ablate & 2)) build_edge_half<MT>(planes, 1, s_row, s_col, s_r, s_d0, ne, w.P, w.Q, s_wrd, s_wrd + H, pre1_o, act1_o);
            lds_barrier();
            STAMP(1);
            if (!(ablate & 4)) tile_gemm_planes<MT, H / 32>(planes, PE, PLDA, fw1, fw, acc.a, carry);
        } else {
        if (d.sin) { sin_features<H, MT>(s_dyn + 24 * H, s_r, s_d0, ne, d); lds_barrier(); }
        if (!(ablate & 2)) build_edge_tile<H, MT>(buf, s_row, s_col, s_r, s_d0, ne, w.P, w.Q, wr4, wd4,
                                                  SAVE ? sv.pre1 + ((size_t)layer * sv.ecap + e0) * H : nullptr,
                                                  SAVE ? sv.act1 + ((size_t)layer * sv.ecap + e0) * H : nullptr,
                                                  d.sin ? s_dyn + 24 * H : nullptr, s_dyn);
        lds_barrier();
        STAMP(1);
        if (!(ablate & 4)) G::template gemm<H / 8>(buf, LDA(H), fw, fw, acc, carry);
        }
        STAMP(2);
        lds_barrier();                         // every wave is done reading the A tile
        STAMP(3);
        acc_foreach_n<MT>(acc, wave, [&](int row, int col, int n, float v) {                 // m_ij
            const float pre = __fmaf_rn(v, inv2, b2v.v[n]);
            buf[row * LDA(H) + col] = SAVE ? pre : silu_f(pre);
        });
        lds_barrier();
        if constexpr (SAVE) {
            const size_t o = ((size_t)layer * sv.ecap + e0) * H;
            save_rows_silu<H, MT>(buf, ne, sv.pre2 + o, sv.act2 ? sv.act2 + o : nullptr);
            lds_barrier();
        }
        STAMP(4);
        if (!(ablate & 16)) {   // attention gate: sigmoid(w_a . m_ij + b_a)
            int r; bool lead;
            const float s = tile_row_dot<H, MT>(buf, s_wa, r, lead);
            if (lead) {
                const float zl = s + ba0;
                s_att[r] = d.attention ? sigmoid_f(zl) : 1.0f;
                if (SAVE && d.attention && r < ne) sv.z[(size_t)layer * sv.ecap + e0 + r] = zl;
            }
        }
        lds_barrier();
        STAMP(5);
        if (!(ablate & 8)) {
            // Segment sum over the tile's rows, one column per thread, edge order preserved
            // (= the reference's sequential scatter_add_).  All LDS reads are issued up front
            // (independent, pipelined); the scan itself runs on registers under scalar control
            // flow driven by a ballot of the segment starts.
            const int c = tid;
            float v[MT];
#pragma unroll
            for (int e = 0; e < MT; ++e) v[e] = buf[e * LDA(H) + c] * s_att[e];
            const int lane = tid & 63;
            const bool st = lane < ne && (lane == 0 || s_row[lane] != s_row[lane > 0 ? lane - 1 : 0]);
            const unsigned long long starts = __ballot(st);          // bit e: row e begins a receiver segment
            float sum = 0.f;
            int seg0 = 0;
#pragma unroll
            for (int e = 0; e < MT; ++e) {
                if (e < ne) {
                    if (e > 0 && ((starts >> e) & 1ull)) {            // wave-uniform: flush the finished segment
                        float* dst = w.agg + (size_t)s_row[seg0] * H + c;
                        if (seg0 == 0) atomicAdd(dst, sum); else *dst = sum;   // a segment may continue from the previous tile
                        seg0 = e; sum = 0.f;
                    }
                    sum += v[e];
                }
            }
            atomicAdd(w.agg + (size_t)s_row[seg0] * H + c, sum);     // ... or into the next one
        }
        lds_barrier();
        STAMP(6);
    }
#if CMDGEN_STAMPS == 1
    // lane 0 of every wave of a SAMPLE of the workgroups that had a tile (every 4th: thousands of same-address atomics per launch would
    // sit in front of the next launch's first loads): [wave][phase] sums, [32 + wave] = wave lifetime, [40] = waves, [41] = tile visits
    if ((tid & 63) == 0 && (blockIdx.x & 3) == 0 && xcd_tile(0, ntiles) >= 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&w.dbg[wave * 8 + i], st_[i]);
        atomicAdd(&w.dbg[32 + wave], __builtin_amdgcn_s_memtime() - st_begin);
        atomicAdd(&w.dbg[40], 1ull);
        if (wave == 0) { int nt = 0; while (xcd_tile(nt, ntiles) >= 0) ++nt; atomicAdd(&w.dbg[41], (unsigned long long)nt); }
    }
#endif
#undef STAMP
}
template <int H, int MT, bool SAVE, bool SP, int FK = 0>
__global__ __launch_bounds__(H, FK ? 3 : 2) void k_edge_msg(Layout lay, Work w, Dims d, LayerW lw, int layer, int ablate, TrainSave sv, int live_thr) {
    __shared__ __attribute__((aligned(16))) EdgeLds<H, MT, FK> L;
    edge_msg_body<H, MT, SAVE, SP, FK>(L, lay, w, d, lw, layer, ablate, sv, live_thr);
}

// ------------------------------------------------------------------------------------
// k_node: GCL.node_model for an MT-node tile (egnn_new.py:48-58)
//   h <- h + W4 SiLU(W3 [h | agg/nf] + b3) + b4
// then, while the new h tile is still in LDS, the projections every later kernel of this
// evaluation gathers: P_c|Q_c for this block's coord MLP and P|Q for block l+1's edge MLP.
// ------------------------------------------------------------------------------------
// node_tile_body: the tile of rows row0 .. min(row0 + MT, row_end) - 1; bufs: (MT <= 32 ? 2 : 1) * MT * LDA(H) floats of LDS.
template <int H, int MT, bool SAVE, bool SP>
__device__ __forceinline__ void node_tile_body(float* bufs, const Layout& lay, const Work& w, const Dims& d, const LayerW& lw, const LayerW& lw_next,
                                               const int layer, const int has_next_arg, const TrainSave& sv, const int row0, const int row_end) {
    const int has_next = has_next_arg & 1;                                     // (bits 1..29 carry the dead-tile threshold of the plane tiles: unused here)
    const bool skip_pc = ((has_next_arg >> 30) & 1) != 0;                     // not the last GCL of its block (inv_sublayers > 1): no P_c | Q_c
    // Tiles of <= 32 rows keep two LDS images: buf0 = h (kept for the residual), buf1 = agg -> T -> h_new,
    // so h and agg are fetched together and the residual needs no second global read.  64-row tiles
    // (66 KB each) use one image so that two workgroups still fit a CU.
    constexpr bool TWO = MT <= 32;
    float* buf0 = bufs;
    float* buf1 = TWO ? bufs + MT * LDA(H) : bufs;
    constexpr int LPR = H / 4;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int nvalid = min(MT, row_end - row0);
    const int c4 = tid % LPR, rsub = tid / LPR;
    // the chain of GEMMs of this tile; each one's last iteration fetches the next one's first fragments
    const bool want_pc = row0 < lay.Nm;              // the tile holds receivers that move
    // (Dead tiles - edge_msg_body - are skipped by the plane tiles of cmdgen_node_planes.h only: the launches of this body end with their phar
    // tiles, which are never dead, and the level check cost them 0.7 us at 64 pockets; profiles/r03_m_node64.txt.)
    typedef Eng<MT, SP> G;
    typedef typename G::Frag Frag;
    const Frag f3a = G::frag(lw.W3, 2 * H / 8, 0, wave), f3b = G::frag(lw.W3, 2 * H / 8, H / 8, wave);
    const Frag f4 = G::frag(lw.W4, H / 8, 0, wave);
    const Frag fn = G::frag(lw_next.Wpq_e, H / 8, 0, wave);
    const Frag fc = skip_pc ? fn : G::frag(lw.Wpq_c, H / 8, 0, want_pc ? wave : H / 64 + wave);      // the GEMM behind W4
    typename G::Carry carry;
    G::prefetch(f3a, carry);
    // every epilogue's bias, fetched now: by the time an epilogue runs its values have long arrived (a load issued where
    // it is used costs that epilogue an L2 round trip: k_node 38.3 -> 34.2 us at B=64)
    const ColVec<MT> b3v = col_load<MT>(lw.b3, wave), b4v = col_load<MT>(lw.b4, wave), b6v = col_load<MT>(lw.b6, wave),
                     b1nv = col_load<MT>(lw_next.b1, wave);
    // materialise the phar coordinates entering this block (see node_pos)
    if (layer >= 1 && tid < MT) {
        const int n = row0 + tid;
        if (tid < nvalid && n < lay.Nm) w.XL[(size_t)layer * lay.Nm + n] = node_pos(lay, w, d, n, layer, true);
    }
    auto load_h = [&]() {
#pragma unroll 4
        for (int pass = 0; pass < MT / 4; ++pass) {
            const int r = pass * 4 + rsub;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nvalid) v = reinterpret_cast<const float4*>(w.h + (size_t)(row0 + r) * H)[c4];
            *reinterpret_cast<float4*>(buf0 + r * LDA(H) + 4 * c4) = v;
        }
    };
    auto load_agg = [&]() {
#pragma unroll 4
        for (int pass = 0; pass < MT / 4; ++pass) {
            const int r = pass * 4 + rsub;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nvalid) {
                float4* g = reinterpret_cast<float4*>(w.agg + (size_t)(row0 + r) * H) + c4;
                v = *g;
                *g = make_float4(0.f, 0.f, 0.f, 0.f);                      // agg is zero between blocks
                const float dv = agg_div(w, d, row0 + r);
                v.x /= dv; v.y /= dv; v.z /= dv; v.w /= dv;
                if (SAVE) reinterpret_cast<float4*>(sv.aggn + ((size_t)layer * lay.N + row0 + r) * H)[c4] = v;
            }
            *reinterpret_cast<float4*>(buf1 + r * LDA(H) + 4 * c4) = v;
        }
    };
#if CMDGEN_STAMPS == 2      // diagnostic build: per-phase cycle stamps of this kernel into w.dbg (same layout as k_edge_msg's)
    unsigned long long nst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nst_t = __builtin_amdgcn_s_memtime();
    const unsigned long long nst_begin = nst_t;
#define NSTAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); nst_[i] += n_ - nst_t; nst_t = n_; } while (0)
#else
#define NSTAMP(i) do {} while (0)
#endif
    TileAcc<MT> acc;
    acc_zero<MT>(acc);
    if constexpr (TWO) {
        // all global loads of both images in flight together, then the LDS writes
        float4 hv[MT / 4], av[MT / 4];
#pragma unroll
        for (int pass = 0; pass < MT / 4; ++pass) {
            const int r = pass * 4 + rsub;
            hv[pass] = make_float4(0.f, 0.f, 0.f, 0.f); av[pass] = hv[pass];
            if (r < nvalid) {
                hv[pass] = reinterpret_cast<const float4*>(w.h + (size_t)(row0 + r) * H)[c4];
                av[pass] = reinterpret_cast<const float4*>(w.agg + (size_t)(row0 + r) * H)[c4];
            }
        }
#pragma unroll
        for (int pass = 0; pass < MT / 4; ++pass) {
            const int r = pass * 4 + rsub;
            if (r < nvalid) reinterpret_cast<float4*>(w.agg + (size_t)(row0 + r) * H)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);   // agg is zero between blocks
            float4 v = av[pass];
            const float dv = r < nvalid ? agg_div(w, d, row0 + r) : 1.0f;
            v.x /= dv; v.y /= dv; v.z /= dv; v.w /= dv;
            if (SAVE && r < nvalid) reinterpret_cast<float4*>(sv.aggn + ((size_t)layer * lay.N + row0 + r) * H)[c4] = v;
            *reinterpret_cast<float4*>(buf0 + r * LDA(H) + 4 * c4) = hv[pass];
            *reinterpret_cast<float4*>(buf1 + r * LDA(H) + 4 * c4) = v;
        }
        lds_barrier();
        NSTAMP(0);
        G::template gemm<H / 8>(buf0, LDA(H), f3a, f3b, acc, carry);                         // h part of [h | agg]
        G::template gemm<H / 8>(buf1, LDA(H), f3b, f4, acc, carry);                          // agg part
        NSTAMP(1);
    } else {
        load_h();
        lds_barrier();
        G::template gemm<H / 8>(buf0, LDA(H), f3a, f3b, acc, carry);
        lds_barrier();
        load_agg();
        lds_barrier();
        G::template gemm<H / 8>(buf1, LDA(H), f3b, f4, acc, carry);
    }
    lds_barrier();
    acc_foreach_n<MT>(acc, wave, [&](int row, int col, int n, float v) {
        const float pre = v + b3v.v[n];
        buf1[row * LDA(H) + col] = SAVE ? pre : silu_f(pre);
    });
    lds_barrier();
    if constexpr (SAVE) {
        const size_t o = ((size_t)layer * lay.N + row0) * H;
        save_rows_silu<H, MT>(buf1, nvalid, sv.pre3 + o, sv.nact + o);
        lds_barrier();
    }
    NSTAMP(2);
    acc_zero<MT>(acc);
    G::template gemm<H / 8>(buf1, LDA(H), f4, fc, acc, carry);
    NSTAMP(3);
    lds_barrier();
    acc_foreach_n<MT>(acc, wave, [&](int row, int col, int n, float v) {
        float hn = 0.f;
        if (row < nvalid) {
            float* hp = w.h + (size_t)(row0 + row) * H + col;
            const float hold = TWO ? buf0[row * LDA(H) + col] : *hp;
            hn = hold + (v + b4v.v[n]);                                                 // residual (egnn_new.py:57)
            if (MT != 32 && !SAVE) *hp = hn;    // 32-row tiles and the training forward store h from the LDS image below, as whole rows
        }
        buf1[row * LDA(H) + col] = hn;
    });
    lds_barrier();
    if constexpr (MT == 32 || SAVE) {       // h_new is in LDS for the projections anyway: it leaves as 1 KiB rows, 16 bytes per lane
                                            // (B=256: +0.7 %; at 16 rows the scalar stores are as good, gpurun_out/r2zw_h_rowstore_ab.txt)
#pragma unroll
        for (int pass = 0; pass < MT / 4; ++pass) {
            const int r = pass * 4 + rsub;
            if (r < nvalid) {
                const float4 hv = *reinterpret_cast<const float4*>(buf1 + r * LDA(H) + 4 * c4);
                reinterpret_cast<float4*>(w.h + (size_t)(row0 + r) * H)[c4] = hv;
                if (SAVE) reinterpret_cast<float4*>(sv.h + ((size_t)(layer + 1) * lay.N + row0 + r) * H)[c4] = hv;   // h entering block layer+1
            }
        }
    }
    NSTAMP(4);
    // coord MLP projections: P_c only where the tile holds phar rows (receivers that move)
    if (!skip_pc) tile_project_pq<H, MT, SP>(buf1, lw.Wpq_c, b6v, w.Pc, w.Qc, row0, nvalid, want_pc, carry, fn);
    NSTAMP(5);
    if (has_next) tile_project_pq<H, MT, SP>(buf1, lw_next.Wpq_e, b1nv, w.P, w.Q, row0, nvalid, true, carry, fn);
    NSTAMP(6);
#if CMDGEN_STAMPS == 2
    if ((tid & 63) == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&w.dbg[wave * 8 + i], nst_[i]);
        atomicAdd(&w.dbg[32 + wave], __builtin_amdgcn_s_memtime() - nst_begin);
        atomicAdd(&w.dbg[40], 1ull);
    }
#endif
#undef NSTAMP
}
template <int H, int MT, bool SAVE, bool SP>
__global__ __launch_bounds__(H, (MT == 16 && SP) ? 1 : 2) void k_node(Layout lay, Work w, Dims d, LayerW lw, LayerW lw_next,
                                               int layer, int has_next, TrainSave sv) {
    __shared__ __attribute__((aligned(16))) float bufs[(MT <= 32 ? 2 : 1) * MT * LDA(H)];
    node_tile_body<H, MT, SAVE, SP>(bufs, lay, w, d, lw, lw_next, layer, has_next, sv, (int)blockIdx.x * MT, lay.N);
}

// ------------------------------------------------------------------------------------
// edge_coord_body / k_edge_coord: EquivariantUpdate.coord_model on the edges whose receiver moves: phar nodes in
// conditional mode (pocket rows are multiplied by update_coords_mask = 0, egnn_new.py:100-101), every
// node in joint mode (update_coords_mask = None, dynamics.py:105-107):
//   phi = w5 . SiLU(W7 SiLU(W6 [h_i, h_j, r, d0] + b6) + b7)
//   ACC[l][i] += (x_i - x_j) / (sqrt(r + 1e-8) + norm_constant) * tanh(phi) * coords_range
// ------------------------------------------------------------------------------------
template <int H, int MT, bool SAVE, bool SP, int FK = 0>
__device__ __forceinline__ void edge_coord_body(EdgeLds<H, MT, FK>& L, const Layout& lay, const Work& w, const Dims& d, const LayerW& lw,
                                                const int layer, const TrainSave& sv) {
    float* buf = L.buf; int* s_row = L.s_row; int* s_col = L.s_col;
    float* s_r = L.s_r; float* s_d0 = L.s_d0; float* s_w5 = L.s_vec; float* s_wrd = L.s_wrd;
    float (*s_cd)[3] = L.s_cd; float (*s_tr)[3] = L.s_tr;
    const int tid = threadIdx.x, wave = tid >> 6;
    s_w5[tid] = lw.w5[tid];                                    // coord_mlp.4 weight, staged once per workgroup (see edge_msg_body)
    constexpr bool PL = SP && H == 256 && MT >= 32;            // plane variant, see edge_msg_body
    static_assert(!FK || (PL && MT == 32 && (!SAVE || FK == 2)), "full-K planes: 32-row tiles on the split engine (training forward: the half engine only)");
    if constexpr (PL && !FK) { s_wrd[tid] = lw.wr_c[tid]; s_wrd[H + tid] = lw.wd_c[tid]; }
    const ColVec<MT> b7v = col_load<MT>(lw.b7, wave);
    const float4 wr4 = reinterpret_cast<const float4*>(lw.wr_c)[tid % (H / 4)], wd4 = reinterpret_cast<const float4*>(lw.wd_c)[tid % (H / 4)];
    extern __shared__ float s_dyn[];          // sin_embedding only: see edge_msg_body
    if constexpr (!FK && !(SP && H == 256 && MT >= 32))
        if (d.sin) for (int i = tid; i < 24 * H; i += H) s_dyn[i] = lw.we_c[i];
    typedef typename EdgeEng<MT, SP, FK>::G G;
    float inv7 = 1.0f;                                         // (see edge_msg_body)
    if constexpr (FK != 0) inv7 = SAVE ? lw.W7.wh_dev[1] : G::inv(lw.W7);
    const typename G::Frag fw = G::frag(lw.W7, H / 8, 0, wave);
    typename G::Carry carry;
    G::prefetch(fw, carry);                                    // unconditional, see edge_msg_body
    const int E = w.totals[1];
    const int ntiles = (E + MT - 1) / MT;
    int nx_row = -1, nx_col = -1; float nx_d0 = 0.f;           // the next tile's triple, one tile ahead (see edge_msg_body)
    {
        const int t0 = xcd_tile(0, ntiles);
        if (t0 >= 0 && tid < MT && t0 * MT + tid < E) { nx_row = w.crow[t0 * MT + tid]; nx_col = w.ccol[t0 * MT + tid]; nx_d0 = w.cd0[t0 * MT + tid]; }
    }
    for (int k = 0, tile; (tile = xcd_tile(k, ntiles)) >= 0; ++k) {
        const int e0 = tile * MT;
        const int ne = min(MT, E - e0);
        if (tid < MT) {
            const int row = nx_row, col = nx_col; const float d0 = nx_d0;               // phar receivers, self loops dropped
            nx_row = -1; nx_col = -1; nx_d0 = 0.f;
            const int tn = xcd_tile(k + 1, ntiles);
            if (tn >= 0 && tn * MT + tid < E) { nx_row = w.crow[tn * MT + tid]; nx_col = w.ccol[tn * MT + tid]; nx_d0 = w.cd0[tn * MT + tid]; }
            float r = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
            if (tid < ne) {
                const float4 pi = node_pos(lay, w, d, row, layer, false);
                const float4 pj = node_pos(lay, w, d, col, layer, false);
                cx = pi.x - pj.x; cy = pi.y - pj.y; cz = pi.z - pj.z;
                r = cx * cx + cy * cy + cz * cz;
                const float den = sqrtf(r + 1e-8f) + d.norm_constant;      // coord2diff, egnn_new.py:265-271
                cx /= den; cy /= den; cz /= den;
                if constexpr (!FK && !(SP && H == 256 && MT >= 32))
                    if (d.sin) r = sumsq_ref(pi, pj);                        // (the features' distance in the reference's rounding)
            }
            float d0f = d0;
            if constexpr (!FK && !(SP && H == 256 && MT >= 32))
                if (d.sin && tid < ne) d0f = sumsq_ref(node_pos(lay, w, d, row, 0, false), node_pos(lay, w, d, col, 0, false));
            s_row[tid] = row; s_col[tid] = col; s_r[tid] = r; s_d0[tid] = d0f;
            s_cd[tid][0] = cx; s_cd[tid][1] = cy; s_cd[tid][2] = cz;
        }
        lds_barrier();
        TileAcc<MT> acc;
        acc_zero<MT>(acc);
        if constexpr (FK) {
            unsigned short* planes = reinterpret_cast<unsigned short*>(buf);
            build_edge_full32<FK>(planes, s_row, s_col, s_r, s_d0, ne, w.Pc, w.Qc, wr4, wd4,
                                  SAVE ? sv.pre6 + ((size_t)layer * sv.eccap + e0) * H : nullptr,
                                  SAVE ? sv.act6 + ((size_t)layer * sv.eccap + e0) * H : nullptr);
            lds_barrier();
            G::gemm(planes, fw, acc.a, carry);
        } else if constexpr (PL) {
            unsigned short* planes = reinterpret_cast<unsigned short*>(buf);
            constexpr int PLDA = SPLIT_PLANE_LDA(H / 2), PE = MT * PLDA;
            const typename G::Frag fw1 = G::frag(lw.W7, H / 8, H / 16, wave);
            float* pre6_o = SAVE ? sv.pre6 + ((size_t)layer * sv.eccap + e0) * H : nullptr;
            float* act6_o = SAVE ? sv.act6 + ((size_t)layer * sv.eccap + e0) * H : nullptr;
            build_edge_half<MT>(planes, 0, s_row, s_col, s_r, s_d0, ne, w.Pc, w.Qc, s_wrd, s_wrd + H, pre6_o, act6_o);
            lds_barrier();
            tile_gemm_planes<MT, H / 32>(planes, PE, PLDA, fw, fw1, acc.a, carry);
            lds_barrier();
            build_edge_half<MT>(planes, 1, s_row, s_col, s_r, s_d0, ne, w.Pc, w.Qc, s_wrd, s_wrd + H, pre6_o, act6_o);
            lds_barrier();
            tile_gemm_planes<MT, H / 32>(planes, PE, PLDA, fw1, fw, acc.a, carry);
        } else {
        if (d.sin) { sin_features<H, MT>(s_dyn + 24 * H, s_r, s_d0, ne, d); lds_barrier(); }
        build_edge_tile<H, MT>(buf, s_row, s_col, s_r, s_d0, ne, w.Pc, w.Qc, wr4, wd4,
                               SAVE ? sv.pre6 + ((size_t)layer * sv.eccap + e0) * H : nullptr,
                               SAVE ? sv.act6 + ((size_t)layer * sv.eccap + e0) * H : nullptr,
                               d.sin ? s_dyn + 24 * H : nullptr, s_dyn);
        lds_barrier();
        G::template gemm<H / 8>(buf, LDA(H), fw, fw, acc, carry);
        }
        lds_barrier();
        acc_foreach_n<MT>(acc, wave, [&](int row, int col, int n, float v) {
            const float pre = __fmaf_rn(v, inv7, b7v.v[n]);
            buf[row * LDA(H) + col] = SAVE ? pre : silu_f(pre);
        });
        lds_barrier();
        if constexpr (SAVE) {
            const size_t o = ((size_t)layer * sv.eccap + e0) * H;
            save_rows_silu<H, MT>(buf, ne, sv.pre7 + o, sv.act7 ? sv.act7 + o : nullptr);
            lds_barrier();
        }
        {
            int r; bool lead;
            const float s = tile_row_dot<H, MT>(buf, s_w5, r, lead);
            if (lead) {
                if (SAVE && r < ne) sv.phi[(size_t)layer * sv.eccap + e0 + r] = s;
                const float g = d.use_tanh ? tanhf(s) * d.coords_range : s;
                s_tr[r][0] = s_cd[r][0] * g; s_tr[r][1] = s_cd[r][1] * g; s_tr[r][2] = s_cd[r][2] * g;
            }
        }
        lds_barrier();
        // Ordered segment sums of the three components: the list is sorted by receiver, so one thread per (row that begins a receiver's run,
        // component) adds the run in list order.  A run inside the tile is complete: plain store (ACC is zero before the launch); a run that
        // touches the tile's first or last row may continue in a neighbouring tile: one float atomic (two per receiver at most while a
        // receiver's edges span two tiles: commutative, so the result does not depend on the order of the workgroups).
        for (int i = tid; i < 3 * MT; i += H) {             // (H threads per workgroup)
            const int e = i / 3, comp = i - 3 * e;
            if (e < ne && (e == 0 || s_row[e] != s_row[e - 1])) {
                const int rr = s_row[e];
                float sum = 0.f;
                int q = e;
                for (; q < ne && s_row[q] == rr; ++q) sum += s_tr[q][comp];
                float* dst = reinterpret_cast<float*>(w.ACC + (size_t)layer * lay.Nm + rr) + comp;
                if (e == 0 || q == ne) atomicAdd(dst, sum); else *dst = sum;
            }
        }
        lds_barrier();
    }
}
template <int H, int MT, bool SAVE, bool SP, int FK = 0>
__global__ __launch_bounds__(H, FK ? 3 : 2) void k_edge_coord(Layout lay, Work w, Dims d, LayerW lw, int layer, TrainSave sv) {
    __shared__ __attribute__((aligned(16))) EdgeLds<H, MT, FK> L;
    edge_coord_body<H, MT, SAVE, SP, FK>(L, lay, w, d, lw, layer, sv);
}

// ------------------------------------------------------------------------------------
// k_readout: embedding_out (drop the time column), decoders, velocity, NaN flag
// (egnn_new.py:205, dynamics.py:110-131).  8 nodes per workgroup, 32 threads per node; the
// node's h row is staged in LDS and the transposed weight is read coalesced.
// eps rows: [vel(3) | decoded features].  The batch-global NaN reset is applied by the
// consumer (k_nan_fix or the DDPM kernels) once the flag is complete.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_readout(Layout lay, Work w, Dims d, SmallW sw,
                                                 float* __restrict__ eps_phar, float* __restrict__ eps_pocket,
                                                 ChainState* chain, TrainSave sv) {
    extern __shared__ float s_hrow[];            // [8][H] node rows, then [H][dyn] embedding_out^T
    __shared__ float s_j[8][CMDGEN_MAX_SMALL + 1];
    __shared__ float s_h1[8][CMDGEN_MAX_SMALL];
    const int tid = threadIdx.x, g = tid >> 5, l32 = tid & 31;
    if (chain && blockIdx.x == 0 && tid == 0) chain->step += 1;      // this evaluation is done (see ChainState)
    const int nnodes = eps_pocket ? lay.N : lay.Nl;
    const int n = blockIdx.x * 8 + g;
    const bool live = n < nnodes;
    const bool ph = n < lay.Nl;
    const int H = d.H;
    float* s_wT = s_hrow + 8 * H;
    for (int i = tid; i < H * d.dyn; i += 256) s_wT[i] = sw.embo_wT[i];        // coalesced, all loads in flight
    if (live) for (int k = l32; k < H; k += 32) s_hrow[g * H + k] = w.h[(size_t)n * H + k];
    __syncthreads();
    if (live) {
        for (int j = l32; j < d.J; j += 32) {
            float s0 = sw.embo_b[j], s1 = 0.f, s2 = 0.f, s3 = 0.f;
            const float* hr = s_hrow + g * H;
            const float* wt = s_wT + j;
            for (int k = 0; k < H; k += 4) {
                s0 = fmaf(hr[k], wt[k * d.dyn], s0);
                s1 = fmaf(hr[k + 1], wt[(k + 1) * d.dyn], s1);
                s2 = fmaf(hr[k + 2], wt[(k + 2) * d.dyn], s2);
                s3 = fmaf(hr[k + 3], wt[(k + 3) * d.dyn], s3);
            }
            s_j[g][j] = (s0 + s1) + (s2 + s3);
            if (sv.hfin) sv.hfin[(size_t)n * d.dyn + j] = s_j[g][j];
        }
    }
    __syncthreads();
    if (live) {
        const int F = ph ? d.P : d.R;
        const float* W0 = ph ? sw.pd0_w : sw.rd0_w; const float* B0 = ph ? sw.pd0_b : sw.rd0_b;
        for (int o = l32; o < 2 * F; o += 32) {
            float s = B0[o];
            for (int k = 0; k < d.J; ++k) s = fmaf(s_j[g][k], W0[(size_t)o * d.J + k], s);
            const float a = silu_f(s);
            s_h1[g][o] = a;
            if (sv.dec1) {
                if (ph) { sv.dec1[(size_t)n * 2 * F + o] = s; sv.deca[(size_t)n * 2 * F + o] = a; }
                else if (sv.qdec1) { sv.qdec1[(size_t)(n - lay.Nl) * 2 * F + o] = s; sv.qdeca[(size_t)(n - lay.Nl) * 2 * F + o] = a; }
            }
        }
    }
    __syncthreads();
    if (live) {
        const int F = ph ? d.P : d.R;
        const float* W2 = ph ? sw.pd2_w : sw.rd2_w; const float* B2 = ph ? sw.pd2_b : sw.rd2_b;
        float* out = ph ? eps_phar + (size_t)n * (3 + d.P) : eps_pocket + (size_t)(n - lay.Nl) * (3 + d.R);
        for (int o = l32; o < F; o += 32) {
            float s = B2[o];
            for (int k = 0; k < 2 * F; ++k) s = fmaf(s_h1[g][k], W2[(size_t)o * 2 * F + k], s);
            out[3 + o] = s;
            if (sv.dec_out) {
                if (ph) sv.dec_out[(size_t)n * F + o] = s;
                else if (sv.qdec_out) sv.qdec_out[(size_t)(n - lay.Nl) * F + o] = s;
            }
        }
        if (l32 == 0) {
            float vx = 0.f, vy = 0.f, vz = 0.f;
            if (n < lay.Nm) {
                // x_final = X[L-1] + ACC[L-1]/nf ; vel = x_final - x_input
                const float4 p = (d.L == 1) ? w.X0[n] : w.XL[(size_t)(d.L - 1) * lay.Nm + n];
                const float4 a = w.ACC[(size_t)(d.L - 1) * lay.Nm + n];
                const float4 x0 = w.X0[n];
                const float dv = agg_div(w, d, n);
                vx = (p.x + a.x / dv) - x0.x;
                vy = (p.y + a.y / dv) - x0.y;
                vz = (p.z + a.z / dv) - x0.z;
                if (isnan(vx) || isnan(vy) || isnan(vz)) atomicOr(w.nan_flag, 1);
            }
            out[0] = vx; out[1] = vy; out[2] = vz;
        }
    }
}

// applies the reference's batch-global NaN reset to an evaluation's output (dynamics.py:129-131)
__global__ void k_nan_fix(Layout lay, Work w, Dims d, float* __restrict__ eps_phar) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0 && *w.nan_flag) atomicAdd(&w.counters[4], 1ull);
    if (n < lay.Nl && *w.nan_flag) {
        float* o = eps_phar + (size_t)n * (3 + d.P);
        o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
    }
}

// Joint mode tail of EGNNDynamics.forward (dynamics.py:129-136): the batch-global NaN reset, then
// remove_mean_batch(vel, mask) over ALL nodes of each sample.  One wave per sample; the sum runs in
// index order (phar rows, then pocket rows) like the reference's scatter_add.
__global__ __launch_bounds__(64) void k_vel_com(Layout lay, Work w, Dims d, float* __restrict__ eps_phar,
                                                float* __restrict__ eps_pocket) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nl = lay.num_phar[b], np = lay.num_pocket[b];
    const int ldp = 3 + d.P, ldq = 3 + d.R;
    float* vp = eps_phar + (size_t)lay.phar_base[b] * ldp;
    float* vq = eps_pocket + (size_t)lay.pocket_base[b] * ldq;
    const bool nan_reset = *w.nan_flag != 0;
    if (b == 0 && lane == 0 && nan_reset) atomicAdd(&w.counters[4], 1ull);
    if (nan_reset) {
        for (int i = lane; i < nl; i += 64) { vp[i * ldp] = 0.f; vp[i * ldp + 1] = 0.f; vp[i * ldp + 2] = 0.f; }
        for (int i = lane; i < np; i += 64) { vq[i * ldq] = 0.f; vq[i * ldq + 1] = 0.f; vq[i * ldq + 2] = 0.f; }
    }
    __syncthreads();
    float mean = 0.f;
    if (lane < 3) {
        float s = 0.f;
        for (int i = 0; i < nl; ++i) s += vp[i * ldp + lane];
        for (int i = 0; i < np; ++i) s += vq[i * ldq + lane];
        mean = s / fmaxf((float)(nl + np), 1.0f);
    }
    const float m0 = __shfl(mean, 0), m1 = __shfl(mean, 1), m2 = __shfl(mean, 2);
    for (int i = lane; i < nl; i += 64) { vp[i * ldp] -= m0; vp[i * ldp + 1] -= m1; vp[i * ldp + 2] -= m2; }
    for (int i = lane; i < np; i += 64) { vq[i * ldq] -= m0; vq[i * ldq + 1] -= m1; vq[i * ldq + 2] -= m2; }
}

// ------------------------------------------------------------------------------------
// host-callable launchers (C++ linkage, used by cmdgen_api.hip)
// ------------------------------------------------------------------------------------
template <int H, int MT, bool SP> static void launch_embed(const EvalLaunch& a, const float* xp, const float* xq, const float* t,
                                                  const float4* coef, ChainState* chain, hipStream_t s) {
    if constexpr (H == 512 && MT == 64) launch_embed<H, 32, SP>(a, xp, xq, t, coef, chain, s);      // (its 64-row tile would need 181 KB of LDS)
    else {
        const int nt = (a.lay.N + MT - 1) / MT;
        const Dims& d = a.d;
        const size_t shm = sizeof(float) * (size_t)(2 * d.P * d.P + 2 * d.P + d.J * 2 * d.P + d.J + 2 * d.R * d.R + 2 * d.R + d.J * 2 * d.R + d.J);
        if (a.save) hipLaunchKernelGGL((k_embed<H, MT, false>), dim3(nt), dim3(H), shm, s, a.lay, a.w, a.d, a.sw, a.layers[0], xp, xq, t, coef,
                                       (const ChainState*)chain, *a.save, PocketCache{});          // training packs: fp32 fragments only
        else hipLaunchKernelGGL((k_embed<H, MT, SP>), dim3(nt), dim3(H), shm, s, a.lay, a.w, a.d, a.sw, a.layers[0], xp, xq, t, coef,
                                (const ChainState*)chain, TrainSave{}, (chain && !t) ? a.pcache : PocketCache{});
    }
}
template <int H, int MT, bool SP> static void launch_write_embed(const EvalLaunch& a, const float* xp, const float* xq, const float* t,
                                                                 const float4* coef, ChainState* chain, hipStream_t s) {
    if constexpr (H == 256) {
        const int nt = (a.lay.N + MT - 1) / MT;
        const Dims& d = a.d;
        const size_t shm_e = sizeof(float) * (size_t)(2 * d.P * d.P + 2 * d.P + d.J * 2 * d.P + d.J + 2 * d.R * d.R + 2 * d.R + d.J * 2 * d.R + d.J);
        const size_t shm_w = (size_t)a.lay.max_n * (sizeof(float4) + 3 * sizeof(int));
        const PocketCache pc = (chain && !t) ? a.pcache : PocketCache{};
        const int npair = pc.c ? (a.lay.Nl + MT - 1) / MT : 0;        // pairs only where the other tiles are cache tiles
        hipLaunchKernelGGL((k_write_embed<MT, SP>), dim3(a.lay.B + nt + npair), dim3(256), shm_e > shm_w ? shm_e : shm_w, s, a.lay, a.w, a.d, a.sw,
                           a.layers[0], xp, xq, t, coef, (const ChainState*)chain, pc, npair);
    }
}
// SAVE variants (training forward) keep the activations; the sampler's instantiations carry no trace of the stores
template <int H, int MT, bool SP> static void launch_node(const EvalLaunch& a, int l, hipStream_t s) {
    if constexpr (MT == 16 && !SP && H >= 128) {
        // 16-row tiles on the split engine (v_mfma_f32_16x16x32_bf16): opt-in, see DESIGN section 4a for why it is not the default
        if (a.split16 && !a.save && a.layers[unit_of(a, l)].W3.ws16) { launch_node<H, 16, true>(a, l, s); return; }
    }
    const int nt = (a.lay.N + MT - 1) / MT;
    if (a.save) hipLaunchKernelGGL((k_node<H, MT, true, false>), dim3(nt), dim3(H), 0, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)],
                                   a.layers[unit_has_next(a, l) ? unit_of(a, l) + 1 : unit_of(a, l)], l, node_flags(a, l), *a.save);
    else if (a.pe_start) hipExtLaunchKernelGGL((k_node<H, MT, false, SP>), dim3(nt), dim3(H), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d,
                                               a.layers[unit_of(a, l)], a.layers[unit_has_next(a, l) ? unit_of(a, l) + 1 : unit_of(a, l)], l, node_flags(a, l), TrainSave{});
    else hipLaunchKernelGGL((k_node<H, MT, false, SP>), dim3(nt), dim3(H), 0, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)],
                            a.layers[unit_has_next(a, l) ? unit_of(a, l) + 1 : unit_of(a, l)], l, node_flags(a, l), TrainSave{});
}
template <int H, int MT, bool SP> static void launch_msg(const EvalLaunch& a, int l, hipStream_t s) {
    const size_t shm = a.d.sin ? (size_t)(24 * H + 24 * MT) * sizeof(float) : 0;      // sin_embedding: feature columns + the tile's features (edge_msg_body)
    // training forward: the split engine only where the step re-packs split weights (H = 256: edge_mlp.2 / coord_mlp.2)
    if (a.save) hipLaunchKernelGGL((k_edge_msg<H, MT, true, SP && H == 256>), dim3(a.edge_grid), dim3(H), 0, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)], l, a.ablate, *a.save, 0);
    else if (a.pe_start) hipExtLaunchKernelGGL((k_edge_msg<H, MT, false, SP>), dim3(a.edge_grid), dim3(H), shm, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d,
                                               a.layers[unit_of(a, l)], l, a.ablate, TrainSave{}, a.live_thr);
    else hipLaunchKernelGGL((k_edge_msg<H, MT, false, SP>), dim3(a.edge_grid), dim3(H), shm, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)], l, a.ablate, TrainSave{}, a.live_thr);
}
// 32-row sampler tiles on the split engine: full-K planes (one build, one GEMM per tile; see cmdgen_split.h) unless CMDGEN_EDGE_FULLK=0
static bool launch_msg_fullk(const EvalLaunch& a, int l, hipStream_t s) {
    if (!a.edge_fullk || (a.save && !a.save_half) || !a.split || a.d.H != 256 || a.edge_mt != 32) return false;
    const LayerW& lw = a.layers[unit_of(a, l)];
    if (a.save) {       // training forward on the half engine (packs and scale re-made on the device every step: WPack::wh_dev)
        hipLaunchKernelGGL((k_edge_msg<256, 32, true, true, 2>), dim3(a.edge_grid), dim3(256), 0, s, a.lay, a.w, a.d, lw, l, a.ablate, *a.save, 0);
        return true;
    }
    if (a.half_engine && lw.W2.wh) {
        if (a.pe_start) hipExtLaunchKernelGGL((k_edge_msg<256, 32, false, true, 2>), dim3(a.edge_grid), dim3(256), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, lw, l, a.ablate, TrainSave{}, a.live_thr);
        else hipLaunchKernelGGL((k_edge_msg<256, 32, false, true, 2>), dim3(a.edge_grid), dim3(256), 0, s, a.lay, a.w, a.d, lw, l, a.ablate, TrainSave{}, a.live_thr);
    } else {
        if (a.pe_start) hipExtLaunchKernelGGL((k_edge_msg<256, 32, false, true, 3>), dim3(a.edge_grid), dim3(256), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, lw, l, a.ablate, TrainSave{}, a.live_thr);
        else hipLaunchKernelGGL((k_edge_msg<256, 32, false, true, 3>), dim3(a.edge_grid), dim3(256), 0, s, a.lay, a.w, a.d, lw, l, a.ablate, TrainSave{}, a.live_thr);
    }
    return true;
}
static bool launch_coord_fullk(const EvalLaunch& a, int l, hipStream_t s) {
    if (!a.edge_fullk || (a.save && !a.save_half) || !a.split || a.d.H != 256 || a.coord_mt != 32) return false;
    const LayerW& lw = a.layers[unit_of(a, l)];
    if (a.save) {
        hipLaunchKernelGGL((k_edge_coord<256, 32, true, true, 2>), dim3(a.coord_grid), dim3(256), 0, s, a.lay, a.w, a.d, lw, l, *a.save);
        return true;
    }
    if (a.half_engine && lw.W7.wh) {
        if (a.pe_start) hipExtLaunchKernelGGL((k_edge_coord<256, 32, false, true, 2>), dim3(a.coord_grid), dim3(256), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, lw, l, TrainSave{});
        else hipLaunchKernelGGL((k_edge_coord<256, 32, false, true, 2>), dim3(a.coord_grid), dim3(256), 0, s, a.lay, a.w, a.d, lw, l, TrainSave{});
    } else {
        if (a.pe_start) hipExtLaunchKernelGGL((k_edge_coord<256, 32, false, true, 3>), dim3(a.coord_grid), dim3(256), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, lw, l, TrainSave{});
        else hipLaunchKernelGGL((k_edge_coord<256, 32, false, true, 3>), dim3(a.coord_grid), dim3(256), 0, s, a.lay, a.w, a.d, lw, l, TrainSave{});
    }
    return true;
}
template <int H, int MT, bool SP> static void launch_coord(const EvalLaunch& a, int l, hipStream_t s) {
    const size_t shm = a.d.sin ? (size_t)(24 * H + 24 * MT) * sizeof(float) : 0;
    if (a.save) hipLaunchKernelGGL((k_edge_coord<H, MT, true, SP && H == 256>), dim3(a.coord_grid), dim3(H), 0, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)], l, *a.save);
    else if (a.pe_start) hipExtLaunchKernelGGL((k_edge_coord<H, MT, false, SP>), dim3(a.coord_grid), dim3(H), shm, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d,
                                               a.layers[unit_of(a, l)], l, TrainSave{});
    else hipLaunchKernelGGL((k_edge_coord<H, MT, false, SP>), dim3(a.coord_grid), dim3(H), shm, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)], l, TrainSave{});
}
// tiles of >= 32 rows run on the split-bf16 engine when the launch asks for it (the training forward: only its two edge
// kernels, and only when the step re-packed split weights for them - save_split); 16-row tiles are always fp32 MFMA
// (there the L2 weight stream, not the matrix rate, binds)
#define MT_DISPATCH(mt, FN, ...) do { const bool sp_ = a.split && (!a.save || a.save_split);                                                 \
        if ((mt) >= 64) { if (sp_) FN<H, 64, true>(__VA_ARGS__); else FN<H, 64, false>(__VA_ARGS__); }                       \
        else if ((mt) == 32) { if (sp_) FN<H, 32, true>(__VA_ARGS__); else FN<H, 32, false>(__VA_ARGS__); }                 \
        else FN<H, 16, false>(__VA_ARGS__); } while (0)

template <int H>
static void launch_eval_H(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket,
                          const float* t_arr, const float4* coef, ChainState* chain,
                          float* eps_phar, float* eps_pocket, hipStream_t s,
                          hipEvent_t* ev /* null or 2*(3+3L) events */) {
    const int B = a.lay.B, N = a.lay.N;
    const size_t shm = (size_t)a.lay.max_n * (sizeof(float4) + 3 * sizeof(int));
    int e = 0;
#define REC() do { if (ev) hipEventRecord(ev[e++], s); } while (0)
    // kernel profiling: the launch itself carries a start and a stop event (hipExtLaunchKernelGGL: the timestamps of the dispatch
    // packet, i.e. what rocprofv3 reports), not events recorded around it on the stream (those include the launch gap)
#define PROF_BEGIN(k) do { if (a.prof_events && !a.save) { hipEventCreate(&a.pe_start); hipEventCreate(&a.pe_stop); \
                           a.prof_events[k].push_back(a.pe_start); a.prof_events[k].push_back(a.pe_stop); } } while (0)
#define PROF_END() do { a.pe_start = nullptr; a.pe_stop = nullptr; } while (0)
    // k_embed's tile: the small one only where the pocket cache serves the pocket rows (inside a conditional chain)
    const int emt = (chain && !t_arr && a.pcache.c) ? a.embed_mt : a.node_mt;
    REC();
    // per-sample graph kernels: one wave scans one receiver at a time, so big samples (full-atom pockets: 381 nodes) get 16 waves
    const int gthr = a.lay.max_n > 128 ? 1024 : 256;
    if (!a.skip_count) hipLaunchKernelGGL(k_edge_count, dim3(B), dim3(gthr), shm, s, a.lay, a.w, a.d, xh_phar, xh_pocket);
    if (a.skip_count == 2) {          // training forward: the graph was built (and its size read back) before the activation store was sized
        REC(); REC();
        MT_DISPATCH(emt, launch_embed, a, xh_phar, xh_pocket, t_arr, coef, chain, s);
    } else if (H == 256 && !ev && !a.save && gthr == 256 && shm <= 64 * 1024 && (size_t)emt * 1812 + 1024 + (shm > 12288 ? shm : 12288) <= 160 * 1024 &&
               a.write_embed) {      // (static LDS of the embedding body is 1812 B per tile row; one launch must hold both bodies' LDS)
        MT_DISPATCH(emt, launch_write_embed, a, xh_phar, xh_pocket, t_arr, coef, chain, s);       // both in one launch
    } else {
        hipLaunchKernelGGL(k_edge_write, dim3(B), dim3(gthr), shm, s, a.lay, a.w, a.d);
        REC(); REC();
        MT_DISPATCH(emt, launch_embed, a, xh_phar, xh_pocket, t_arr, coef, chain, s);
    }
    REC();
    // dead work (conditional sampler, pocket output not asked for): see edge_msg_body / cmdgen_node_planes.h
    const bool live_last = a.dead_skip && !eps_pocket && !a.save && !a.d.joint && a.stop_block < 0 && a.w.need_qc != nullptr;
    for (int l = 0; l < a.d.L; ++l) {
        a.live_thr = live_last ? (a.dead_skip >= 2 ? a.d.L - l : (l == a.d.L - 1 ? 1 : 0)) : 0;     // dead_skip 1: the last block only; 2: every block (a kernel argument of its own: any n_layers)
        const int stop = a.stop_block == l ? a.stop_stage : 0;        // parity aid: leave intermediates in the workspace
        // the block's GCLs (inv_sublayers, egnn_new.py:152-154): message + node kernel per unit; only the last one projects P_c | Q_c.
        // (The per-stage events and the prefix stops belong to the block's last unit; cmdgen_profile_evaluation asks for S = 1.)
        for (int sub = 0; sub < a.d.S; ++sub) {
            const bool last = sub == a.d.S - 1;
            a.unit = l * a.d.S + sub; a.skip_pc = last ? 0 : 1;
            if (last) REC();
            PROF_BEGIN(0);
            if (!cmdgen_launch_msg128(a, l, s) && !launch_msg_fullk(a, l, s)) MT_DISPATCH(a.edge_mt, launch_msg, a, l, s);
            PROF_END();
            if (last) { REC(); REC(); }
            if (last && stop == 1) { a.unit = -1; a.skip_pc = 0; return; }
            PROF_BEGIN(1);
            if (!(a.node64 && cmdgen_launch_node64(a, l, s)) && !cmdgen_launch_node16w(a, l, s)) MT_DISPATCH(a.node_mt, launch_node, a, l, s);
            PROF_END();
            if (last) { REC(); REC(); }
            if (last && stop == 2) { a.unit = -1; a.skip_pc = 0; return; }
        }
        PROF_BEGIN(2);
        if (!cmdgen_launch_coord128(a, l, s) && !launch_coord_fullk(a, l, s)) MT_DISPATCH(a.coord_mt, launch_coord, a, l, s);
        PROF_END();
        REC();
        a.unit = -1; a.skip_pc = 0;
        if (stop == 3) return;
    }
    REC();
    const int nn = eps_pocket ? N : a.lay.Nl;
    hipLaunchKernelGGL(k_readout, dim3((nn + 7) / 8), dim3(256), (8 + a.d.dyn) * a.d.H * sizeof(float), s, a.lay, a.w, a.d, a.sw,
                       eps_phar, eps_pocket, chain, a.save ? *a.save : TrainSave{});
    if (a.d.joint) hipLaunchKernelGGL(k_vel_com, dim3(B), dim3(64), 0, s, a.lay, a.w, a.d, eps_phar, eps_pocket);
    REC();
#undef REC
#undef PROF_BEGIN
#undef PROF_END
}

void cmdgen_launch_eval(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket,
                        const float* t_arr, const float4* coef, ChainState* chain,
                        float* eps_phar, float* eps_pocket, hipStream_t s, hipEvent_t* ev) {
    switch (a.d.H) {
        case 512: launch_eval_H<512>(a, xh_phar, xh_pocket, t_arr, coef, chain, eps_phar, eps_pocket, s, ev); break;
        case 256: launch_eval_H<256>(a, xh_phar, xh_pocket, t_arr, coef, chain, eps_phar, eps_pocket, s, ev); break;
        case 128: launch_eval_H<128>(a, xh_phar, xh_pocket, t_arr, coef, chain, eps_phar, eps_pocket, s, ev); break;
        case 64:  launch_eval_H<64>(a, xh_phar, xh_pocket, t_arr, coef, chain, eps_phar, eps_pocket, s, ev); break;
        default: break;   // rejected in cmdgen_create
    }
}

// Chain-start cache of k_embed's pocket rows: stage 0 copies the rows of an evaluation at t = 0, stage 1 turns row 0 of
// an evaluation at t = 1 into the three difference vectors (identical for every pocket row: the time column of the embedding
// and its image under the first edge-MLP layer).
__global__ void k_pocket_cache(Layout lay, Work w, int H, float* __restrict__ c, float* __restrict__ P0, float* __restrict__ Q0,
                               float* __restrict__ dh, float* __restrict__ dP, float* __restrict__ dQ, int stage) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t base = (size_t)lay.Nl * H;
    if (stage == 0) {
        if (i < (size_t)lay.Np * H) { c[i] = w.h[base + i]; P0[i] = w.P[base + i]; Q0[i] = w.Q[base + i]; }
    } else if (i < (size_t)H) {
        dh[i] = w.h[base + i] - c[i]; dP[i] = w.P[base + i] - P0[i]; dQ[i] = w.Q[base + i] - Q0[i];
    }
}
template <int H> static void embed_only_H(const EvalLaunch& a, const float* xp, const float* xq, const float* t, hipStream_t s) {
    MT_DISPATCH(a.node_mt, launch_embed, a, xp, xq, t, nullptr, nullptr, s);
}
// builds the cache from two embed-only passes with the time feature pinned to 0 and to 1 (t01: device [2][B])
void cmdgen_build_pocket_cache(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket, const float* t01,
                               float* c, float* P0, float* Q0, float* dh, float* dP, float* dQ, hipStream_t s) {
    if (a.lay.Np == 0) return;
    const int H = a.d.H;
    for (int stage = 0; stage < 2; ++stage) {
        const float* t = t01 + (size_t)stage * a.lay.B;
        switch (H) {
            case 512: embed_only_H<512>(a, xh_phar, xh_pocket, t, s); break;
            case 256: embed_only_H<256>(a, xh_phar, xh_pocket, t, s); break;
            case 128: embed_only_H<128>(a, xh_phar, xh_pocket, t, s); break;
            case 64:  embed_only_H<64>(a, xh_phar, xh_pocket, t, s); break;
            default: break;
        }
        const size_t n = stage == 0 ? (size_t)a.lay.Np * H : (size_t)H;
        hipLaunchKernelGGL(k_pocket_cache, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.lay, a.w, H, c, P0, Q0, dh, dP, dQ, stage);
    }
}

// positions entering every block (and after the last) for ALL nodes, as the backward pass indexes them: X[l][n], l = 0..L
__global__ void k_save_positions(Layout lay, Work w, Dims d, float4* __restrict__ X) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= lay.N) return;
    for (int l = 0; l <= d.L; ++l) {
        float4 p;
        if (n >= lay.Nm) p = w.XP[n - lay.Nl];
        else if (l == 0) p = w.X0[n];
        else if (l < d.L) p = w.XL[(size_t)l * lay.Nm + n];
        else {
            const float4 q = (d.L == 1) ? w.X0[n] : w.XL[(size_t)(d.L - 1) * lay.Nm + n];
            const float4 a = w.ACC[(size_t)(d.L - 1) * lay.Nm + n];
            const float dv = agg_div(w, d, n);
            p = make_float4(q.x + a.x / dv, q.y + a.y / dv, q.z + a.z / dv, 0.f);
        }
        X[(size_t)l * lay.N + n] = p;
    }
}
void cmdgen_launch_save_positions(const EvalLaunch& a, float4* X, hipStream_t s) {
    hipLaunchKernelGGL(k_save_positions, dim3((a.lay.N + 255) / 256), dim3(256), 0, s, a.lay, a.w, a.d, X);
}

// radius graph only (the training path builds its own evaluation on top of the same compact lists)
void cmdgen_launch_edges(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket, hipStream_t s) {
    const size_t shm = (size_t)a.lay.max_n * (sizeof(float4) + 3 * sizeof(int));
    const int gthr = a.lay.max_n > 128 ? 1024 : 256;
    hipLaunchKernelGGL(k_edge_count, dim3(a.lay.B), dim3(gthr), shm, s, a.lay, a.w, a.d, xh_phar, xh_pocket);
    hipLaunchKernelGGL(k_edge_write, dim3(a.lay.B), dim3(gthr), shm, s, a.lay, a.w, a.d);
}

// dynamic LDS above the 64 KiB default needs an explicit opt-in per kernel (samples of more than ~2700 nodes)
void cmdgen_edge_kernels_allow_lds(size_t bytes) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_edge_count), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_edge_write), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// k_readout stages 8 node rows + embedding_out^T in dynamic LDS: above the 64 KiB default (hidden_nf 512) the kernel needs the opt-in
void cmdgen_readout_allow_lds(size_t bytes) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_readout), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

void cmdgen_launch_nan_fix(const EvalLaunch& a, float* eps_phar, hipStream_t s) {
    hipLaunchKernelGGL(k_nan_fix, dim3((a.lay.Nl + 255) / 256), dim3(256), 0, s, a.lay, a.w, a.d, eps_phar);
}

// (the kernel the evaluation itself would run for this block: the 128-row kernel, then the full-K 32-row tiles, then the generic dispatch;
// weight unit of the block's first GCL when a block has several)
template <int H> static void launch_msg_only_H(const EvalLaunch& a, int layer, hipStream_t s) {
    a.unit = layer * a.d.S;
    if (!cmdgen_launch_msg128(a, layer, s) && !launch_msg_fullk(a, layer, s)) MT_DISPATCH(a.edge_mt, launch_msg, a, layer, s);
    a.unit = -1;
}
void cmdgen_launch_edge_msg_only(const EvalLaunch& a, int layer, hipStream_t s) {
    switch (a.d.H) {
        case 512: launch_msg_only_H<512>(a, layer, s); break;
        case 256: launch_msg_only_H<256>(a, layer, s); break;
        case 128: launch_msg_only_H<128>(a, layer, s); break;
        case 64:  launch_msg_only_H<64>(a, layer, s); break;
        default: break;
    }
}
