// kernels_egnn_graph.hip - the radius graph (k_edge_count / k_edge_write, dynamics.py:141-147) and k_embed / k_write_embed (encoders, time
// column, embedding, P | Q of block 0) with the chain-start pocket cache.  Shared helpers: cmdgen_egnn_common.h.
// Build time: this file is compiled TWICE - as itself (CMDGEN_H_PART 0: hidden_nf = 256 and everything that does not depend on the width) and through
// the two-line wrapper kernels_egnn_graph_hx.hip (CMDGEN_H_PART 1: the widths 64 / 128 / 512, reached from the dispatchers below through *_hx).
#ifndef CMDGEN_H_PART
#define CMDGEN_H_PART 0
#endif
#include "cmdgen_egnn_common.h"

// ------------------------------------------------------------------------------------
// Radius graph.  One workgroup per sample; a wave scans the candidate senders of one
// receiver at a time, so neighbours come out in ascending sender order and ballots give
// both the degree and the compaction offsets.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int flat_node(int i, int nl, int pb, int qb, int Nl) {
    return i < nl ? pb + i : Nl + qb + (i - nl);
}

#if CMDGEN_H_PART == 0
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
#endif

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
#if CMDGEN_H_PART == 0
__global__ void k_edge_write(Layout lay, Work w, Dims d) { edge_write_body(lay, w, d, blockIdx.x); }
#endif

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
// host-callable launchers (C++ linkage)
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
template <int H> static void embed_only_H(const EvalLaunch& a, const float* xp, const float* xq, const float* t, hipStream_t s) {
    MT_DISPATCH(a.node_mt, launch_embed, a, xp, xq, t, nullptr, nullptr, s);
}
template <int H> static void embed_tiles_H(const EvalLaunch& a, int mt, const float* xp, const float* xq, const float* t, const float4* coef, ChainState* chain, hipStream_t s) {
    MT_DISPATCH(mt, launch_embed, a, xp, xq, t, coef, chain, s);
}
#if CMDGEN_H_PART == 0
void cmdgen_launch_embed_tiles_hx(const EvalLaunch& a, int mt, const float* xp, const float* xq, const float* t, const float4* coef, ChainState* chain, hipStream_t s);   // kernels_egnn_graph_hx.hip
void cmdgen_launch_embed_tiles(const EvalLaunch& a, int mt, const float* xp, const float* xq, const float* t, const float4* coef, ChainState* chain, hipStream_t s) {
    if (a.d.H == 256) embed_tiles_H<256>(a, mt, xp, xq, t, coef, chain, s); else cmdgen_launch_embed_tiles_hx(a, mt, xp, xq, t, coef, chain, s);
}
#else
void cmdgen_launch_embed_tiles_hx(const EvalLaunch& a, int mt, const float* xp, const float* xq, const float* t, const float4* coef, ChainState* chain, hipStream_t s) {
    switch (a.d.H) {
        case 512: embed_tiles_H<512>(a, mt, xp, xq, t, coef, chain, s); break;
        case 128: embed_tiles_H<128>(a, mt, xp, xq, t, coef, chain, s); break;
        case 64:  embed_tiles_H<64>(a, mt, xp, xq, t, coef, chain, s); break;
        default: break;   // rejected in cmdgen_create
    }
}
#endif
#if CMDGEN_H_PART == 0
void cmdgen_launch_write_embed_tiles(const EvalLaunch& a, int mt, const float* xp, const float* xq, const float* t, const float4* coef, ChainState* chain, hipStream_t s) {
    constexpr int H = 256;
    MT_DISPATCH(mt, launch_write_embed, a, xp, xq, t, coef, chain, s);
}
void cmdgen_launch_edge_count(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket, hipStream_t s) {
    const size_t shm = (size_t)a.lay.max_n * (sizeof(float4) + 3 * sizeof(int));
    hipLaunchKernelGGL(k_edge_count, dim3(a.lay.B), dim3(a.lay.max_n > 128 ? 1024 : 256), shm, s, a.lay, a.w, a.d, xh_phar, xh_pocket);
}
void cmdgen_launch_edge_write(const EvalLaunch& a, hipStream_t s) {
    const size_t shm = (size_t)a.lay.max_n * (sizeof(float4) + 3 * sizeof(int));
    hipLaunchKernelGGL(k_edge_write, dim3(a.lay.B), dim3(a.lay.max_n > 128 ? 1024 : 256), shm, s, a.lay, a.w, a.d);
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
void cmdgen_embed_only_hx(const EvalLaunch& a, const float* xp, const float* xq, const float* t, hipStream_t s);     // kernels_egnn_graph_hx.hip
// builds the cache from two embed-only passes with the time feature pinned to 0 and to 1 (t01: device [2][B])
void cmdgen_build_pocket_cache(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket, const float* t01,
                               float* c, float* P0, float* Q0, float* dh, float* dP, float* dQ, hipStream_t s) {
    if (a.lay.Np == 0) return;
    const int H = a.d.H;
    for (int stage = 0; stage < 2; ++stage) {
        const float* t = t01 + (size_t)stage * a.lay.B;
        if (H == 256) embed_only_H<256>(a, xh_phar, xh_pocket, t, s); else cmdgen_embed_only_hx(a, xh_phar, xh_pocket, t, s);
        const size_t n = stage == 0 ? (size_t)a.lay.Np * H : (size_t)H;
        hipLaunchKernelGGL(k_pocket_cache, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.lay, a.w, H, c, P0, Q0, dh, dP, dQ, stage);
    }
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
#endif      // CMDGEN_H_PART == 0
#if CMDGEN_H_PART == 1
void cmdgen_embed_only_hx(const EvalLaunch& a, const float* xp, const float* xq, const float* t, hipStream_t s) {
    switch (a.d.H) {
        case 512: embed_only_H<512>(a, xp, xq, t, s); break;
        case 128: embed_only_H<128>(a, xp, xq, t, s); break;
        case 64:  embed_only_H<64>(a, xp, xq, t, s); break;
        default: break;
    }
}
#endif
