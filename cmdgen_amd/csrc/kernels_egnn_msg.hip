// kernels_egnn_msg.hip - k_edge_msg: GCL.edge_model + attention gate + segment sum by receiver (egnn_new.py:31-52) on 16- / 32- / 64-row
// tiles of the compact edge list (the 128-row form lives in kernels_edge128.hip).  Shared helpers: cmdgen_egnn_common.h.
// Build time: this file is compiled TWICE - as itself (CMDGEN_H_PART 0: hidden_nf = 256 and everything that does not depend on the width) and through
// the two-line wrapper kernels_egnn_msg_hx.hip (CMDGEN_H_PART 1: the widths 64 / 128 / 512, reached from the dispatchers below through *_hx).
#ifndef CMDGEN_H_PART
#define CMDGEN_H_PART 0
#endif
#include "cmdgen_egnn_common.h"

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
                                                     SAVE ? sv.pre1 + ((size_t)sv.slot * sv.ecap + e0) * H : nullptr,
                                                     (SAVE && sv.act1) ? sv.act1 + ((size_t)sv.slot * sv.ecap + e0) * H : nullptr);
            lds_barrier();
            STAMP(1);
            if (!(ablate & 4)) G::gemm(planes, fw, acc.a, carry);
        } else if constexpr (PL) {
            // two half-K passes: build columns [0,128) as bf16 planes -> GEMM over k 0..127 -> build [128,256) -> GEMM over the rest
            unsigned short* planes = reinterpret_cast<unsigned short*>(buf);
            constexpr int PLDA = SPLIT_PLANE_LDA(H / 2), PE = MT * PLDA;
            const typename G::Frag fw1 = G::frag(lw.W2, H / 8, H / 16, wave);
            float* pre1_o = SAVE ? sv.pre1 + ((size_t)sv.slot * sv.ecap + e0) * H : nullptr;
            float* act1_o = (SAVE && sv.act1) ? sv.act1 + ((size_t)sv.slot * sv.ecap + e0) * H : nullptr;
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
                                                  SAVE ? sv.pre1 + ((size_t)sv.slot * sv.ecap + e0) * H : nullptr,
                                                  (SAVE && sv.act1) ? sv.act1 + ((size_t)sv.slot * sv.ecap + e0) * H : nullptr,
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
            const size_t o = ((size_t)sv.slot * sv.ecap + e0) * H;
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
                if (SAVE && d.attention && r < ne) sv.z[(size_t)sv.slot * sv.ecap + e0 + r] = zl;
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
// host-callable launchers (C++ linkage)
// ------------------------------------------------------------------------------------
template <int H, int MT, bool SP> static void launch_msg(const EvalLaunch& a, int l, hipStream_t s) {
    const size_t shm = a.d.sin ? (size_t)(24 * H + 24 * MT) * sizeof(float) : 0;      // sin_embedding: feature columns + the tile's features (edge_msg_body)
    // training forward: the split engine only where the step re-packs split weights (H = 256: edge_mlp.2 / coord_mlp.2)
    ++a.frag_launches;
    if (a.save) { TrainSave sv = *a.save; sv.slot = unit_of(a, l); hipLaunchKernelGGL((k_edge_msg<H, MT, true, SP && H == 256>), dim3(a.edge_grid), dim3(H), 0, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)], l, a.ablate, sv, 0); }
    else if (a.pe_start) hipExtLaunchKernelGGL((k_edge_msg<H, MT, false, SP>), dim3(a.edge_grid), dim3(H), shm, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d,
                                               a.layers[unit_of(a, l)], l, a.ablate, TrainSave{}, a.live_thr);
    else hipLaunchKernelGGL((k_edge_msg<H, MT, false, SP>), dim3(a.edge_grid), dim3(H), shm, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)], l, a.ablate, TrainSave{}, a.live_thr);
}
#if CMDGEN_H_PART == 0
// 32-row sampler tiles on the split engine: full-K planes (one build, one GEMM per tile; see cmdgen_split.h) unless CMDGEN_EDGE_FULLK=0
static bool launch_msg_fullk(const EvalLaunch& a, int l, hipStream_t s) {
    if (!a.edge_fullk || (a.save && !a.save_half) || !a.split || a.d.H != 256 || a.edge_mt != 32) return false;
    const LayerW& lw = a.layers[unit_of(a, l)];
    if (a.save) {       // training forward on the half engine (packs and scale re-made on the device every step: WPack::wh_dev)
        TrainSave sv = *a.save; sv.slot = unit_of(a, l);
        hipLaunchKernelGGL((k_edge_msg<256, 32, true, true, 2>), dim3(a.edge_grid), dim3(256), 0, s, a.lay, a.w, a.d, lw, l, a.ablate, sv, 0);
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
#endif
template <int H> static void msg_tiles_H(const EvalLaunch& a, int l, hipStream_t s) { MT_DISPATCH(a.edge_mt, launch_msg, a, l, s); }
// the launch of one block's message kernel on <= 64-row tiles: the full-K 32-row form where it applies, else the generic dispatch
#if CMDGEN_H_PART == 0
void cmdgen_launch_msg_tiles_hx(const EvalLaunch& a, int l, hipStream_t s);       // kernels_egnn_msg_hx.hip
void cmdgen_launch_msg_tiles(const EvalLaunch& a, int l, hipStream_t s) {
    if (launch_msg_fullk(a, l, s)) return;
    if (a.d.H == 256) msg_tiles_H<256>(a, l, s); else cmdgen_launch_msg_tiles_hx(a, l, s);
}
#else
void cmdgen_launch_msg_tiles_hx(const EvalLaunch& a, int l, hipStream_t s) {
    switch (a.d.H) {
        case 512: msg_tiles_H<512>(a, l, s); break;
        case 128: msg_tiles_H<128>(a, l, s); break;
        case 64:  msg_tiles_H<64>(a, l, s); break;
        default: break;   // rejected in cmdgen_create
    }
}
#endif

// (the kernel the evaluation itself would run for this block: the 128-row kernel, then the full-K 32-row tiles, then the generic dispatch;
// weight unit of the block's first GCL when a block has several)
#if CMDGEN_H_PART == 0
template <int H> static void launch_msg_only_H(const EvalLaunch& a, int layer, hipStream_t s) {
    a.unit = layer * a.d.S;
    if (!cmdgen_launch_msg128(a, layer, s) && !launch_msg_fullk(a, layer, s)) MT_DISPATCH(a.edge_mt, launch_msg, a, layer, s);
    a.unit = -1;
}
void cmdgen_launch_edge_msg_only_hx(const EvalLaunch& a, int layer, hipStream_t s);       // kernels_egnn_msg_hx.hip
void cmdgen_launch_edge_msg_only(const EvalLaunch& a, int layer, hipStream_t s) {
    if (a.d.H == 256) launch_msg_only_H<256>(a, layer, s); else cmdgen_launch_edge_msg_only_hx(a, layer, s);
}
#else
template <int H> static void launch_msg_only_H(const EvalLaunch& a, int layer, hipStream_t s) {
    a.unit = layer * a.d.S;
    MT_DISPATCH(a.edge_mt, launch_msg, a, layer, s);          // (the 128-row and full-K forms are hidden_nf = 256 only)
    a.unit = -1;
}
void cmdgen_launch_edge_msg_only_hx(const EvalLaunch& a, int layer, hipStream_t s) {
    switch (a.d.H) {
        case 512: launch_msg_only_H<512>(a, layer, s); break;
        case 128: launch_msg_only_H<128>(a, layer, s); break;
        case 64:  launch_msg_only_H<64>(a, layer, s); break;
        default: break;
    }
}
#endif
