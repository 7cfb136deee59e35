// kernels_egnn_coord.hip - k_edge_coord: EquivariantUpdate.coord_model (egnn_new.py:87-104) on 16- / 32- / 64-row tiles of the coordinate
// list (the 128-row form lives in kernels_edge128.hip).  Shared helpers: cmdgen_egnn_common.h.
// Build time: this file is compiled TWICE - as itself (CMDGEN_H_PART 0: hidden_nf = 256 and everything that does not depend on the width) and through
// the two-line wrapper kernels_egnn_coord_hx.hip (CMDGEN_H_PART 1: the widths 64 / 128 / 512, reached from the dispatchers below through *_hx).
#ifndef CMDGEN_H_PART
#define CMDGEN_H_PART 0
#endif
#include "cmdgen_egnn_common.h"

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
                                  (SAVE && sv.act6) ? sv.act6 + ((size_t)layer * sv.eccap + e0) * H : nullptr);
            lds_barrier();
            G::gemm(planes, fw, acc.a, carry);
        } else if constexpr (PL) {
            unsigned short* planes = reinterpret_cast<unsigned short*>(buf);
            constexpr int PLDA = SPLIT_PLANE_LDA(H / 2), PE = MT * PLDA;
            const typename G::Frag fw1 = G::frag(lw.W7, H / 8, H / 16, wave);
            float* pre6_o = SAVE ? sv.pre6 + ((size_t)layer * sv.eccap + e0) * H : nullptr;
            float* act6_o = (SAVE && sv.act6) ? sv.act6 + ((size_t)layer * sv.eccap + e0) * H : nullptr;
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
                               (SAVE && sv.act6) ? sv.act6 + ((size_t)layer * sv.eccap + e0) * H : nullptr,
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
// host-callable launchers (C++ linkage)
// ------------------------------------------------------------------------------------
#if CMDGEN_H_PART == 0
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
#endif
template <int H, int MT, bool SP> static void launch_coord(const EvalLaunch& a, int l, hipStream_t s) {
    const size_t shm = a.d.sin ? (size_t)(24 * H + 24 * MT) * sizeof(float) : 0;
    ++a.frag_launches;
    if (a.save) hipLaunchKernelGGL((k_edge_coord<H, MT, true, SP && H == 256>), dim3(a.coord_grid), dim3(H), 0, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)], l, *a.save);
    else if (a.pe_start) hipExtLaunchKernelGGL((k_edge_coord<H, MT, false, SP>), dim3(a.coord_grid), dim3(H), shm, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d,
                                               a.layers[unit_of(a, l)], l, TrainSave{});
    else hipLaunchKernelGGL((k_edge_coord<H, MT, false, SP>), dim3(a.coord_grid), dim3(H), shm, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)], l, TrainSave{});
}
template <int H> static void coord_tiles_H(const EvalLaunch& a, int l, hipStream_t s) { MT_DISPATCH(a.coord_mt, launch_coord, a, l, s); }
#if CMDGEN_H_PART == 0
void cmdgen_launch_coord_tiles_hx(const EvalLaunch& a, int l, hipStream_t s);     // kernels_egnn_coord_hx.hip
void cmdgen_launch_coord_tiles(const EvalLaunch& a, int l, hipStream_t s) {
    if (launch_coord_fullk(a, l, s)) return;
    if (a.d.H == 256) coord_tiles_H<256>(a, l, s); else cmdgen_launch_coord_tiles_hx(a, l, s);
}
#else
void cmdgen_launch_coord_tiles_hx(const EvalLaunch& a, int l, hipStream_t s) {
    switch (a.d.H) {
        case 512: coord_tiles_H<512>(a, l, s); break;
        case 128: coord_tiles_H<128>(a, l, s); break;
        case 64:  coord_tiles_H<64>(a, l, s); break;
        default: break;   // rejected in cmdgen_create
    }
}
#endif
