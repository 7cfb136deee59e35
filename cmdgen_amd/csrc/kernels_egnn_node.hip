// kernels_egnn_node.hip - k_node: GCL.node_model for an MT-node tile (egnn_new.py:48-58) and the projections P_c | Q_c / P | Q the later
// kernels gather, on the register-split tiles (16 / 32 / 64 rows; the plane tiles live in kernels_node64.hip, the eight-wave 16-row tile
// in kernels_node16w.hip).  Shared helpers: cmdgen_egnn_common.h.
// Build time: this file is compiled TWICE - as itself (CMDGEN_H_PART 0: hidden_nf = 256 and everything that does not depend on the width) and through
// the two-line wrapper kernels_egnn_node_hx.hip (CMDGEN_H_PART 1: the widths 64 / 128 / 512, reached from the dispatchers below through *_hx).
#ifndef CMDGEN_H_PART
#define CMDGEN_H_PART 0
#endif
#include "cmdgen_egnn_common.h"

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
                if (SAVE) reinterpret_cast<float4*>(sv.aggn + ((size_t)sv.slot * lay.N + row0 + r) * H)[c4] = v;
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
            if (SAVE && r < nvalid) reinterpret_cast<float4*>(sv.aggn + ((size_t)sv.slot * lay.N + row0 + r) * H)[c4] = v;
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
        const size_t o = ((size_t)sv.slot * lay.N + row0) * H;
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
                if (SAVE) reinterpret_cast<float4*>(sv.h + ((size_t)(sv.slot + 1) * lay.N + row0 + r) * H)[c4] = hv;   // h entering block layer+1
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
// host-callable launchers (C++ linkage)
// ------------------------------------------------------------------------------------
// SAVE variants (training forward) keep the activations; the sampler's instantiations carry no trace of the stores
template <int H, int MT, bool SP> static void launch_node(const EvalLaunch& a, int l, hipStream_t s) {
    if constexpr (MT == 16 && !SP && H >= 128) {
        // 16-row tiles on the split engine (v_mfma_f32_16x16x32_bf16): opt-in, see DESIGN section 4a for why it is not the default
        if (a.split16 && !a.save && a.layers[unit_of(a, l)].W3.ws16) { launch_node<H, 16, true>(a, l, s); return; }
    }
    const int nt = (a.lay.N + MT - 1) / MT;
    ++a.frag_launches;
    if (a.save) {
        TrainSave sv = *a.save; sv.slot = unit_of(a, l);
        hipLaunchKernelGGL((k_node<H, MT, true, false>), dim3(nt), dim3(H), 0, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)],
                           a.layers[unit_has_next(a, l) ? unit_of(a, l) + 1 : unit_of(a, l)], l, node_flags(a, l), sv);
    }
    else if (a.pe_start) hipExtLaunchKernelGGL((k_node<H, MT, false, SP>), dim3(nt), dim3(H), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d,
                                               a.layers[unit_of(a, l)], a.layers[unit_has_next(a, l) ? unit_of(a, l) + 1 : unit_of(a, l)], l, node_flags(a, l), TrainSave{});
    else hipLaunchKernelGGL((k_node<H, MT, false, SP>), dim3(nt), dim3(H), 0, s, a.lay, a.w, a.d, a.layers[unit_of(a, l)],
                            a.layers[unit_has_next(a, l) ? unit_of(a, l) + 1 : unit_of(a, l)], l, node_flags(a, l), TrainSave{});
}
template <int H> static void node_tiles_H(const EvalLaunch& a, int l, hipStream_t s) { MT_DISPATCH(a.node_mt, launch_node, a, l, s); }
#if CMDGEN_H_PART == 0
void cmdgen_launch_node_tiles_hx(const EvalLaunch& a, int l, hipStream_t s);      // kernels_egnn_node_hx.hip
void cmdgen_launch_node_tiles(const EvalLaunch& a, int l, hipStream_t s) {
    if (a.d.H == 256) node_tiles_H<256>(a, l, s); else cmdgen_launch_node_tiles_hx(a, l, s);
}
#else
void cmdgen_launch_node_tiles_hx(const EvalLaunch& a, int l, hipStream_t s) {
    switch (a.d.H) {
        case 512: node_tiles_H<512>(a, l, s); break;
        case 128: node_tiles_H<128>(a, l, s); break;
        case 64:  node_tiles_H<64>(a, l, s); break;
        default: break;   // rejected in cmdgen_create
    }
}
#endif
