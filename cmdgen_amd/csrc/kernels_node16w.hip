// kernels_node16w.hip - k_node for SMALL batches: a 16-row tile on EIGHT waves (two per SIMD).
//
// Why.  At 64 C-alpha pockets the node launch has 236 tiles of 16 rows for 256 CUs: one workgroup per CU, and k_node<256, 16> (kernels_egnn.hip)
// runs it on four waves - ONE wave per SIMD, each owning 64 output columns.  Measured (profiles/r04_p_node16_eight_waves.txt): ~1000 cycles per
// 32-k block where its 24 MFMAs need 384; a third less weight traffic buys 6 %, NO weight traffic at all 20 % - the launch is neither bound by
// the matrix pipe nor by the weight stream but by one wave's serial issue: every LDS round trip, every s_waitcnt, every VALU instruction of the
// A-side split (4 cycles each when a wave is alone on its SIMD, 2 when two alternate; MI355X_MICROARCH.md constants table) stands in the
// critical path of the only wave the SIMD has.  Here the same tile runs on eight waves of 32 columns each: two waves per SIMD cover each
// other's stalls, the A-side split is issued at the two-wave rate, and a wave's MFMA chain per k-block is 12 long instead of 24.
// Arithmetic, weight packs (WPack::ws16), LDS images and epilogues are those of Eng<16, true> / node_tile_body: GCL.node_model
// (egnn_new.py:48-58) and the projections of the first edge / coordinate MLP layers; every accumulator sees the MFMAs of k_node<256, 16, false, true>
// in the same order.
#include "cmdgen_dev.h"
#include "cmdgen_split.h"
#include <hip/hip_ext.h>

#define NW_NPL 2
namespace nw_half {
#include "cmdgen_node16w_body.h"
}
#undef NW_NPL
#undef NW_MFMA
#define NW_NPL 3
namespace nw_bf3 {
#include "cmdgen_node16w_body.h"
}
#undef NW_NPL
#undef NW_MFMA

// launcher: true when the eight-wave kernel took the launch (H = 256, 16-row tiles on the split engine, sampler)
bool cmdgen_launch_node16w(const EvalLaunch& a, int l, hipStream_t s) {
    const LayerW& lw = a.layers[unit_of(a, l)];
    const LayerW& ln = a.layers[unit_has_next(a, l) ? unit_of(a, l) + 1 : unit_of(a, l)];
    const int nt = (a.lay.N + 15) / 16;
    if (a.save) {       // training forward: the half form with save hooks, where the step re-made the 16-row half packs (EvalLaunch::save_half16)
        if (a.d.H != 256 || a.node_mt != 16 || !a.save_half16 || !lw.W3.wh16 || !lw.W3.wh_dev) return false;
        TrainSave sv = *a.save; sv.slot = unit_of(a, l);
        hipLaunchKernelGGL(nw_half::k_node16w<true>, dim3(nt), dim3(512), 0, s, a.lay, a.w, a.d, lw, ln, l, node_flags(a, l), sv);
        return true;
    }
    if (a.d.H != 256 || a.node_mt != 16 || !a.split16 || !a.node16w || !lw.W3.ws16) return false;
    if (a.half_engine && lw.W3.wh16) {
        if (a.pe_start) hipExtLaunchKernelGGL(nw_half::k_node16w<false>, dim3(nt), dim3(512), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, lw, ln, l, node_flags(a, l), TrainSave{});
        else hipLaunchKernelGGL(nw_half::k_node16w<false>, dim3(nt), dim3(512), 0, s, a.lay, a.w, a.d, lw, ln, l, node_flags(a, l), TrainSave{});
    } else {
        if (a.pe_start) hipExtLaunchKernelGGL(nw_bf3::k_node16w<false>, dim3(nt), dim3(512), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, lw, ln, l, node_flags(a, l), TrainSave{});
        else hipLaunchKernelGGL(nw_bf3::k_node16w<false>, dim3(nt), dim3(512), 0, s, a.lay, a.w, a.d, lw, ln, l, node_flags(a, l), TrainSave{});
    }
    return true;
}
