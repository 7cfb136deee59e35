// kernels_edge128.hip - the two edge kernels of an EquivariantBlock for LARGE edge lists (sampler, H = 256, split engine):
//   k_edge128<false>  GCL.edge_model + attention gate + segment sum by receiver         (egnn_new.py:31-52, :276-292)
//   k_edge128<true>   EquivariantUpdate.coord_model on the coordinate list               (egnn_new.py:87-104)
//
// Why another pair of edge kernels.  The 64-row tiles of kernels_egnn.hip stream the whole split weight (256 x 256 x 6 B = 384 KB) and
// gather 128 KB of P / Q rows through ONE CU's vector L1 per 64 edges - 512 KB per 12.3k cycles of matrix work, which is what that L1
// delivers (~45 B/clk measured): the GEMM phases ran at 79 % behind their weight stream, the tile builds waited for gathers queued behind
// the partner workgroup's stream, and the epilogue went through LDS three times (m tile out, row dots, 64-row scans); stamps in
// profiles/r03_h: 51.5k cycles per wave and tile for 12.3k cycles of MFMA issue.  Here
//   * a workgroup multiplies up to 128 rows per weight fragment (4 x 2 accumulator tiles of 32 x 32 per wave): half the weight bytes per edge;
//     the A operand is built a QUARTER of K at a time (three bf16 planes of [128][64 + 8], 55 KB) so that two workgroups still share a CU and
//     one's builds and epilogue run beside the other's MFMAs;
//   * the epilogue stays in registers: SiLU on the accumulators, the gate's (or coord_mlp.4's) row dot as in-lane FMAs + a value-halving
//     reduction over the 32 lanes that hold a row (round 6: v_permlane16_swap + DPP adds, no LDS round trip), one 2 KB exchange of the four
//     waves' partial sums, and the ordered segment sum by receiver as an IN-REGISTER SCAN (round 5; it replaced round 4's segment-sum-as-MFMA,
//     which spread a non-finite message over the whole tile through 0 x NaN): v_permlane32_swap turns the 32 x 32 accumulator layout into one
//     where a register holds ONE row, the rows are visited in list order by wave-uniform code driven by a 128-bit mask of segment starts, a
//     receiver's sum is a chain of fma(message, gate, sum) in ascending sender order like the reference's CPU scatter_add_, and a finished
//     receiver leaves as one 256-byte row segment (plain store inside the tile, one float atomic where a tile or chunk boundary cuts it);
//   * round 6, half engine: the bias enters as one rank-1 MFMA per accumulator tile, and on lists that give a workgroup more than one tile the
//     BUILD of quarter q + 1 is issued inside the GEMM over quarter q (double-buffered, XOR-swizzled planes; x_main in cmdgen_edge128_body.h);
//   * every workgroup owns ONE contiguous chunk of the list, sized so that all workgroups of the launch finish together (a chunk is cut into
//     equal tiles of 32 .. 128 rows; accumulator tiles beyond a tile's rows are not multiplied): no tail round of whole tiles.
// Results: the same sums as kernels_egnn.hip up to fp32 re-association (row dots add in another order; receiver sums in list order);
// deterministic run to run as long as a receiver's edges span at most two tiles (one float atomic each: commutative).
#include "cmdgen_dev.h"
#include <hip/hip_ext.h>

// half engine, twice: the FUSED main loop (the next quarter's build inside the GEMM; long lists) and the plain one (build -> barrier -> GEMM -> barrier;
// lists that give a workgroup a single tile - 64 C-alpha pockets - where the unrolled fused loop's larger code and prologue cost 1-2 %: profiles/r06_f)
#define E128_NPL 2
#define CMDGEN_E128_FUSED 1
namespace e128_half {
#include "cmdgen_edge128_body.h"
}
#undef CMDGEN_E128_FUSED
#undef E128_MFMA
#define CMDGEN_E128_FUSED 0
namespace e128_half_u {
#include "cmdgen_edge128_body.h"
}
#undef CMDGEN_E128_FUSED
#undef E128_NPL
#undef E128_MFMA
#define E128_NPL 3
namespace e128_bf3 {
#include "cmdgen_edge128_body.h"
}
#undef E128_NPL
#undef E128_MFMA


// launchers: true when the 128-row kernels took the launch (H = 256, split engine, sampler)
#define E128_LAUNCH(NSP, COORD_, LIVE)                                                                                                              \
    do {                                                                                                                                            \
        {                                                                                                                                           \
            const int grid = (a.e128_wgs >= 1 && a.e128_wgs <= 4 ? a.e128_wgs : 2) * a.n_cus;                                                       \
            static bool lds_set = false;     /* (one flag per expansion = per kernel instantiation) */                                              \
            if (!lds_set) { hipFuncSetAttribute(reinterpret_cast<const void*>(NSP::k_edge128<COORD_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(NSP::E128Lds)); lds_set = true; } \
            if (a.pe_start) hipExtLaunchKernelGGL(NSP::k_edge128<COORD_>, dim3(grid), dim3(256), sizeof(NSP::E128Lds), s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, a.layers[unit_of(a, l)], l, LIVE); \
            else hipLaunchKernelGGL(NSP::k_edge128<COORD_>, dim3(grid), dim3(256), sizeof(NSP::E128Lds), s, a.lay, a.w, a.d, a.layers[unit_of(a, l)], l, LIVE);         \
        }                                                                                                                                           \
    } while (0)
bool cmdgen_launch_msg128(const EvalLaunch& a, int l, hipStream_t s) {
    const WPack& W = a.layers[unit_of(a, l)].W2;
    if (a.edge_mt != 128 || a.d.H != 256 || !a.split || a.save || !W.ws) return false;
    if (a.half_engine && W.wh) { if (a.e128_fused & 1) E128_LAUNCH(e128_half, false, a.live_thr); else E128_LAUNCH(e128_half_u, false, a.live_thr); }
    else E128_LAUNCH(e128_bf3, false, a.live_thr);
    return true;
}
bool cmdgen_launch_coord128(const EvalLaunch& a, int l, hipStream_t s) {
    const WPack& W = a.layers[unit_of(a, l)].W7;
    if (a.coord_mt != 128 || a.d.H != 256 || !a.split || a.save || !W.ws) return false;
    if (a.half_engine && W.wh) { if (a.e128_fused & 2) E128_LAUNCH(e128_half, true, 0); else E128_LAUNCH(e128_half_u, true, 0); }
    else E128_LAUNCH(e128_bf3, true, 0);
    return true;
}
#undef E128_LAUNCH
