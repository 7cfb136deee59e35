// kernels_node64.hip - k_node for LARGE batches: 64-row tiles, the A operand as producer-side bf16 planes.
//
// Why.  GCL.node_model + the projections (egnn_new.py:48-58 and the first-layer factorisation of the two edge MLPs) are a
// chain of seven [rows, 256] x [256, 256] products per tile.  On the split engine (cmdgen_split.h: six bf16 MFMAs per fp32
// product) the node kernels of kernels_egnn.hip keep the tile as an fp32 LDS image and every wave splits the fragments it
// reads in registers - four waves splitting the SAME A fragments, ~130 VALU operations per k-block next to 24 MFMAs: measured
// ~1100 cycles per k-block where the MFMAs take 768 (profiles/r03_m_node64.txt, which also holds the numbers of the register-split
// 64-row variant this file replaced).  Here whoever WRITES the tile splits each element once and stores three bf16 planes (as the
// edge kernels do), so the GEMM loops carry no conversion work at all: 24 MFMAs and 12 loads per k-block, one load pinned in the
// shadow of every two MFMAs (~840 cycles per k-block).
//
// LDS: one plane image of 64 rows x 256 k (3 x 64 x 264 bf16 = 99 KB; one workgroup per CU) that holds, one after the other,
// h, agg / nf, T = SiLU(.), h_new.  The residual's h comes back from L2 in the accumulator layout while the W4 product runs;
// agg is requested into registers while the h-part of the first product runs.
// Tile: 64 rows x 256 columns, 4 waves x 64 columns (2 x 2 accumulator tiles of 32 x 32), v_mfma_f32_32x32x16_bf16; weight
// fragments three k-blocks ahead in a ring of four register sets carried from one GEMM of the chain into the next.
//
// Round 6: what bounds a node tile is the chain's 1.8 MB of weight fragments through its CU's L1 (64 B/clk; every tile streams all of it,
// co-resident workgroups share nothing: TCP_TCC_READ_REQ, profiles/r06_n_node64e.txt).  Four tiles of one body, picked by how the layout's
// tiles fill the CUs (make_launch): k_node32p (two workgroups per CU: a partner beside every phase, but 85 B/clk asked in the GEMM phases),
// k_node64e (64 rows on eight waves: 43 B/clk, GEMM phases at the matrix pipe's rate, no partner - so its stores leave under the next GEMM's
// MFMAs), k_node64d (lean 64-row tile, two workgroups per CU: both), and k_node64 (the round-3 tile, option node64 = 1).
#include "cmdgen_dev.h"
#include <hip/hip_ext.h>

#define N64_NPL 2
namespace n64_half {
#include "cmdgen_node_planes.h"
__global__ __launch_bounds__(256, 1) void k_node64(Layout lay, Work w, Dims d, LayerW lw, LayerW lw_next, int layer, int has_next) {
    __shared__ __attribute__((aligned(16))) unsigned short planes[NPL * 64 * NPLD + 64 + 64 * 256 * 2];      // + the A prefetch's overshoot past the last row + the fp32 h tile (residual)
    node_planes_tile<64>(planes, lay, w, d, lw, lw_next, layer, has_next, (int)blockIdx.x * 64, lay.N);
}
// the same tile at 32 rows (measurement aid: option node64 = 32 runs every node tile of a layout through it; profiles/r03_m_node64.txt)
__global__ __launch_bounds__(256, 1) void k_node32p(Layout lay, Work w, Dims d, LayerW lw, LayerW lw_next, int layer, int has_next) {
    __shared__ __attribute__((aligned(16))) unsigned short planes[NPL * 32 * NPLD + 64 + 32 * 256 * 2];
    node_planes_tile<32>(planes, lay, w, d, lw, lw_next, layer, has_next, (int)blockIdx.x * 32, lay.N);
}
// the 64-row four-wave tile with TWO workgroups per CU (ring of four k-blocks, no fp32 h tile: 68 KB of LDS, <= 256 registers): for layouts with more
// 64-row tiles than CUs.  Eight waves per CU ask the L1 for 43 B/clk of weight fragments (k_node32p's two workgroups: 85, the L1 fills 64), and the two
// workgroups run their memory phases beside each other's GEMMs.  Option node64 = 2.
__global__ __launch_bounds__(256, 2) void k_node64d(Layout lay, Work w, Dims d, LayerW lw, LayerW lw_next, int layer, int has_next) {
    __shared__ __attribute__((aligned(16))) unsigned short planes[NPL * 64 * NPLD + 64];
    node_planes_tile<64, 2, true>(planes, lay, w, d, lw, lw_next, layer, has_next, (int)blockIdx.x * 64, lay.N);
}
// the 64-row tile on EIGHT waves (512 threads, one 32-column tile per wave): the weights of the chain are streamed once per 64 rows as in k_node64
// (k_node32p's two co-resident workgroups stream them twice per CU - 779 MB per launch at 256 pockets, ~17 TB/s out of the L2s, TCP_TCC_READ_REQ),
// with two waves per SIMD to overlap the epilogues and memory phases that k_node64's single wave per SIMD runs back to back.  Option node64 = 8.
__global__ __launch_bounds__(512, 1) void k_node64e(Layout lay, Work w, Dims d, LayerW lw, LayerW lw_next, int layer, int has_next) {
    __shared__ __attribute__((aligned(16))) unsigned short planes[2 * (NPL * 64 * NPLD + 64)];          // two plane images (h / T, agg / h_new), each + the A prefetch's overshoot
    node_planes_tile<64, 1>(planes, lay, w, d, lw, lw_next, layer, has_next, (int)blockIdx.x * 64, lay.N);
}
}
#undef N64_NPL
#undef N64_MFMA
#undef NPLD
#undef N64_ZERO
#undef N64_ROW
#define N64_NPL 3
namespace n64_bf3 {
#include "cmdgen_node_planes.h"
__global__ __launch_bounds__(256, 1) void k_node64(Layout lay, Work w, Dims d, LayerW lw, LayerW lw_next, int layer, int has_next) {
    __shared__ __attribute__((aligned(16))) unsigned short planes[NPL * 64 * NPLD + 64];
    node_planes_tile<64>(planes, lay, w, d, lw, lw_next, layer, has_next, (int)blockIdx.x * 64, lay.N);
}
__global__ __launch_bounds__(256, 1) void k_node32p(Layout lay, Work w, Dims d, LayerW lw, LayerW lw_next, int layer, int has_next) {
    __shared__ __attribute__((aligned(16))) unsigned short planes[NPL * 32 * NPLD + 64];
    node_planes_tile<32>(planes, lay, w, d, lw, lw_next, layer, has_next, (int)blockIdx.x * 32, lay.N);
}
}

// launcher: true when the 64-row kernel took the launch (H = 256, split engine, sampler)
#define N64_LAUNCH(NSP, EIGHT)                                                                                                                              \
    do {                                                                                                                                             \
        const LayerW& lw_ = a.layers[unit_of(a, l)]; const LayerW& ln_ = a.layers[unit_has_next(a, l) ? unit_of(a, l) + 1 : unit_of(a, l)];          \
        if (a.node64 == 2 && EIGHT) {                                                                                                                \
            const int nt = (a.lay.N + 63) / 64;                                                                                                      \
            if (a.pe_start) hipExtLaunchKernelGGL(n64_half::k_node64d, dim3(nt), dim3(256), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, lw_, ln_, l, node_flags(a, l)); \
            else hipLaunchKernelGGL(n64_half::k_node64d, dim3(nt), dim3(256), 0, s, a.lay, a.w, a.d, lw_, ln_, l, node_flags(a, l));                  \
        } else if (a.node64 == 8 && EIGHT) {                                                                                                         \
            const int nt = (a.lay.N + 63) / 64;                                                                                                      \
            if (a.pe_start) hipExtLaunchKernelGGL(n64_half::k_node64e, dim3(nt), dim3(512), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, lw_, ln_, l, node_flags(a, l)); \
            else hipLaunchKernelGGL(n64_half::k_node64e, dim3(nt), dim3(512), 0, s, a.lay, a.w, a.d, lw_, ln_, l, node_flags(a, l));                  \
        } else if (a.node64 == 32) {                                                                                                                 \
            const int nt32 = (a.lay.N + 31) / 32;                                                                                                    \
            if (a.pe_start) hipExtLaunchKernelGGL(NSP::k_node32p, dim3(nt32), dim3(256), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, lw_, ln_, l, node_flags(a, l)); \
            else hipLaunchKernelGGL(NSP::k_node32p, dim3(nt32), dim3(256), 0, s, a.lay, a.w, a.d, lw_, ln_, l, node_flags(a, l));                     \
        } else {                                                                                                                                     \
            const int nt = (a.lay.N + 63) / 64;                                                                                                      \
            if (a.pe_start) hipExtLaunchKernelGGL(NSP::k_node64, dim3(nt), dim3(256), 0, s, a.pe_start, a.pe_stop, 0, a.lay, a.w, a.d, lw_, ln_, l, node_flags(a, l)); \
            else hipLaunchKernelGGL(NSP::k_node64, dim3(nt), dim3(256), 0, s, a.lay, a.w, a.d, lw_, ln_, l, node_flags(a, l));                        \
        }                                                                                                                                            \
    } while (0)
bool cmdgen_launch_node64(const EvalLaunch& a, int l, hipStream_t s) {
    if (a.d.H != 256 || !a.split || a.save || !a.node64 || !a.layers[unit_of(a, l)].W3.ws) return false;
    if (a.half_engine && a.layers[unit_of(a, l)].W3.wh) N64_LAUNCH(n64_half, true); else N64_LAUNCH(n64_bf3, false);
    return true;
}
#undef N64_LAUNCH
