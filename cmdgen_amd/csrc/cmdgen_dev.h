// cmdgen_dev.h - device-side types and helpers shared by the gfx950 kernels.
//
// Written for CDNA4 only: 64-lane wavefronts, v_mfma_f32_32x32x2_f32 (exact fp32,
// 64 cycles/SIMD), 160 KiB LDS per CU.  No CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "cmdgen_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CMDGEN_MAX_LAYERS 16
#define CMDGEN_MAX_SMALL 64     // upper bound for phar_nf*2, residue_nf*2, joint_nf+1

// ---------------------------------------------------------------------------------
// Packed weight layout ("B fragments" of v_mfma_f32_32x32x2_f32)
//
// A Linear weight W[out][in] (nn.Linear layout) is stored as float4
//     Wp[(nt * KB + kb) * 64 + lane] = { W[o][k+0], W[o][k+1], W[o][k+2], W[o][k+3] }
//     o = 32*nt + (lane & 31),  k = 8*kb + 4*(lane >> 5)
// with nt over out/32 and kb over in/8 (in padded with zeros to a multiple of 8).
// One 16-byte load per lane then feeds four MFMA k-steps: in step j lanes 0-31 supply
// k = 8kb+j and lanes 32-63 supply k = 8kb+4+j, and the A operand is read from LDS with
// the same pairing (one ds_read_b128 per lane).  The k order inside a block of 8 is thus
// (0,4),(1,5),(2,6),(3,7): a fixed re-association of the fp32 dot product.
// ---------------------------------------------------------------------------------

struct WPack {                  // one Linear weight in the MFMA fragment orders (see above / below)
    const float4* w32;          // v_mfma_f32_32x32x2_f32 order  (64- and 32-row tiles)
    const float4* w16;          // v_mfma_f32_16x16x4_f32 order  (16-row tiles)
    const void*   ws;           // three bf16 pieces per weight in v_mfma_f32_32x32x16_bf16 order (cmdgen_split.h); null in training
    const void*   ws16;         // the same pieces in v_mfma_f32_16x16x32_bf16 order (16-row tiles on the split engine); null in training
    const void*   wh;           // TWO fp16 pieces of (weight * wh_scale) per weight in v_mfma_f32_32x32x16_f16 order (cmdgen_split.h, "half" engine); null in training
    const void*   wh16;         // the same two pieces in v_mfma_f32_16x16x32_f16 order (16-row tiles); null in training / where in % 128 != 0
    float         wh_scale;     // the power of two the matrix was multiplied by before its split (keeps both pieces in fp16's normal range)
    float         wh_inv;       // 1 / wh_scale
    const float*  wh_dev;       // training forward: {wh_scale, 1 / wh_scale} of a pack re-made on the device every step (k_repack_half); null in the sampler
};

struct LayerW {                 // device pointers to one EquivariantBlock's packed weights
    // GCL.edge_mlp (egnn_new.py:15-19): layer 0 split by input columns [h_row | h_col | radial | d0]
    WPack Wpq_e;                // [2H out][H in]: rows 0..H-1 act on h_row (-> P), H..2H-1 on h_col (-> Q)
    const float*  b1;           // folded into P
    const float*  wr_e;         // column 2H   (radial)
    const float*  wd_e;         // column 2H+1 (d0)
    const float*  we_e;         // sin_embedding: columns 2H .. 2H+23 transposed, [24][H] (wr_e / wd_e unused); else null
    WPack W2;                   // edge_mlp.2 [H][H]
    const float*  b2;
    const float*  wa;           // att_mlp.0 weight [H]
    const float*  ba;           // att_mlp.0 bias [1] (device pointer: the training forward reads it straight from theta)
    // GCL.node_mlp (egnn_new.py:21-24)
    WPack W3;                   // node_mlp.0 [H][2H]  (in = [h | agg])
    const float*  b3;
    WPack W4;                   // node_mlp.2 [H][H]
    const float*  b4;
    // EquivariantUpdate.coord_mlp (egnn_new.py:78-83)
    WPack Wpq_c;                // coord_mlp.0 split like Wpq_e
    const float*  b6;
    const float*  wr_c;
    const float*  wd_c;
    const float*  we_c;         // (as we_e, for coord_mlp.0)
    WPack W7;                   // coord_mlp.2 [H][H]
    const float*  b7;
    const float*  w5;           // coord_mlp.4 weight [H], no bias
};

struct SmallW {                 // encoders / decoders / embeddings, plain [out][in] fp32 on device
    const float *pe0_w, *pe0_b, *pe2_w, *pe2_b;     // phar_encoder   (dynamics.py:21-25)
    const float *pd0_w, *pd0_b, *pd2_w, *pd2_b;     // phar_decoder   (:27-31)
    const float *re0_w, *re0_b, *re2_w, *re2_b;     // residue_encoder (:33-37)
    const float *rd0_w, *rd0_b, *rd2_w, *rd2_b;     // residue_decoder (:39-43)
    const float *emb_wT, *emb_b;                    // egnn.embedding, stored transposed [J+1][H]
    const float *embo_wT, *embo_b;                  // egnn.embedding_out, stored transposed [H][J+1]
    const float *enc_pack;                          // the eight encoder tensors (pe0_w, pe0_b, pe2_w, pe2_b, re0_w, re0_b, re2_w, re2_b)
                                                    // back to back in one buffer (k_embed copies them to LDS); null in training
};

struct Dims {
    int P, R, J, H, L;          // phar_nf, residue_nf, joint_nf, hidden_nf, n_layers
    int dyn;                    // J + condition_time
    int attention, use_tanh, condition_time;
    int no_com;                 // SimpleConditionalDDPM: no centre-of-mass projection in the sampler
    int joint;                  // update_pocket_coords (mode 'joint'): pocket nodes move, velocity COM removed
    float cutoff2;              // cutoff^2, < 0: no cutoff
    float norm_constant, norm_factor, coords_range;
    int S;                      // inv_sublayers: GCLs per block (the weight units of an evaluation are L * S, LayerW of unit l * S + s)
    int sin;                    // sin_embedding (egnn_new.py:249-260): the two distance features of an edge are 12 sines / cosines each
    float sin_freq[6];          // 2 pi 4^k / 15 in the reference's fp32 arithmetic
    int agg_mean;               // aggregation_method 'mean': segment sums are divided by Work::adiv[receiver] instead of norm_factor
    float norm_x, norm_h, bias_h;
};

struct Layout {                 // one flat batch; all pointers device
    int B, Nl, Np, N;           // samples, phar nodes, pocket nodes, total
    int Nm;                     // nodes whose coordinates move: Nl (conditional) or N (joint); flat ids < Nm
    int max_n;                  // max nodes of one sample
    const int* num_phar;        // [B]
    const int* num_pocket;      // [B]
    const int* phar_base;       // [B] exclusive prefix of num_phar
    const int* pocket_base;     // [B]
    const int* node_sample;     // [N] sample index of each node in flat order (phar first)
    const int64_t* pocket_gid;  // [B] global pocket ids (Philox key)
};

struct Work {                   // per-layout workspace; all pointers device
    float4* X0;                 // [Nm] input coordinates of the moving nodes of this evaluation
    float4* XP;                 // [Np] pocket input coordinates
    float4* XL;                 // [L][Nm] coordinates entering block l (l >= 1 materialised by the node kernel)
    float4* ACC;                // [L][Nm] sum of coordinate messages of block l (zeroed per evaluation)
    float*  h;                  // [N][H]
    float*  P;                  // [N][H]  edge-MLP layer-0 receiver part (+b1)
    float*  Q;                  // [N][H]  sender part
    float*  Pc;                 // [N][H]  coord-MLP receiver part (+b6), only phar rows read
    float*  Qc;                 // [N][H]
    float*  agg;                // [N][H]  zero between blocks
    float*  adiv;               // [N] (flat node order) agg_mean only: max(degree, 1) as the divisor of the node's segment sums, written by pass 2 of the radius graph
    int*    degL;               // [N] degree in sample-local order
    int     hop_levels;         // how many hop levels the graph pass computes into need_qc (1, or n_layers when every block skips its dead tiles)
    int*    need_qc;            // [N] hop level from the moving nodes: 0 moves, 1 sends along a coordinate edge (its Q_c row is read), 2.. L, 255 none; may be null
    int*    pocketE;            // [B] edges per sample
    int*    pocketEph;          // [B] edges with phar receiver per sample
    int*    pocketEns;          // [B] ... of those that are not self loops
    int*    pocketEnsQ;         // [B] edges with pocket receiver that are not self loops (joint mode's coordinate list)
    int*    ehop;               // [Ecap] hop level of every listed edge's RECEIVER (see need_qc); may be null
    int*    erow; int* ecol; float* ed0;        // [Ecap] compact edge list sorted by flat (row, col); the first
                                                //        pocketEph-sum entries are the phar-receiver edges
    int*    crow; int* ccol; float* cd0;        // [Eccap] coordinate-update edges: moving receivers, self loops dropped
    int*    totals;             // [0]=E, [1]=Ec (coordinate list) of the current evaluation
    unsigned long long* counters;   // cmdgen_counters
    int*    nan_flag;           // [1] set by readout when any velocity is NaN
    float*  eps_tmp;            // [Nl][3+P] evaluation output used by the chain
    unsigned long long* dbg;    // [64] diagnostic builds only (-DCMDGEN_STAMPS): summed in-kernel cycle stamps
};

// divisor of node n's segment sums (egnn_new.py:277-292): normalization_factor ('sum'), or the receiver's edge count ('mean')
__device__ __forceinline__ float agg_div(const Work& w, const Dims& d, int n) { return d.agg_mean ? w.adiv[n] : d.norm_factor; }

struct ChainState {             // device-resident denoising-loop state
    int step;                   // evaluations completed so far = index into coef[] of the NEXT evaluation; bumped by
                                // k_readout (which does not read it), so no kernel reads it while it changes: the
                                // evaluation kernels of step e read e, the sampler kernel that follows reads step - 1
    int K;                      // posterior steps
    int pad0, pad1;
};

struct ChainBuf {               // device pointers owned by the handle for one chain
    float* z_phar;              // [Nl][3+P] current z_t (normalised space)
    float* xh_pocket;           // [Np][3+R] current (translated) pocket, h columns normalised
    const float4* coef;         // [K+1]: per posterior step (alpha_ts, sigma2_ts/alpha_ts/sigma_t, sigma_ts*sigma_s/sigma_t, t);
                                //        [K] = final decode (sigma_0, alpha_0, exp(gamma_0/2), t=0)
    const float* noise;         // [K+2][Nl][3+P] or null (Philox on device)
    unsigned long long seed;
    float* z_steps;             // [K][Nl][3+P] or null
    float* pocket_steps;        // [K][Np][3] or null
    unsigned int* check;        // [K+3][2] float bits: max|x|, max|sum x| per check point
    ChainState* state;
};

struct JointBuf {               // joint-model chain (EnVariationalDiffusion.sample / .inpaint), device pointers
    float* z_phar;              // [Nl][3+P] current z
    float* z_pocket;            // [Np][3+R]
    float* e_phar;              // [Nl][3+P] scratch: the combined draw in use (x part COM-projected)
    float* e_pocket;            // [Np][3+R]
    float* zk_phar;             // [Nl][3+P] scratch: noised known part (inpainting)
    float* zk_pocket;           // [Np][3+R]
    float* x0_phar;             // [Nl][3+P] centred known input [x | one_hot] (inpainting), raw scale (quirk Q14)
    float* x0_pocket;           // [Np][3+R]
    const float* fix_phar;      // [Nl] 1 = known; null: plain sampling
    const float* fix_pocket;    // [Np]
    const float4* coef;         // [n_steps+1] (alpha_ts, sigma2_ts/alpha_ts/sigma_t, sigma_ts*sigma_s/sigma_t, t); last: decode row
    const float4* coef2;        // [n_steps]   (alpha_s, sigma_s, jump alpha_t|s, jump sigma_t|s)
    const int4* iop;            // [n_steps+1] (flags: 1 = jump back after the step, first draw index, 0, 0)
    const float* noise;         // [D][Nl*(3+P) + Np*(3+R)] or null
    unsigned long long seed;
    float* z_steps;             // [n_steps][Nl*(3+P) + Np*(3+R)] or null
    unsigned int* check;        // [n_steps+3][2]
    ChainState* state;
};

struct PocketCache {            // chain-invariant part of k_embed's output for POCKET rows (conditional sampler): the pocket's
                                // features never change during a chain and the embedding is affine in the time feature, so
                                //   h(t) = c + t dh,  P(t) = P0 + t dP,  Q(t) = Q0 + t dQ   (dh, dP, dQ: one row of H values each)
    const float *c, *P0, *Q0;   // [Np][H] values at t = 0   (null: no cache - every row takes the full path)
    const float *dh, *dP, *dQ;  // [H]     value(t = 1) - value(t = 0)
};

struct TrainSave {              // activation store of the TRAINING forward (cmdgen_train.hip); every pointer null when sampling.
    // The fused evaluation kernels write what the backward pass reads, so the training forward IS the sampler's
    // evaluation (3 launches per block) instead of a layer-by-layer pass.  Per-block arrays: base pointer + l * stride.
    float *enc1_l, *enca_l, *enc1_p, *enca_p;   // encoder layer 0, pre-activation and SiLU: [Nl][2P], [Np][2R]
    float *hdyn;                                // [N][dyn] encoder output | time column
    float *h;                                   // [L+1][N][H] node features entering every block (h[L] = final)
    float *pre1, *act1, *pre2, *act2, *z;       // edge MLP: [L][ecap][H] x4, attention logit [L][ecap]
    float *aggn, *pre3, *nact;                  // node MLP: [L][N][H]
    float *pre6, *act6, *pre7, *act7, *phi;     // coordinate MLP: [L][eccap][H] x4, head output before tanh [L][eccap]
    float *hfin, *dec1, *deca, *dec_out;        // readout: [N][dyn], phar decoder [Nl][2P] x2, [Nl][P]
    float *qdec1, *qdeca, *qdec_out;            // residue decoder (joint model) [Np][2R] x2, [Np][R]
    size_t ecap, eccap;                         // row capacities of the per-block edge arrays
    int slot;                                   // index of this launch's GCL in the per-GCL arrays (h, pre1 .. z, aggn .. nact): block * inv_sublayers + sub (set by the launcher; the coordinate arrays are per block)
};

struct TrainTune {              // launch choices of the training step's gradient kernels (cmdgen_set_option; defaults from sweeps on MI355X)
    int wgrad_split = -1;       // weight gradients as three-piece split products: -1 = wherever the handle runs the split engine (tile by wgrad_k128), 0 = never (fp32 instruction), 1 = always on 128 x 128 tiles where the shape allows
    int wgrad_tile = 0;         // 64: never the 128 x 128-tile kernel
    int wgrad_k128 = 131072;    // rows x 128-tiles of a launch from which a three-piece weight gradient uses 128 x 128 tiles (below: 64 x 64)
    int wgrad_split_wgs128 = 384, wgrad_split_wgs64 = 512;   // workgroups the split-K factor of the two split kernels aims at
    int wgrad_wgs = 768;        // ... of the fp32-instruction kernel (3 workgroups of 49 KB LDS per CU; profiles/r02_t3_training_round2.txt)
    int dgrad_mt = 0;           // rows per tile of the data-gradient kernel: 0 = by row count (64 from 24576 rows), 32, 64
    int dgrad_tail = 1;         // 1: dpre of an edge list is consumed inside the second-layer data gradient (k_dgrad_tail)
    unsigned long long* dbg = nullptr;   // Work::dbg for the stamped diagnostic build of k_dgrad_tail (tools/tail_stamps.py)
    int wgrad_stream = 1;       // 1: weight / bias gradients on the handle's second stream, beside the chain of data gradients (cmdgen_train.hip)
};

struct EvalLaunch {             // everything one evaluation's launches need (host side)
    Layout lay; Work w; Dims d; SmallW sw;
    const LayerW* layers;       // host array [L]
    int edge_grid, coord_grid;  // workgroups of the persistent-style edge kernels (tiles are taken round-robin)
    int node_mt, edge_mt, coord_mt;   // rows per tile (64, 32 or 16) chosen per launch from the row counts
    int embed_mt = 16;                // k_embed's tile (its critical path is a phar tile: encoders, embedding, two projections)
    std::vector<hipEvent_t>* prof_events;  // when non-null: [3] vectors, (start, stop) event pairs of every msg / node / coord launch
    mutable hipEvent_t pe_start = nullptr, pe_stop = nullptr;   // the pair the next profiled launch carries (hipExtLaunchKernelGGL)
    int ablate;                 // timing-only builds of the edge kernel (cmdgen_time_edge_kernel); 0 in production
    int stop_block = -1, stop_stage = 0;   // cmdgen_debug_eval_prefix: stop after stage 1..3 of this block (-1: run everything)
    const TrainSave* save = nullptr;   // training forward: keep the activations (see TrainSave)
    PocketCache pcache{};       // conditional chains: pocket tiles of k_embed are an axpy from the cache
    int skip_count = 0;         // 1: the radius-graph count pass has run (fused step kernel); 2: both passes have (training)
    int split = 0;              // 1: tiles of >= 32 rows multiply on the bf16 matrix pipe (Eng<MT, true>)
    int split16 = 0;            // 1: 16-row node tiles too (Eng<16, true>; option "node16_split")
    int n_cus = 256;
    int edge_fullk = 0;         // 1: 32-row edge tiles of the sampler build all 256 columns at once (full-K planes; option "edge_fullk")
    int node16w = 1;            // 1: 16-row node tiles of H = 256 on eight waves (kernels_node16w.hip)
    int node64 = 0;             // 1: large batches run k_node as 64-row tiles with both images in LDS (kernels_node64.hip)
    int dead_skip = 0;          // 2: every block of a conditional evaluation skips tiles whose new h nobody reads (by hop level); 1: the last block only; 0: off (option "dead_skip")
    mutable int unit = -1;      // weight unit (GCL) of the launches being issued: block l, sub-layer s -> l * S + s; -1: the block index itself (S = 1)
    mutable int skip_pc = 0;    // 1: the unit is not the last GCL of its block - its node kernel projects no P_c | Q_c
    mutable int frag_launches = 0;   // tile launches of this evaluation that read the fp32 FRAGMENT packs (the generic k_edge_msg / k_node / k_edge_coord forms): the training step re-packs those only when one will run
    mutable int live_thr = 0;   // set around a block's launches when that applies: nodes within this many hops of a moving node are still read
    int save_half = 0;          // with `save`: W2 / W7 (.wh, .wh_dev) carry half packs re-made this step: the two edge kernels run their 32-row full-K half form
    int save_half16 = 0;        // with `save`: W3 / W4 / Wpq_c / Wpq_e (.wh16, .wh_dev) carry 16-row half packs re-made this step: k_node16w<true>
    int save_split = 0;         // with `save`: W2 / W7 carry split packs, the two edge kernels may use them (H = 256)
    int write_embed = 1;        // 1: pass 2 of the radius graph and k_embed share a launch (k_write_embed) where both fit (option "write_embed")
    int e128_wgs = 2;           // workgroups per CU of the 128-row edge kernels (kernels_edge128.hip)
    int e128_fused = 3;         // bit 0 / 1: the 128-row message / coordinate kernel runs its fused main loop (half engine; lists long enough for more than one tile per workgroup)
    int half_engine = 1;        // 1: split-engine kernels that have a HALF form (two fp16 pieces per operand, three MFMAs per product; cmdgen_split.h) use it
};
// weight unit of block l's launches (EvalLaunch::unit), and the has_next argument of its node kernel: bit 0 = another unit follows
// (its P | Q are projected), bits 1..29 = the dead-tile threshold of the plane tiles, bit 30 = no P_c | Q_c (EvalLaunch::skip_pc)
inline int unit_of(const EvalLaunch& a, int l) { return a.unit >= 0 ? a.unit : l; }
inline int unit_has_next(const EvalLaunch& a, int l) { return unit_of(a, l) + 1 < a.d.L * a.d.S ? 1 : 0; }
inline int node_flags(const EvalLaunch& a, int l) { return unit_has_next(a, l) | (a.live_thr << 1) | (a.skip_pc << 30); }


// ---------------------------------------------------------------------------------
// SiLU / sigmoid on the fast hardware path: v_exp_f32 (base 2) + v_rcp_f32, 5 VALU instructions.
// (__frcp_rn / a plain division expand to the 10-instruction IEEE sequence; fp32 MFMA does not
// co-execute with VALU on gfx950 - profiles/r01_c_mfma_valu_coexec.txt - so every VALU instruction
// in the tile kernels is paid in full.)  Error ~2 ulp, far inside the 2e-5 evaluation tolerance.
__device__ __forceinline__ float silu_f(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
__device__ __forceinline__ float sigmoid_f(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
// squared distance with a FIXED rounding sequence (explicit fma chain): the radius-graph count pass and write pass
// live in translation units built with different -ffp-contract settings and must agree on every pair at the cutoff
__device__ __forceinline__ float dist2(const float4& a, const float4& b) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
}

// ---------------------------------------------------------------------------------
// Tile GEMM: a workgroup owns MT rows (MT = 64, 32 or 16) and wave w owns 64 output columns.
//   MT = 64 / 32: v_mfma_f32_32x32x2_f32, MT/32 x 2 tiles of 32x32 per wave (64 cycles each)
//   MT = 16     : v_mfma_f32_16x16x4_f32, 1 x 4 tiles of 16x16 per wave (32 cycles each) - used
//                 when a launch has too few rows to fill 256 CUs with bigger tiles.
// Rows of A come from LDS (row stride lda floats, 16-byte aligned), weights stream from L2 in
// fragment order two k-blocks ahead of their use (pinned with sched_barrier so hipcc cannot
// sink the prefetches back down to their uses; it still places the counted s_waitcnt).
//
// 16x16x4 fragment order: Wp16[(nt*KB16 + kb)*64 + lane] = { W[o][k..k+3] },
//   o = 16*nt + (lane & 15), k = 16*kb + 4*(lane >> 4): step j pairs k-slot g = lane>>4 with
//   k = 16kb + 4g + j, and the A row is read with one ds_read_b128 at the same k.
// ---------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int MT> struct TileAcc;
template <> struct TileAcc<64> { f32x16 a[2][2]; };
template <> struct TileAcc<32> { f32x16 a[1][2]; };
template <> struct TileAcc<16> { f32x4v a[4]; };

template <int MT>
__device__ __forceinline__ void acc_zero(TileAcc<MT>& acc) {
    if constexpr (MT == 16) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.a[n][r] = 0.0f;
    } else {
#pragma unroll
        for (int m = 0; m < MT / 32; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc.a[m][n][r] = 0.0f;
    }
}

// C/D layouts (cdna guide section 3):
//   32x32: lane l, reg r -> row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31
//   16x16: lane l, reg r -> row = 4 * (l >> 4) + r,                      col = l & 15
template <int MT, class F>
__device__ __forceinline__ void acc_foreach(const TileAcc<MT>& acc, int wave, F f) {
    const int lane = threadIdx.x & 63;
    if constexpr (MT == 16) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) f(4 * (lane >> 4) + r, wave * 64 + n * 16 + (lane & 15), acc.a[n][r]);
    } else {
#pragma unroll
        for (int m = 0; m < MT / 32; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    f(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), wave * 64 + n * 32 + (lane & 31), acc.a[m][n][r]);
    }
}

// the same walk with the wave-local column-tile index n (0..3 at 16 rows, 0..1 otherwise), for epilogues that keep
// per-column vectors (biases) in registers: ColVec holds v[col] for the lane's columns, fetched at kernel start so that no
// epilogue waits for an L2 round trip of its own
template <int MT, class F>
__device__ __forceinline__ void acc_foreach_n(const TileAcc<MT>& acc, int wave, F f) {
    const int lane = threadIdx.x & 63;
    if constexpr (MT == 16) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) f(4 * (lane >> 4) + r, wave * 64 + n * 16 + (lane & 15), n, acc.a[n][r]);
    } else {
#pragma unroll
        for (int m = 0; m < MT / 32; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    f(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), wave * 64 + n * 32 + (lane & 31), n, acc.a[m][n][r]);
    }
}
template <int MT> struct ColVec { float v[MT == 16 ? 4 : 2]; };
template <int MT>
__device__ __forceinline__ ColVec<MT> col_load(const float* __restrict__ vec, int wave) {
    const int lane = threadIdx.x & 63;
    ColVec<MT> c;
    if constexpr (MT == 16) {
#pragma unroll
        for (int n = 0; n < 4; ++n) c.v[n] = vec[wave * 64 + n * 16 + (lane & 15)];
    } else {
#pragma unroll
        for (int n = 0; n < 2; ++n) c.v[n] = vec[wave * 64 + n * 32 + (lane & 31)];
    }
    return c;
}

#define CMDGEN_MFMA32(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, ACC, 0, 0, 0)
#define CMDGEN_MFMA16(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, ACC, 0, 0, 0)

// A wave's view of one packed weight matrix: base of its fragment stream and the stride between
// its n-tiles (float4 units).  kb0_8 / kb_total8 are in blocks of 8 k; cg = the wave's 64-column group.
struct FragPtr { const float4* p; unsigned ns; };
template <int MT>
__device__ __forceinline__ FragPtr frag_ptr(const WPack& W, int kb_total8, int kb0_8, int cg) {
    const int lane = threadIdx.x & 63;
    FragPtr f;
    if constexpr (MT == 16) {
        const int kbt = kb_total8 / 2;
        f.p = W.w16 + ((size_t)(4 * cg) * kbt + kb0_8 / 2) * 64 + lane;
        f.ns = (unsigned)kbt * 64u;
    } else {
        f.p = W.w32 + ((size_t)(2 * cg) * kb_total8 + kb0_8) * 64 + lane;
        f.ns = (unsigned)kb_total8 * 64u;
    }
    return f;
}
// First two k-blocks of B fragments of a GEMM, fetched ahead of it (by the previous GEMM's last
// iteration, or at kernel start) so that a GEMM never starts with an empty pipeline.
template <int MT> struct BCarry { float4 a[MT == 16 ? 4 : 2], b[MT == 16 ? 4 : 2]; };

template <int MT>
__device__ __forceinline__ void gemm_prefetch(const FragPtr& f, BCarry<MT>& c) {
    constexpr int NT = MT == 16 ? 4 : 2;
#pragma unroll
    for (int n = 0; n < NT; ++n) { c.a[n] = f.p[n * f.ns]; c.b[n] = f.p[n * f.ns + 64u]; }
}

// acc += A(lds) x W^T over KB8*8 k values.  Four register sets (A..D) hold the fragments of
// k-blocks kb..kb+3 and are refilled in place two blocks ahead of their use, so the pipeline needs
// no register-rotation moves (VALU work is paid in full next to fp32 MFMA on gfx950).  `carry`
// brings this GEMM's first two blocks and leaves with the first two blocks of `next`.
template <int MT, int KB8>
__device__ __forceinline__ void tile_gemm(const float* __restrict__ ldsA, int lda, const FragPtr cur,
                                          const FragPtr next, TileAcc<MT>& acc, BCarry<MT>& carry) {
    const int lane = threadIdx.x & 63;
    if constexpr (MT == 16) {
        constexpr int KB = KB8 / 2;                      // k-blocks of 16
        static_assert(KB >= 4 && KB % 4 == 0, "K must be a multiple of 64");
        const float* ap = ldsA + (lane & 15) * lda + (lane >> 4) * 4;
        float4 bC[4], bD[4], aA, aB, aC, aD;
#define LOADB16(DST, F, KBI) _Pragma("unroll") for (int n = 0; n < 4; ++n) DST[n] = F.p[n * F.ns + (unsigned)(KBI) * 64u];
#define LOADA16(DST, KBI) DST = *reinterpret_cast<const float4*>(ap + (KBI) * 16);
#define STEP16(AV, BV)                                                                        \
        _Pragma("unroll") for (int n = 0; n < 4; ++n) CMDGEN_MFMA16(acc.a[n], AV.x, BV[n].x);  \
        _Pragma("unroll") for (int n = 0; n < 4; ++n) CMDGEN_MFMA16(acc.a[n], AV.y, BV[n].y);  \
        _Pragma("unroll") for (int n = 0; n < 4; ++n) CMDGEN_MFMA16(acc.a[n], AV.z, BV[n].z);  \
        _Pragma("unroll") for (int n = 0; n < 4; ++n) CMDGEN_MFMA16(acc.a[n], AV.w, BV[n].w);
#define QUAD16(KB_, FN, KN4, KN5)                                                             \
            LOADB16(bC, cur, KB_ + 2) LOADA16(aC, KB_ + 2)                                    \
            __builtin_amdgcn_sched_barrier(0);                                                \
            STEP16(aA, carry.a)                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                \
            LOADB16(bD, cur, KB_ + 3) LOADA16(aD, KB_ + 3)                                    \
            __builtin_amdgcn_sched_barrier(0);                                                \
            STEP16(aB, carry.b)                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                \
            LOADB16(carry.a, FN, KN4)                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                \
            STEP16(aC, bC)                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                \
            LOADB16(carry.b, FN, KN5)                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                \
            STEP16(aD, bD)                                                                    \
            __builtin_amdgcn_sched_barrier(0);
        LOADA16(aA, 0) LOADA16(aB, 1)
#pragma unroll 1
        for (int kb = 0; kb < KB - 4; kb += 4) {
            QUAD16(kb, cur, kb + 4, kb + 5)
            LOADA16(aA, kb + 4) LOADA16(aB, kb + 5)
        }
        QUAD16(KB - 4, next, 0, 1)
#undef QUAD16
#undef STEP16
#undef LOADB16
#undef LOADA16
    } else {
        constexpr int NMT = MT / 32;
        constexpr int KB = KB8;
        static_assert(KB >= 4 && KB % 4 == 0, "K must be a multiple of 32");
        const float* a0p = ldsA + (lane & 31) * lda + (lane >> 5) * 4;
        float4 bC[2], bD[2];
        float4 aA[NMT], aB[NMT], aC[NMT], aD[NMT];
#define LOADB32(DST, F, KBI) DST[0] = F.p[(unsigned)(KBI) * 64u]; DST[1] = F.p[F.ns + (unsigned)(KBI) * 64u];
#define LOADA32(DST, KBI) _Pragma("unroll") for (int m = 0; m < NMT; ++m) DST[m] = *reinterpret_cast<const float4*>(a0p + m * 32 * lda + (KBI) * 8);
#define STEP32(AV, BV)                                                                        \
        _Pragma("unroll") for (int m = 0; m < NMT; ++m) { CMDGEN_MFMA32(acc.a[m][0], AV[m].x, BV[0].x); CMDGEN_MFMA32(acc.a[m][1], AV[m].x, BV[1].x); } \
        _Pragma("unroll") for (int m = 0; m < NMT; ++m) { CMDGEN_MFMA32(acc.a[m][0], AV[m].y, BV[0].y); CMDGEN_MFMA32(acc.a[m][1], AV[m].y, BV[1].y); } \
        _Pragma("unroll") for (int m = 0; m < NMT; ++m) { CMDGEN_MFMA32(acc.a[m][0], AV[m].z, BV[0].z); CMDGEN_MFMA32(acc.a[m][1], AV[m].z, BV[1].z); } \
        _Pragma("unroll") for (int m = 0; m < NMT; ++m) { CMDGEN_MFMA32(acc.a[m][0], AV[m].w, BV[0].w); CMDGEN_MFMA32(acc.a[m][1], AV[m].w, BV[1].w); }
#define QUAD32(KB_, FN, KN4, KN5)                                                             \
            LOADB32(bC, cur, KB_ + 2) LOADA32(aC, KB_ + 2)                                    \
            __builtin_amdgcn_sched_barrier(0);                                                \
            STEP32(aA, carry.a)                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                \
            LOADB32(bD, cur, KB_ + 3) LOADA32(aD, KB_ + 3)                                    \
            __builtin_amdgcn_sched_barrier(0);                                                \
            STEP32(aB, carry.b)                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                \
            LOADB32(carry.a, FN, KN4)                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                \
            STEP32(aC, bC)                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                \
            LOADB32(carry.b, FN, KN5)                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                \
            STEP32(aD, bD)                                                                    \
            __builtin_amdgcn_sched_barrier(0);
        LOADA32(aA, 0) LOADA32(aB, 1)
#pragma unroll 1
        for (int kb = 0; kb < KB - 4; kb += 4) {
            QUAD32(kb, cur, kb + 4, kb + 5)
            LOADA32(aA, kb + 4) LOADA32(aB, kb + 5)
        }
        QUAD32(KB - 4, next, 0, 1)
#undef QUAD32
#undef STEP32
#undef LOADB32
#undef LOADA32
    }
}

// ---------------------------------------------------------------------------------
// GEMM engine of a tile kernel.  SP = false: the exact fp32 MFMA path above.  SP = true (MT >= 32): the same tile, the
// same fp32 LDS image and the same accumulator layout, multiplied on the bf16 matrix pipe as six bf16 products per fp32
// product (cmdgen_split.h: fp32-accurate, ~2x the delivered matrix rate; the split of the A fragments runs on the VALU
// in the shadow of the MFMAs, the weights are stored pre-split).
// ---------------------------------------------------------------------------------
template <int MT, bool SP> struct Eng;
template <int MT> struct Eng<MT, false> {
    typedef FragPtr Frag;
    typedef BCarry<MT> Carry;
    static __device__ __forceinline__ Frag frag(const WPack& W, int kb_total8, int kb0_8, int cg) { return frag_ptr<MT>(W, kb_total8, kb0_8, cg); }
    static __device__ __forceinline__ void prefetch(const Frag& f, Carry& c) { gemm_prefetch<MT>(f, c); }
    template <int KB8>
    static __device__ __forceinline__ void gemm(const float* lds, int lda, const Frag cur, const Frag next, TileAcc<MT>& acc, Carry& c) {
        tile_gemm<MT, KB8>(lds, lda, cur, next, acc, c);
    }
};
template <> struct Eng<16, true> {     // 16-row tiles: v_mfma_f32_16x16x32_bf16, register split, fragments three k-blocks ahead
    typedef S16FragPtr Frag;
    typedef S16Carry Carry;
    static __device__ __forceinline__ Frag frag(const WPack& W, int kb_total8, int kb0_8, int cg) { return sfrag16_ptr(W.ws16, kb_total8 / 4, kb0_8 / 4, cg); }
    static __device__ __forceinline__ void prefetch(const Frag& f, Carry& c) { split16_prefetch(f, c); }
    template <int KB8>
    static __device__ __forceinline__ void gemm(const float* lds, int lda, const Frag cur, const Frag next, TileAcc<16>& acc, Carry& c) {
        tile_gemm_rsplit16<KB8 / 4>(lds, lda, cur, next, acc.a, c);
    }
};
template <int MT> struct Eng<MT, true> {
    static_assert(MT == 64 || MT == 32, "32x32 MFMA tiles (16-row tiles: the specialisation above)");
    typedef SFragPtr Frag;
    typedef SCarry Carry;
    static __device__ __forceinline__ Frag frag(const WPack& W, int kb_total8, int kb0_8, int cg) { return sfrag_ptr(W.ws, kb_total8 / 2, kb0_8 / 2, cg); }
    static __device__ __forceinline__ void prefetch(const Frag& f, Carry& c) { split_prefetch(f, c); }
    template <int KB8>
    static __device__ __forceinline__ void gemm(const float* lds, int lda, const Frag cur, const Frag next, TileAcc<MT>& acc, Carry& c) {
        tile_gemm_rsplit<MT, KB8 / 2>(lds, lda, cur, next, acc.a, c);
    }
};

// Philox4x32-10 (Salmon et al. 2011), counter-based: results depend only on (key, counter).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// four standard normals from one Philox call (Box-Muller)
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2,
                                               uint32_t c3, float (&z)[4]) {
    uint32_t r[4];
    philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float two_pi = 6.283185307179586f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float u1 = ((float)(r[2 * i] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0,1)
        const float u2 = ((float)(r[2 * i + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float rad = sqrtf(-2.0f * logf(u1));
        z[2 * i] = rad * cosf(two_pi * u2);
        z[2 * i + 1] = rad * sinf(two_pi * u2);
    }
}
