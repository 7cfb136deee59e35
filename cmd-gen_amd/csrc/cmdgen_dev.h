// cmdgen_dev.h - device-side types and helpers shared by the gfx950 kernels.
//
// Written for CDNA4 only: 64-lane wavefronts, v_mfma_f32_32x32x2_f32 (exact fp32,
// 64 cycles/SIMD), 160 KiB LDS per CU.  No CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CMDGEN_TILE 64          // rows (edges or nodes) per workgroup tile
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

struct LayerW {                 // device pointers to one EquivariantBlock's packed weights
    // GCL.edge_mlp (egnn_new.py:15-19): layer 0 split by input columns [h_row | h_col | radial | d0]
    const float4* Wpq_e;        // [2H out][H in]: rows 0..H-1 act on h_row (-> P), H..2H-1 on h_col (-> Q)
    const float*  b1;           // folded into P
    const float*  wr_e;         // column 2H   (radial)
    const float*  wd_e;         // column 2H+1 (d0)
    const float4* W2;           // edge_mlp.2 [H][H]
    const float*  b2;
    const float*  wa;           // att_mlp.0 weight [H]
    float         ba;
    // GCL.node_mlp (egnn_new.py:21-24)
    const float4* W3;           // node_mlp.0 [H][2H]  (in = [h | agg])
    const float*  b3;
    const float4* W4;           // node_mlp.2 [H][H]
    const float*  b4;
    // EquivariantUpdate.coord_mlp (egnn_new.py:78-83)
    const float4* Wpq_c;        // coord_mlp.0 split like Wpq_e
    const float*  b6;
    const float*  wr_c;
    const float*  wd_c;
    const float4* W7;           // coord_mlp.2 [H][H]
    const float*  b7;
    const float*  w5;           // coord_mlp.4 weight [H], no bias
};

struct SmallW {                 // encoders / decoders / embeddings, plain [out][in] fp32 on device
    const float *pe0_w, *pe0_b, *pe2_w, *pe2_b;     // phar_encoder   (dynamics.py:21-25)
    const float *pd0_w, *pd0_b, *pd2_w, *pd2_b;     // phar_decoder   (:27-31)
    const float *re0_w, *re0_b, *re2_w, *re2_b;     // residue_encoder (:33-37)
    const float *rd0_w, *rd0_b, *rd2_w, *rd2_b;     // residue_decoder (:39-43)
    const float *emb_w, *emb_b;                     // egnn.embedding      [H][J+1]
    const float *embo_wT, *embo_b;                  // egnn.embedding_out, stored transposed [H][J+1]
};

struct Dims {
    int P, R, J, H, L;          // phar_nf, residue_nf, joint_nf, hidden_nf, n_layers
    int dyn;                    // J + condition_time
    int attention, use_tanh, condition_time;
    float cutoff2;              // cutoff^2, < 0: no cutoff
    float norm_constant, norm_factor, coords_range;
    float norm_x, norm_h, bias_h;
};

struct Layout {                 // one flat batch; all pointers device
    int B, Nl, Np, N;           // samples, phar nodes, pocket nodes, total
    int max_n;                  // max nodes of one sample
    const int* num_phar;        // [B]
    const int* num_pocket;      // [B]
    const int* phar_base;       // [B] exclusive prefix of num_phar
    const int* pocket_base;     // [B]
    const int* node_sample;     // [N] sample index of each node in flat order (phar first)
    const int64_t* pocket_gid;  // [B] global pocket ids (Philox key)
};

struct Work {                   // per-layout workspace; all pointers device
    float4* X0;                 // [Nl] phar input coordinates of this evaluation
    float4* XP;                 // [Np] pocket coordinates
    float4* XL;                 // [L][Nl] phar coordinates entering block l (l >= 1 materialised by the node kernel)
    float4* ACC;                // [L][Nl] sum of coordinate messages of block l (zeroed per evaluation)
    float*  h;                  // [N][H]
    float*  P;                  // [N][H]  edge-MLP layer-0 receiver part (+b1)
    float*  Q;                  // [N][H]  sender part
    float*  Pc;                 // [N][H]  coord-MLP receiver part (+b6), only phar rows read
    float*  Qc;                 // [N][H]
    float*  agg;                // [N][H]  zero between blocks
    int*    degL;               // [N] degree in sample-local order
    int*    pocketE;            // [B] edges per sample
    int*    pocketEph;          // [B] edges with phar receiver per sample
    int*    erow; int* ecol; float* ed0;        // [Ecap] compact edge list sorted by flat (row, col); the first
                                                //        totals[1] entries are the phar-receiver edges
    int*    totals;             // [0]=E, [1]=Ec of the current evaluation
    unsigned long long* counters;   // cmdgen_counters
    int*    nan_flag;           // [1] set by readout when any velocity is NaN
    float*  eps_tmp;            // [Nl][3+P] evaluation output used by the chain
};

struct ChainState {             // device-resident denoising-loop state
    int step;                   // index into coef[], incremented by the first kernel of each evaluation
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
    unsigned int* check;        // [K+3][2] float bits: max|x|, max|sum x| per check point
    ChainState* state;
};

struct EvalLaunch {             // everything one evaluation's launches need (host side)
    Layout lay; Work w; Dims d; SmallW sw;
    const LayerW* layers;       // host array [L]
    int edge_grid;              // workgroups of the persistent-style edge kernels
    std::vector<hipEvent_t>* msg_events;   // when non-null: event pair around every edge-message launch
};

// ---------------------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float v) {
    // v * sigmoid(v); v_exp_f32 + v_rcp_f32 path (about 2 ulp)
    return v * __frcp_rn(1.0f + __expf(-v));
}
__device__ __forceinline__ float sigmoid_f(float v) {
    return __frcp_rn(1.0f + __expf(-v));
}
__device__ __forceinline__ float dist2(const float4& a, const float4& b) {
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
}

// 64x64 output block per wave: acc[mt][nt] are 32x32 tiles; rows of A come from LDS
// (row stride lda floats, 16-byte aligned rows), columns from the packed weight Wp.
// KB k-blocks of 8 starting at kb0 of a matrix with kb_total k-blocks; n-tiles nt0, nt0+1.
// Weight fragments are fetched two k-blocks ahead straight from L2 into registers.
template <int KB>
__device__ __forceinline__ void mfma_tile_64x64(const float* __restrict__ ldsA, int lda,
                                                const float4* __restrict__ Wp, int kb_total,
                                                int kb0, int nt0, f32x16 (&acc)[2][2]) {
    const int lane = threadIdx.x & 63;
    const float* a0p = ldsA + (lane & 31) * lda + (lane >> 5) * 4;
    const float* a1p = a0p + 32 * lda;
    const float4* b0p = Wp + ((size_t)nt0 * kb_total + kb0) * 64 + lane;
    const float4* b1p = Wp + ((size_t)(nt0 + 1) * kb_total + kb0) * 64 + lane;

    float4 bA0 = b0p[0], bA1 = b1p[0];
    float4 bB0 = b0p[64], bB1 = b1p[64];
    float4 aA0 = *reinterpret_cast<const float4*>(a0p);
    float4 aA1 = *reinterpret_cast<const float4*>(a1p);

#define CMDGEN_MFMA4(A0, A1, B0, B1)                                                         \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x, B0.x, acc[0][0], 0, 0, 0);        \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x, B1.x, acc[0][1], 0, 0, 0);        \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x, B0.x, acc[1][0], 0, 0, 0);        \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x, B1.x, acc[1][1], 0, 0, 0);        \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.y, B0.y, acc[0][0], 0, 0, 0);        \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.y, B1.y, acc[0][1], 0, 0, 0);        \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.y, B0.y, acc[1][0], 0, 0, 0);        \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.y, B1.y, acc[1][1], 0, 0, 0);        \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.z, B0.z, acc[0][0], 0, 0, 0);        \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.z, B1.z, acc[0][1], 0, 0, 0);        \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.z, B0.z, acc[1][0], 0, 0, 0);        \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.z, B1.z, acc[1][1], 0, 0, 0);        \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.w, B0.w, acc[0][0], 0, 0, 0);        \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.w, B1.w, acc[0][1], 0, 0, 0);        \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.w, B0.w, acc[1][0], 0, 0, 0);        \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.w, B1.w, acc[1][1], 0, 0, 0);

    static_assert(KB % 2 == 0, "KB must be even");
    // Software pipeline, pinned with sched_barrier so hipcc cannot sink the prefetches back
    // down to their uses: while the 16 MFMAs (1024 cycles) of block kb run, the weight
    // fragments of block kb+2 (L2, ~500-900 cycles) and the A rows of block kb+1 (LDS) are
    // in flight.  The compiler still places the counted s_waitcnt at the first use.
#pragma unroll 1
    for (int kb = 0; kb < KB; kb += 2) {
        const int k2 = (kb + 2 < KB) ? kb + 2 : KB - 1;     // clamped: the tail re-reads valid memory
        const int k3 = (kb + 3 < KB) ? kb + 3 : KB - 1;
        float4 bC0 = b0p[k2 * 64], bC1 = b1p[k2 * 64];
        float4 aB0 = *reinterpret_cast<const float4*>(a0p + (kb + 1) * 8);
        float4 aB1 = *reinterpret_cast<const float4*>(a1p + (kb + 1) * 8);
        __builtin_amdgcn_sched_barrier(0);
        CMDGEN_MFMA4(aA0, aA1, bA0, bA1)
        __builtin_amdgcn_sched_barrier(0);
        float4 bD0 = b0p[k3 * 64], bD1 = b1p[k3 * 64];
        aA0 = *reinterpret_cast<const float4*>(a0p + k2 * 8);
        aA1 = *reinterpret_cast<const float4*>(a1p + k2 * 8);
        __builtin_amdgcn_sched_barrier(0);
        CMDGEN_MFMA4(aB0, aB1, bB0, bB1)
        __builtin_amdgcn_sched_barrier(0);
        bA0 = bC0; bA1 = bC1; bB0 = bD0; bB1 = bD1;
    }
#undef CMDGEN_MFMA4
}

__device__ __forceinline__ void acc_zero(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
}

// C/D layout of the 32x32 MFMA: lane l, register r hold
//   row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5),  col = l & 31   (cdna guide section 3)
#define CMDGEN_ACC_FOREACH(wave, BODY)                                                      \
    {                                                                                        \
        const int _lane = threadIdx.x & 63;                                                  \
        _Pragma("unroll") for (int _m = 0; _m < 2; ++_m)                                     \
        _Pragma("unroll") for (int _n = 0; _n < 2; ++_n)                                     \
        _Pragma("unroll") for (int _r = 0; _r < 16; ++_r) {                                  \
            const int row = _m * 32 + (_r & 3) + 8 * (_r >> 2) + 4 * (_lane >> 5);           \
            const int col = (wave) * 64 + _n * 32 + (_lane & 31);                            \
            const float v = acc[_m][_n][_r];                                                    \
            BODY                                                                             \
        }                                                                                    \
    }

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
