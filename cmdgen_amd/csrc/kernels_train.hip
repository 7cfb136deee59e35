// kernels_train.hip - device kernels of the TRAINING step other than its forward pass (which is the
// sampler's fused evaluation in kernels_egnn.hip, instantiated with save hooks): the loss side (noising, loss terms,
// dL/d eps), the backward pass - data gradients on the split-bf16 engine (k_dgrad_split; k_dgrad_tail with the whole
// tail pass of an edge list inside), grouped weight gradients (k_wgrad_group, k_wgrad_split), the gate / head adjoints
// with their parameter reductions, the stand-alone tail pass and the generic exact-fp32 GEMM of the fp32-instruction
// mode and of the small encoder / decoder products -, the per-step re-packing of the parameters, and AdamW.
// the data-gradient kernels keep the plane GEMM's loads in a burst per k-block: pinned one by one under the MFMAs (the sampler's edge
// kernels: -6 % per launch) k_dgrad_split<32,3> takes 1.23 instead of 0.95 ms per three steps (profiles/r03_n_wgrad.txt)
#define CMDGEN_PLANE_PIN 0
#include "cmdgen_dev.h"
#include "cmdgen_split.h"

// ------------------------------------------------------------------------------------
// C[M,N] (+)= alpha * op(A)[M,K] * op(B)[K,N] (+ bias[N])          exact fp32 on v_mfma_f32_32x32x2_f32
//   TA = false: A stored [M][K] (lda)      TA = true: A stored [K][M] (lda)     (wgrad: A = dY^T)
//   TB = true : B stored [N][K] (ldb) - the nn.Linear weight layout (forward)
//   TB = false: B stored [K][N] (ldb) - dgrad (dX = dY W) and wgrad (dW = dY^T X)
// 64x64 tile per workgroup (4 waves, 32x32 each), K tile 16, both operands staged through LDS as
// [row][k] so that one ds_read_b128 per lane feeds four MFMA k-steps (same pairing as the sampler's
// tiles).  blockIdx.z splits K (wgrad over tens of thousands of edges); split results are combined with
// float atomics.  All loads are bounds-checked scalars: operands are arbitrary sub-blocks (column slices of
// edge_mlp.0, feature columns of xh) with arbitrary leading dimensions.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_exact(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float dsilu(float v) {            // d/dv v*sigmoid(v) = s * (1 + v * (1 - s))
    const float s = sigmoid_f(v);        // v_exp_f32 + v_rcp_f32 (2 ulp), the forward's own sigmoid: expf + an IEEE division
                                         // cost ~25 instructions per element in every dgrad epilogue
    return s * (1.0f + v * (1.0f - s));
}

// k extent KT of one staged tile is a template parameter: 64 for launches with few workgroups (node-level products: one
// workgroup per CU, so fewer, fatter barrier-separated iterations and the register prefetch below are what hides the
// global latency), 32 for big launches (edge-level products: 18 KB of LDS per workgroup keeps 8 of them on a CU, which
// hides it better - 61 vs 54 TF/s on [25600,256]x[256,256]).
#define TG_LD(KT) ((KT) + 4)
#define TG_P(KT) ((KT) / 16)     // float4 fetches per thread and operand tile: 64 rows x KT k over 256 threads

// One 64 x KT operand tile: fetch() brings this thread's TG_P float4 pieces into registers (so the NEXT tile's
// global loads are in flight while the current tile's MFMAs run - a small GEMM such as a node-level [3.8k,256]x[256,256]
// product has one workgroup per CU and nothing else to hide that latency behind), put_*() writes them to LDS as [row][k].
//   operand stored [row][k] (k contiguous): thread -> (row = tid/16 + 16*p, k4 = (tid%16)*4)
//   operand stored [k][row] (row contiguous): thread -> (k = tid/16 + 16*p, row4 = (tid%16)*4)
// VEC: every 4-element group a thread loads is 16-byte aligned and entirely in range (host-checked: base pointers,
// leading dimensions, the contiguous extent and the k chunking are multiples of 4) -> one global_load_dwordx4.
template <int KT, bool TRANS, bool VEC>
__device__ __forceinline__ void tg_fetch(float4 (&v)[TG_P(KT)], const float* __restrict__ G, int ld, int r0, int R, int k0, int k_end) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < TG_P(KT); ++p) {
        v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!TRANS) {
            constexpr int TPR = KT / 4, RPP = 256 / TPR;      // threads per row, rows per pass
            const int gr = r0 + tid / TPR + RPP * p, gk = k0 + (tid % TPR) * 4;
            if (gr < R) {
                const float* src = G + (size_t)gr * ld + gk;
                if (VEC && gk + 3 < k_end) v[p] = *reinterpret_cast<const float4*>(src);
                else {
                    if (gk < k_end) v[p].x = src[0];
                    if (gk + 1 < k_end) v[p].y = src[1];
                    if (gk + 2 < k_end) v[p].z = src[2];
                    if (gk + 3 < k_end) v[p].w = src[3];
                }
            }
        } else {
            const int gk = k0 + (tid >> 4) + 16 * p, gr = r0 + (tid & 15) * 4;
            if (gk < k_end) {
                const float* src = G + (size_t)gk * ld + gr;
                if (VEC && gr + 3 < R) v[p] = *reinterpret_cast<const float4*>(src);
                else {
                    if (gr < R) v[p].x = src[0];
                    if (gr + 1 < R) v[p].y = src[1];
                    if (gr + 2 < R) v[p].z = src[2];
                    if (gr + 3 < R) v[p].w = src[3];
                }
            }
        }
    }
}
template <int KT, bool TRANS>
__device__ __forceinline__ void tg_put_f32(float* S, const float4 (&v)[TG_P(KT)]) {
    const int tid = threadIdx.x;
    constexpr int TPR = KT / 4, RPP = 256 / TPR;
#pragma unroll
    for (int p = 0; p < TG_P(KT); ++p) {
        if (!TRANS) *reinterpret_cast<float4*>(S + (tid / TPR + RPP * p) * TG_LD(KT) + (tid % TPR) * 4) = v[p];
        else {
            const int k = (tid >> 4) + 16 * p, rq = (tid & 15) * 4;
            S[(rq + 0) * TG_LD(KT) + k] = v[p].x; S[(rq + 1) * TG_LD(KT) + k] = v[p].y;
            S[(rq + 2) * TG_LD(KT) + k] = v[p].z; S[(rq + 3) * TG_LD(KT) + k] = v[p].w;
        }
    }
}

// shared epilogue: C (+)= alpha * acc + bias, split-K partials by atomics, the two SiLU epilogues
__device__ __forceinline__ void tg_epilogue(const f32x16& acc, int M, int N, float* __restrict__ C, int ldc,
                                            const float* __restrict__ bias, float alpha, int accumulate, int epi,
                                            float* __restrict__ aux, int ldaux) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int gn = n0 + wn + (lane & 31);
    if (gn >= N) return;
    const float bv = (bias && blockIdx.z == 0) ? bias[gn] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gm >= M) continue;
        float* c = C + (size_t)gm * ldc + gn;
        float v = alpha * acc[r] + bv;
        if (gridDim.z > 1) { atomicAdd(c, v); continue; }
        if (accumulate) v += *c;
        // epilogues (never with split-K): 1 = also write SiLU(v) to aux (forward: pre-activation and activation in one
        // pass); 2 = multiply by SiLU'(aux) (dgrad through the activation that produced this operand)
        if (epi == 2) v *= dsilu(aux[(size_t)gm * ldaux + gn]);
        *c = v;
        if (epi == 1) aux[(size_t)gm * ldaux + gn] = silu_exact(v);
    }
}

template <int KT, bool TA, bool TB, bool VECA, bool VECB>
__global__ __launch_bounds__(256) void k_sgemm(int M, int N, int K, const float* __restrict__ A, int lda,
                                               const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                               const float* __restrict__ bias, float alpha, int accumulate, int kchunk,
                                               int epi, float* __restrict__ aux, int ldaux) {
    __shared__ __attribute__((aligned(16))) float As[64 * TG_LD(KT)];
    __shared__ __attribute__((aligned(16))) float Bs[64 * TG_LD(KT)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int k_begin = blockIdx.z * kchunk, k_end = min(K, k_begin + kchunk);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 ra[TG_P(KT)], rb[TG_P(KT)];
    tg_fetch<KT, TA, VECA>(ra, A, lda, m0, M, k_begin, k_end);
    tg_fetch<KT, !TB, VECB>(rb, B, ldb, n0, N, k_begin, k_end);
    for (int k0 = k_begin; k0 < k_end; k0 += KT) {
        tg_put_f32<KT, TA>(As, ra);
        tg_put_f32<KT, !TB>(Bs, rb);
        __syncthreads();
        if (k0 + KT < k_end) {                        // next tile's loads fly during this tile's MFMAs
            tg_fetch<KT, TA, VECA>(ra, A, lda, m0, M, k0 + KT, k_end);
            tg_fetch<KT, !TB, VECB>(rb, B, ldb, n0, N, k0 + KT, k_end);
        }
#pragma unroll
        for (int kb = 0; kb < KT / 8; ++kb) {
            const float4 a = *reinterpret_cast<const float4*>(As + (wm + (lane & 31)) * TG_LD(KT) + kb * 8 + 4 * (lane >> 5));
            const float4 b = *reinterpret_cast<const float4*>(Bs + (wn + (lane & 31)) * TG_LD(KT) + kb * 8 + 4 * (lane >> 5));
            CMDGEN_MFMA32(acc, a.x, b.x);
            CMDGEN_MFMA32(acc, a.y, b.y);
            CMDGEN_MFMA32(acc, a.z, b.z);
            CMDGEN_MFMA32(acc, a.w, b.w);
        }
        __syncthreads();
    }
    tg_epilogue(acc, M, N, C, ldc, bias, alpha, accumulate, epi, aux, ldaux);
}

// The same GEMM with bf16 OPERANDS and fp32 accumulation (v_mfma_f32_32x32x16_bf16: 16x the fp32 matrix rate) - the
// opt-in mixed-precision policy of the training step (master weights, optimizer state, activations in HBM and all
// elementwise math stay fp32; operands are rounded to nearest-even bf16 while they are staged into LDS).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define TGH_LD(KT) ((KT) + 8)   // halfs per LDS row: keeps the 16-byte operand reads aligned

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);     // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
template <int KT, bool TRANS>
__device__ __forceinline__ void tg_put_bf16(unsigned short* S, const float4 (&v)[TG_P(KT)]) {
    const int tid = threadIdx.x;
    constexpr int TPR = KT / 4, RPP = 256 / TPR;
#pragma unroll
    for (int p = 0; p < TG_P(KT); ++p) {
        if (!TRANS) {
            uint2 pk;
            pk.x = (unsigned)f2bf(v[p].x) | ((unsigned)f2bf(v[p].y) << 16);
            pk.y = (unsigned)f2bf(v[p].z) | ((unsigned)f2bf(v[p].w) << 16);
            *reinterpret_cast<uint2*>(S + (tid / TPR + RPP * p) * TGH_LD(KT) + (tid % TPR) * 4) = pk;
        } else {
            const int k = (tid >> 4) + 16 * p, rq = (tid & 15) * 4;
            S[(rq + 0) * TGH_LD(KT) + k] = f2bf(v[p].x); S[(rq + 1) * TGH_LD(KT) + k] = f2bf(v[p].y);
            S[(rq + 2) * TGH_LD(KT) + k] = f2bf(v[p].z); S[(rq + 3) * TGH_LD(KT) + k] = f2bf(v[p].w);
        }
    }
}

template <int KT, bool TA, bool TB, bool VECA, bool VECB>
__global__ __launch_bounds__(256) void k_sgemm_bf16(int M, int N, int K, const float* __restrict__ A, int lda,
                                                    const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                    const float* __restrict__ bias, float alpha, int accumulate, int kchunk,
                                                    int epi, float* __restrict__ aux, int ldaux) {
    __shared__ __attribute__((aligned(16))) unsigned short As[64 * TGH_LD(KT)];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[64 * TGH_LD(KT)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int k_begin = blockIdx.z * kchunk, k_end = min(K, k_begin + kchunk);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 ra[TG_P(KT)], rb[TG_P(KT)];
    tg_fetch<KT, TA, VECA>(ra, A, lda, m0, M, k_begin, k_end);
    tg_fetch<KT, !TB, VECB>(rb, B, ldb, n0, N, k_begin, k_end);
    for (int k0 = k_begin; k0 < k_end; k0 += KT) {
        tg_put_bf16<KT, TA>(As, ra);
        tg_put_bf16<KT, !TB>(Bs, rb);
        __syncthreads();
        if (k0 + KT < k_end) {
            tg_fetch<KT, TA, VECA>(ra, A, lda, m0, M, k0 + KT, k_end);
            tg_fetch<KT, !TB, VECB>(rb, B, ldb, n0, N, k0 + KT, k_end);
        }
#pragma unroll
        for (int ks = 0; ks < KT / 16; ++ks) {      // lanes 0-31 supply k 0..7, lanes 32-63 k 8..15 of the 16-k step
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + (wm + (lane & 31)) * TGH_LD(KT) + ks * 16 + 8 * (lane >> 5));
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + (wn + (lane & 31)) * TGH_LD(KT) + ks * 16 + 8 * (lane >> 5));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    tg_epilogue(acc, M, N, C, ldc, bias, alpha, accumulate, epi, aux, ldaux);
}

// ------------------------------------------------------------------------------------
// Grouped weight gradients: up to 8 products dW_p (+)= dY_p^T X_p (and, optionally, db_p += column sums of dY_p) in ONE
// launch.  The node-level weight gradients of a block are seven [256 x 256] products over K = #nodes: each alone is a
// 16-tile launch that needs split-K to fill the chip and still costs ~23 us of launch + atomic latency; together they
// are one launch of the same duration.  Operands: dY_p [K][M_p] (lddy), X_p [K][N_p] (ldx), both 16-byte aligned with
// leading dimensions that are multiples of 4; results are added with float atomics (the destination holds the running
// gradient).  blockIdx.z = problem * zsplit + k-split.
// ------------------------------------------------------------------------------------
struct WgradBatch {
    const float* dy[8]; const float* x[8]; float* dw[8]; float* db[8];
    int M[8], N[8], lddy[8], ldx[8], ldw[8];
    int n;
    int xs[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // 1: X_p holds PRE-activations - SiLU is applied while the operand is staged (the forward then stores pre1 / pre6 only)
};
__device__ __forceinline__ float4 silu4(const float4& v) { return make_float4(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w)); }

template <bool BF, bool XS = false>      // XS: some X_p holds pre-activations (WgradBatch::xs) - a separate instantiation: the default one carries no trace of it
__global__ __launch_bounds__(256) void k_wgrad_group(WgradBatch g, int K, int kchunk, int zsplit) {
    constexpr int KT = 64;
    // fp32 path: both operands lie in memory k-major ([k][channel]) and stay that way in LDS ([k][WG_LDK]): v_mfma_f32_32x32x2
    // takes ONE value per lane (row = lane % 32, k = lane / 32), so a wave reads 32 consecutive channels of two k rows per
    // MFMA straight from the staged rows - no transposition on the way in (the [channel][k] image needed scalar, bank-
    // conflicting LDS stores: 32 per thread and k step).  WG_LDK = 96: the two k rows of a read fall on disjoint banks.
    constexpr int WG_LDK = 96;
    __shared__ __attribute__((aligned(16))) float smem[2 * KT * WG_LDK];
    float* As = smem; float* Bs = smem + KT * WG_LDK;
    unsigned short* Ah = reinterpret_cast<unsigned short*>(smem);
    unsigned short* Bh = Ah + 64 * TGH_LD(KT);
    const int p = blockIdx.z / zsplit, kz = blockIdx.z - p * zsplit;
    const int M = g.M[p], N = g.N[p];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    if (m0 >= M || n0 >= N) return;
    const int k_begin = kz * kchunk, k_end = min(K, k_begin + kchunk);
    if (k_begin >= k_end) return;
    const float* __restrict__ A = g.dy[p]; const float* __restrict__ B = g.x[p];
    const int lda = g.lddy[p], ldb = g.ldx[p];
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const bool want_bias = g.db[p] != nullptr && n0 == 0;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 ra[TG_P(KT)], rb[TG_P(KT)];
    float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);      // this thread's rows m0 + (tid&15)*4 .. +3, its k slots
    tg_fetch<KT, true, true>(ra, A, lda, m0, M, k_begin, k_end);
    tg_fetch<KT, true, true>(rb, B, ldb, n0, N, k_begin, k_end);
    for (int k0 = k_begin; k0 < k_end; k0 += KT) {
        if (want_bias) {
#pragma unroll
            for (int q = 0; q < TG_P(KT); ++q) { colsum.x += ra[q].x; colsum.y += ra[q].y; colsum.z += ra[q].z; colsum.w += ra[q].w; }
        }
        if constexpr (XS) if (g.xs[p]) {
#pragma unroll
            for (int q = 0; q < TG_P(KT); ++q) rb[q] = silu4(rb[q]);
        }
        if (BF) { tg_put_bf16<KT, true>(Ah, ra); tg_put_bf16<KT, true>(Bh, rb); }
        else {
#pragma unroll
            for (int q = 0; q < TG_P(KT); ++q) {        // thread -> (k = tid / 16 + 16 q, channels (tid % 16) * 4 .. + 3), as fetched
                *reinterpret_cast<float4*>(As + ((tid >> 4) + 16 * q) * WG_LDK + (tid & 15) * 4) = ra[q];
                *reinterpret_cast<float4*>(Bs + ((tid >> 4) + 16 * q) * WG_LDK + (tid & 15) * 4) = rb[q];
            }
        }
        __syncthreads();
        if (k0 + KT < k_end) {
            tg_fetch<KT, true, true>(ra, A, lda, m0, M, k0 + KT, k_end);
            tg_fetch<KT, true, true>(rb, B, ldb, n0, N, k0 + KT, k_end);
        }
        if (BF) {
#pragma unroll
            for (int ks = 0; ks < KT / 16; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ah + (wm + (lane & 31)) * TGH_LD(KT) + ks * 16 + 8 * (lane >> 5));
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bh + (wn + (lane & 31)) * TGH_LD(KT) + ks * 16 + 8 * (lane >> 5));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
        } else {
            const float* ap = As + (lane >> 5) * WG_LDK + wm + (lane & 31);
            const float* bp = Bs + (lane >> 5) * WG_LDK + wn + (lane & 31);
#pragma unroll
            for (int k2 = 0; k2 < KT / 2; ++k2) CMDGEN_MFMA32(acc, ap[k2 * 2 * WG_LDK], bp[k2 * 2 * WG_LDK]);
        }
        __syncthreads();
    }
    float* C = g.dw[p];
    const int ldc = g.ldw[p];
    const int gn = n0 + wn + (lane & 31);
    if (gn < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (gm < M) atomicAdd(C + (size_t)gm * ldc + gn, acc[r]);
        }
    }
    if (want_bias) {            // 16 threads (tid >> 4) hold partial sums of the same four rows: combine through LDS
        float* red = smem;       // [16][64]
        const int rq = (tid & 15) * 4, slot = tid >> 4;
        red[slot * 64 + rq + 0] = colsum.x; red[slot * 64 + rq + 1] = colsum.y;
        red[slot * 64 + rq + 2] = colsum.z; red[slot * 64 + rq + 3] = colsum.w;
        __syncthreads();
        if (tid < 64 && m0 + tid < M) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += red[q * 64 + tid];
            atomicAdd(g.db[p] + m0 + tid, sum);
        }
    }
}

// ------------------------------------------------------------------------------------
// The same grouped products on the split-bf16 engine (cmdgen_split.h; every M_p, N_p a multiple of 64, operands 16-byte
// aligned): the fp32 instruction needs 512 matrix-pipe cycles per 16 k values of a 32 x 32 tile, six bf16 MFMAs need 192.
// BOTH operands are activations here, so both are split while they are staged: a thread loads eight consecutive k rows of
// four channels (8 x 16 bytes), splits the eight k values of each channel into the three bf16 pieces in registers (the
// fragment layout of v_mfma_f32_32x32x16_bf16 wants eight consecutive k per lane: the k-major memory layout is transposed
// in registers, not with scalar LDS stores) and writes one 16-byte piece per channel and plane: planes [channel][64 + 8].
// Threads 0-127 stage dY, 128-255 stage X; the next k tile's rows are in flight during the MFMAs.  NPC = 1: leading pieces
// only (bf16 operands).
// ------------------------------------------------------------------------------------
template <int NPC, bool XS = false>
__global__ __launch_bounds__(256, 2) void k_wgrad_split(WgradBatch g, int K, int kchunk, int zsplit) {
    constexpr int KT = 64, PLD = KT + 8, PE = 64 * PLD;               // bf16 elements per plane row / per plane
    __shared__ __attribute__((aligned(16))) unsigned short planes[2 * NPC * PE];
    const int p = blockIdx.z / zsplit, kz = blockIdx.z - p * zsplit;
    const int M = g.M[p], N = g.N[p];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    if (m0 >= M || n0 >= N) return;
    const int k_begin = kz * kchunk, k_end = min(K, k_begin + kchunk);
    if (k_begin >= k_end) return;
    const bool isB = tid >= 128;
    const int t = tid & 127, cg = t & 15, kb = t >> 4;                 // channels 4 cg .. 4 cg + 3, k rows 8 kb .. 8 kb + 7 of the tile
    const float* __restrict__ G = isB ? g.x[p] : g.dy[p];
    const int ld = isB ? g.ldx[p] : g.lddy[p];
    const float* gsrc = G + (isB ? n0 : m0) + 4 * cg;
    unsigned short* mine = planes + (isB ? NPC * PE : 0);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const bool want_bias = g.db[p] != nullptr && n0 == 0 && !isB;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 v[8];
    float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + 8 * kb + i;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < k_end) v[i] = *reinterpret_cast<const float4*>(gsrc + (size_t)k * ld);
        }
    };
    fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += KT) {
        if (want_bias) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { colsum.x += v[i].x; colsum.y += v[i].y; colsum.z += v[i].z; colsum.w += v[i].w; }
        }
        {   // transpose in registers: channel c gets the eight k values v[0..7].c, split into the pieces, one 16-byte store per plane
            if constexpr (XS) if (isB && g.xs[p]) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = silu4(v[i]);
            }
            const float c0[8] = {v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x};
            const float c1[8] = {v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y};
            const float c2[8] = {v[0].z, v[1].z, v[2].z, v[3].z, v[4].z, v[5].z, v[6].z, v[7].z};
            const float c3[8] = {v[0].w, v[1].w, v[2].w, v[3].w, v[4].w, v[5].w, v[6].w, v[7].w};
            const float* cc[4] = {c0, c1, c2, c3};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned short* dst = mine + (4 * cg + c) * PLD + 8 * kb;
                if constexpr (NPC == 3) {
                    sbf16x8 p0, p1, p2;
                    split8(make_float4(cc[c][0], cc[c][1], cc[c][2], cc[c][3]), make_float4(cc[c][4], cc[c][5], cc[c][6], cc[c][7]), p0, p1, p2);
                    *reinterpret_cast<sbf16x8*>(dst) = p0; *reinterpret_cast<sbf16x8*>(dst + PE) = p1; *reinterpret_cast<sbf16x8*>(dst + 2 * PE) = p2;
                } else {
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    const u4 q = {cvt_pk_bf16(cc[c][0], cc[c][1]), cvt_pk_bf16(cc[c][2], cc[c][3]), cvt_pk_bf16(cc[c][4], cc[c][5]), cvt_pk_bf16(cc[c][6], cc[c][7])};
                    *reinterpret_cast<u4*>(dst) = q;
                }
            }
        }
        __syncthreads();
        if (k0 + KT < k_end) fetch(k0 + KT);
        const unsigned short* ap = planes + (wm + (lane & 31)) * PLD + 8 * (lane >> 5);
        const unsigned short* bp = planes + NPC * PE + (wn + (lane & 31)) * PLD + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KT / 16; ++ks) {
            sbf16x8 a[NPC], b[NPC];
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
                a[q] = *reinterpret_cast<const sbf16x8*>(ap + q * PE + ks * 16);
                b[q] = *reinterpret_cast<const sbf16x8*>(bp + q * PE + ks * 16);
            }
            if constexpr (NPC == 3) {       // small terms first; every product is exact in fp32
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
        __syncthreads();
    }
    float* C = g.dw[p];
    const int ldc = g.ldw[p];
    const int gn = n0 + wn + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        atomicAdd(C + (size_t)gm * ldc + gn, acc[r]);
    }
    if (g.db[p] != nullptr && n0 == 0) {   // 8 threads (kb) hold partial sums of the same four channels: combine through LDS
        float* red = reinterpret_cast<float*>(planes);       // [8][64]
        if (!isB) *reinterpret_cast<float4*>(red + kb * 64 + 4 * cg) = colsum;
        __syncthreads();
        if (tid < 64) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) sum += red[q * 64 + tid];
            atomicAdd(g.db[p] + m0 + tid, sum);
        }
    }
}

// ------------------------------------------------------------------------------------
// k_wgrad_split128: the same products on 128 x 128 output tiles (round 3).  k_wgrad_split's 64 x 64 tile gives a wave ONE
// 32 x 32 accumulator: six 16-byte LDS reads per six MFMAs (144 KB of LDS reads per 64 k values of a tile - the LDS pipe, not the
// matrix pipe, bound it, and three-piece weight gradients only tied the fp32 instruction).  Here a wave owns 64 x 64 (2 x 2
// accumulators): twelve reads per 24 MFMAs, half the LDS bytes per FLOP; k tiles of 32 keep the two operands' planes at 60 KB
// (two workgroups per CU: one stages and splits while the other multiplies).  Staging as before - a thread transposes eight
// consecutive k rows of four channels in registers and writes one 16-byte piece per channel and plane - with the k-octet
// position XOR-swizzled by the channel group, so the sixteen lanes of a store cycle fall on sixteen different bank groups
// (channel rows four apart put them on four).  Every M_p, N_p a multiple of 128.
// ------------------------------------------------------------------------------------
template <int NPC, bool XS = false>
__global__ __launch_bounds__(256, 2) void k_wgrad_split128(WgradBatch g, int K, int kchunk, int zsplit) {
    constexpr int KT = 32, PLD = KT + 8, TM = 128, PE = TM * PLD;     // bf16 elements per plane row / per plane
    __shared__ __attribute__((aligned(16))) unsigned short planes[2 * NPC * PE];
    const int p = blockIdx.z / zsplit, kz = blockIdx.z - p * zsplit;
    const int M = g.M[p], N = g.N[p];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TM;
    if (m0 >= M || n0 >= N) return;
    const int k_begin = kz * kchunk, k_end = min(K, k_begin + kchunk);
    if (k_begin >= k_end) return;
    const bool isB = tid >= 128;
    const int t = tid & 127, cg = t & 31, kb = t >> 5;                 // channels 4 cg .. 4 cg + 3, k rows 8 kb .. 8 kb + 7 of the tile
    const float* __restrict__ G = isB ? g.x[p] : g.dy[p];
    const int ld = isB ? g.ldx[p] : g.lddy[p];
    const float* gsrc = G + (isB ? n0 : m0) + 4 * cg;
    // octet kb of channel row ch lies at element 8 (kb ^ ((ch >> 4) & 3)) of the row
    unsigned short* mine = planes + (isB ? NPC * PE : 0) + 4 * cg * PLD + 8 * (kb ^ ((cg >> 2) & 3));
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const bool want_bias = g.db[p] != nullptr && n0 == 0 && !isB;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    float4 v[8];
    float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + 8 * kb + i;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < k_end) v[i] = *reinterpret_cast<const float4*>(gsrc + (size_t)k * ld);
        }
    };
    // fragment rows of this lane: row (lane & 31) of each 32-row tile, k octet (lane >> 5) of a 16-k block; (row >> 4) & 3 = 2 (tile & 1) + ((lane >> 4) & 1)
    const int lrow = lane & 31, loct = lane >> 5, lsw = (lane >> 4) & 1;
    fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += KT) {
        if (want_bias) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { colsum.x += v[i].x; colsum.y += v[i].y; colsum.z += v[i].z; colsum.w += v[i].w; }
        }
        {
            if constexpr (XS) if (isB && g.xs[p]) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = silu4(v[i]);
            }
            const float c0[8] = {v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x};
            const float c1[8] = {v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y};
            const float c2[8] = {v[0].z, v[1].z, v[2].z, v[3].z, v[4].z, v[5].z, v[6].z, v[7].z};
            const float c3[8] = {v[0].w, v[1].w, v[2].w, v[3].w, v[4].w, v[5].w, v[6].w, v[7].w};
            const float* cc[4] = {c0, c1, c2, c3};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned short* dst = mine + c * PLD;
                if constexpr (NPC == 3) {
                    sbf16x8 p0, p1, p2;
                    split8(make_float4(cc[c][0], cc[c][1], cc[c][2], cc[c][3]), make_float4(cc[c][4], cc[c][5], cc[c][6], cc[c][7]), p0, p1, p2);
                    *reinterpret_cast<sbf16x8*>(dst) = p0; *reinterpret_cast<sbf16x8*>(dst + PE) = p1; *reinterpret_cast<sbf16x8*>(dst + 2 * PE) = p2;
                } else {
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    const u4 q = {cvt_pk_bf16(cc[c][0], cc[c][1]), cvt_pk_bf16(cc[c][2], cc[c][3]), cvt_pk_bf16(cc[c][4], cc[c][5]), cvt_pk_bf16(cc[c][6], cc[c][7])};
                    *reinterpret_cast<u4*>(dst) = q;
                }
            }
        }
        __syncthreads();
        if (k0 + KT < k_end) fetch(k0 + KT);
#pragma unroll
        for (int ks = 0; ks < KT / 16; ++ks) {
            sbf16x8 a[2][NPC], b[2][NPC];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                // tile rows wm + 32 m + lrow (wm a multiple of 64): channel group bits (row >> 4) & 3 = 2 m + lsw
                const int oa = (wm + 32 * m + lrow) * PLD + 8 * ((2 * ks + loct) ^ (2 * m + lsw));
                const int ob = NPC * PE + (wn + 32 * m + lrow) * PLD + 8 * ((2 * ks + loct) ^ (2 * m + lsw));
#pragma unroll
                for (int q = 0; q < NPC; ++q) {
                    a[m][q] = *reinterpret_cast<const sbf16x8*>(planes + oa + q * PE);
                    b[m][q] = *reinterpret_cast<const sbf16x8*>(planes + ob + q * PE);
                }
            }
#define WG_MF(AI, BI) _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < 2; ++n)                   \
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][AI], b[n][BI], acc[m][n], 0, 0, 0);
            if constexpr (NPC == 3) { WG_MF(2, 0) WG_MF(1, 1) WG_MF(0, 2) WG_MF(1, 0) WG_MF(0, 1) }      // small terms first; every product is exact in fp32
            WG_MF(0, 0)
#undef WG_MF
        }
        __syncthreads();
    }
    float* C = g.dw[p];
    const int ldc = g.ldw[p];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int gn = n0 + wn + 32 * n + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                atomicAdd(C + (size_t)gm * ldc + gn, acc[m][n][r]);
            }
        }
    if (g.db[p] != nullptr && n0 == 0) {   // 4 threads (kb) hold partial sums of the same four channels: combine through LDS
        float* red = reinterpret_cast<float*>(planes);       // [4][128]
        if (!isB) *reinterpret_cast<float4*>(red + kb * TM + 4 * cg) = colsum;
        __syncthreads();
        if (tid < TM) atomicAdd(g.db[p] + m0 + tid, (red[tid] + red[TM + tid]) + (red[2 * TM + tid] + red[3 * TM + tid]));
    }
}

// split3: the handle runs on the split engine, so three-piece (fp32-accurate) weight gradients are allowed; force3: a test asks for them.
// Which kernel (tools/bench_wgrad.py, profiles/r03_n_wgrad.txt; one 256 x 256 product, us per launch):
//     K                fp32 instruction   three pieces 64-tile / 128-tile   bf16 operands 64-tile / 128-tile
//     3 721 (nodes)          18               14 / 17                               12 / 13
//    16 127                  32               32 / 35                               23 / 24
//    36 147                  58               59 / 51                               38 / 37
//   156 316                 253              227 / 175                             155 / 103
// These products are tall and skinny (256 x 256 outputs over 1e4-1e5 rows): operand reads and the split-K atomics bind them, not the
// matrix pipe.  The 128-tile kernel reads each operand row half as often and wins once K >= ~32k (three pieces) / ~64k (bf16); below
// that the fp32 instruction (three pieces) and the 64-tile kernel (bf16) stay.  Options (TrainTune, cmdgen_set_option): wgrad_split = 0: never
// three pieces; = 1: always where the shape allows; wgrad_tile = 64: never the 128-tile kernel.
thread_local TrainTune g_train_tune;      // set from the handle at every entry of the training step (cmdgen_train.hip)
void cmdgen_wgrad_group(const WgradBatch& g, int K, bool bf16, hipStream_t s, bool split3 = false, bool force3 = false) {
    if (g.n <= 0 || K <= 0) return;
    bool xs = false;
    for (int p = 0; p < g.n; ++p) xs = xs || g.xs[p] != 0;
    int tm = 1, tn = 1;
    for (int p = 0; p < g.n; ++p) { tm = max(tm, (g.M[p] + 63) / 64); tn = max(tn, (g.N[p] + 63) / 64); }
    {
        const int env3 = g_train_tune.wgrad_split;
        const bool tile64 = g_train_tune.wgrad_tile == 64;
        bool ok64 = true, ok128 = !tile64;
        for (int p = 0; p < g.n; ++p) {
            ok64 = ok64 && g.M[p] % 64 == 0 && g.N[p] % 64 == 0 && g.lddy[p] % 4 == 0 && g.ldx[p] % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(g.dy[p]) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.x[p]) & 15) == 0;
            ok128 = ok128 && g.M[p] % 128 == 0 && g.N[p] % 128 == 0;
        }
        ok128 = ok128 && ok64;
        // Round 5: the weight gradients run BESIDE the chain of data gradients (cmdgen_train.hip), where the fp32 instruction's long hold on the
        // matrix pipe costs the other stream more than its own launch gains: three pieces whenever the handle is on the split engine, 64 x 64
        // tiles for short products and 128 x 128 from TrainTune::wgrad_k128 rows (same-box sweeps with both streams running: profiles/r05_an, r05_ao)
        // (the 128-tile kernel splits K over wgrad_split_wgs128 / tiles workgroups: it pays once a workgroup's share is ~500 rows, i.e. from
        // K x tiles >= TrainTune::wgrad_k128 - a 4-product node-level group at 15k rows, a single product from 49k)
        long tiles128 = 0;
        for (int p = 0; p < g.n; ++p) tiles128 += (long)(g.M[p] / 128) * (g.N[p] / 128);
        const bool three = !bf16 && (force3 || (split3 && env3 != 0 && ok64));
        const bool sp = ok64 && (bf16 || three);
        const bool sp128 = sp && ok128 && (force3 || env3 == 1 || (bf16 ? K >= 65536 : (long)K * tiles128 >= (long)g_train_tune.wgrad_k128));
        if (sp128) {
            int tm = 1, tn = 1, tiles = 0;
            for (int p = 0; p < g.n; ++p) { tm = max(tm, g.M[p] / 128); tn = max(tn, g.N[p] / 128); tiles += (g.M[p] / 128) * (g.N[p] / 128); }
            const int wgs = g_train_tune.wgrad_split_wgs128;
            int zsplit = (wgs + tiles - 1) / tiles;
            const int max_split = (K + 127) / 128;
            if (zsplit > max_split) zsplit = max_split;
            if (zsplit < 1) zsplit = 1;
            const int kchunk = ((K + zsplit - 1) / zsplit + 31) / 32 * 32;
            zsplit = (K + kchunk - 1) / kchunk;
            const dim3 grid(tn, tm, g.n * zsplit);
            if (xs) { if (bf16) hipLaunchKernelGGL((k_wgrad_split128<1, true>), grid, dim3(256), 0, s, g, K, kchunk, zsplit);
                      else hipLaunchKernelGGL((k_wgrad_split128<3, true>), grid, dim3(256), 0, s, g, K, kchunk, zsplit); }
            else if (bf16) hipLaunchKernelGGL(k_wgrad_split128<1>, grid, dim3(256), 0, s, g, K, kchunk, zsplit);
            else hipLaunchKernelGGL(k_wgrad_split128<3>, grid, dim3(256), 0, s, g, K, kchunk, zsplit);
            return;
        }
        if (sp) {
            int tm = 1, tn = 1;
            for (int p = 0; p < g.n; ++p) { tm = max(tm, g.M[p] / 64); tn = max(tn, g.N[p] / 64); }
            const int wgs = g_train_tune.wgrad_split_wgs64;
            int zsplit = (wgs + tm * tn * g.n - 1) / (tm * tn * g.n);
            const int max_split = (K + 127) / 128;
            if (zsplit > max_split) zsplit = max_split;
            if (zsplit < 1) zsplit = 1;
            const int kchunk = ((K + zsplit - 1) / zsplit + 63) / 64 * 64;
            zsplit = (K + kchunk - 1) / kchunk;
            const dim3 grid(tn, tm, g.n * zsplit);
            if (xs) { if (bf16) hipLaunchKernelGGL((k_wgrad_split<1, true>), grid, dim3(256), 0, s, g, K, kchunk, zsplit);
                      else hipLaunchKernelGGL((k_wgrad_split<3, true>), grid, dim3(256), 0, s, g, K, kchunk, zsplit); }
            else if (bf16) hipLaunchKernelGGL(k_wgrad_split<1>, grid, dim3(256), 0, s, g, K, kchunk, zsplit);
            else hipLaunchKernelGGL(k_wgrad_split<3>, grid, dim3(256), 0, s, g, K, kchunk, zsplit);
            return;
        }
    }
    const int target_wgs = g_train_tune.wgrad_wgs;      // default 768: 3 workgroups (49 KB of LDS each) per CU; sweep: profiles/r02_t3_training_round2.txt
    int zsplit = (target_wgs + tm * tn * g.n - 1) / (tm * tn * g.n);
    const int max_split = (K + 127) / 128;
    if (zsplit > max_split) zsplit = max_split;
    if (zsplit < 1) zsplit = 1;
    const int kchunk = ((K + zsplit - 1) / zsplit + 63) / 64 * 64;
    zsplit = (K + kchunk - 1) / kchunk;
    const dim3 grid(tn, tm, g.n * zsplit), block(256);
    if (xs) { if (bf16) hipLaunchKernelGGL((k_wgrad_group<true, true>), grid, block, 0, s, g, K, kchunk, zsplit);
              else hipLaunchKernelGGL((k_wgrad_group<false, true>), grid, block, 0, s, g, K, kchunk, zsplit); }
    else if (bf16) hipLaunchKernelGGL(k_wgrad_group<true>, grid, block, 0, s, g, K, kchunk, zsplit);
    else hipLaunchKernelGGL(k_wgrad_group<false>, grid, block, 0, s, g, K, kchunk, zsplit);
}

// ------------------------------------------------------------------------------------
// Per-step re-pack of the parameters the optimizer has just updated into the layouts the fused evaluation kernels
// stream (cmdgen_dev.h): every block's six Linears in both MFMA fragment orders, the transposed embedding tables and
// the radial / d0 weight columns.  Two table-driven launches; ~36 MB of traffic per step.
// ------------------------------------------------------------------------------------
struct RepackFrag { int src_off, ld, out, in, row_split, col_shift; float* dst32; float* dst16; };
struct RepackMisc { int src_off, ld, rows, cols; float* dst; };      // dst[c * rows + r] = theta[src_off + r * ld + c]

__global__ void k_repack_frags(const float* __restrict__ theta, const RepackFrag* __restrict__ tab) {
    const RepackFrag f = tab[blockIdx.y];
    const int n4 = f.out * f.in / 4;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    auto fetch = [&](int row, int k) {
        int r = row, c = k;
        if (f.row_split && row >= f.row_split) { r = row - f.row_split; c = k + f.col_shift; }
        const float* src = theta + f.src_off + (size_t)r * f.ld + c;
        return make_float4(src[0], src[1], src[2], src[3]);
    };
    {   // v_mfma_f32_32x32x2_f32 order: [(nt * KB + kb) * 64 + lane] = W[32 nt + (lane & 31)][8 kb + 4 (lane >> 5) .. +3]
        const int KB = f.in / 8, lane = idx & 63, kb = (idx >> 6) % KB, nt = (idx >> 6) / KB;
        reinterpret_cast<float4*>(f.dst32)[idx] = fetch(32 * nt + (lane & 31), 8 * kb + 4 * (lane >> 5));
    }
    {   // v_mfma_f32_16x16x4_f32 order: [(nt * KB16 + kb) * 64 + lane] = W[16 nt + (lane & 15)][16 kb + 4 (lane >> 4) .. +3]
        const int KB = f.in / 16, lane = idx & 63, kb = (idx >> 6) % KB, nt = (idx >> 6) / KB;
        reinterpret_cast<float4*>(f.dst16)[idx] = fetch(16 * nt + (lane & 15), 16 * kb + 4 * (lane >> 4));
    }
}
__global__ void k_repack_misc(const float* __restrict__ theta, const RepackMisc* __restrict__ tab) {
    const RepackMisc m = tab[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m.rows * m.cols) return;
    const int c = idx / m.rows, r = idx - c * m.rows;
    m.dst[idx] = theta[m.src_off + (size_t)r * m.ld + c];
}
void tr_repack(const float* theta, const void* frag_tab, int n_frag, int max_frag4, const void* misc_tab, int n_misc, int max_misc,
               hipStream_t s) {
    hipLaunchKernelGGL(k_repack_frags, dim3((max_frag4 + 255) / 256, n_frag), dim3(256), 0, s, theta, (const RepackFrag*)frag_tab);
    hipLaunchKernelGGL(k_repack_misc, dim3((max_misc + 255) / 256, n_misc), dim3(256), 0, s, theta, (const RepackMisc*)misc_tab);
}

// split_k: 0 = choose so that the launch fills the chip (wgrad: few output tiles, K = thousands of rows); 1 = none
void cmdgen_sgemm(bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                  int ldc, const float* bias, float alpha, bool accumulate, int split_k, hipStream_t s,
                  int epi = 0, float* aux = nullptr, int ldaux = 0, bool bf16 = false) {
    if (M <= 0 || N <= 0 || K <= 0) return;
    const int tiles = ((N + 63) / 64) * ((M + 63) / 64);
    if (split_k == 0) {
        split_k = (1024 + tiles - 1) / tiles;
        const int max_split = (K + 127) / 128;
        if (split_k > max_split) split_k = max_split;
        if (split_k < 1) split_k = 1;
    }
    int kchunk = K, z = 1;
    if (split_k > 1) {
        kchunk = ((K + split_k - 1) / split_k + 63) / 64 * 64;
        z = (K + kchunk - 1) / kchunk;
    }
    // split-K partials are combined with atomics: the destination must already hold the value to add to
    const dim3 grid((N + 63) / 64, (M + 63) / 64, z), block(256);
    const int acc = accumulate ? 1 : 0;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool va = al(A) && (lda % 4 == 0), vb = al(B) && (ldb % 4 == 0);      // weights inside the flat buffer may be unaligned
    const bool fat = (long)grid.x * grid.y * grid.z < 1024;      // few workgroups: 64-wide k tiles (see TG_LD)
#define SGK(KT_, TA_, TB_, VA_, VB_) do { if (bf16) hipLaunchKernelGGL((k_sgemm_bf16<KT_, TA_, TB_, VA_, VB_>), grid, block, 0, s, M, N, K, A, lda, B, ldb, C, ldc, bias, alpha, acc, kchunk, z > 1 ? 0 : epi, aux, ldaux); \
        else hipLaunchKernelGGL((k_sgemm<KT_, TA_, TB_, VA_, VB_>), grid, block, 0, s, M, N, K, A, lda, B, ldb, C, ldc, bias, alpha, acc, kchunk, z > 1 ? 0 : epi, aux, ldaux); } while (0)
#define SG(TA_, TB_, VA_, VB_) do { if (fat) SGK(64, TA_, TB_, VA_, VB_); else SGK(32, TA_, TB_, VA_, VB_); } while (0)
#define SG2(TA_, TB_) do { if (va && vb) SG(TA_, TB_, true, true); else if (va) SG(TA_, TB_, true, false); \
                           else if (vb) SG(TA_, TB_, false, true); else SG(TA_, TB_, false, false); } while (0)
    if (!ta && tb) SG2(false, true); else if (!ta && !tb) SG2(false, false);
    else if (ta && !tb) SG2(true, false); else SG2(true, true);
#undef SG2
#undef SG
#undef SGK
}

// ------------------------------------------------------------------------------------
// Data gradients on the split-bf16 matrix engine (cmdgen_split.h: fp32-accurate, six bf16 MFMAs per fp32 product):
//   Y[M][256] (+)= ( A0[M][256] W0 + A1[M][256] W1 ) / div  (* SiLU'(pre[M][256]))
// where W0 / W1 are 256 x 256 sub-blocks W[:, col0 : col0 + 256] of nn.Linear weights (dX = dY W), streamed as split
// fragment packs of their TRANSPOSES (k_repack_split_t, re-made every step from the parameters the optimizer has just
// updated).  All [.,256] x [256,256] products of the backward pass have this shape: the second layer of the edge /
// coordinate MLP over the edges, and the seven node-level products of a block - the two halves of edge_mlp.0 and
// coord_mlp.0 (A0 = dP, A1 = dQ: one launch instead of two), node_mlp.2, and the two halves of node_mlp.0.
// One workgroup per MT rows: the A tile is staged once in LDS as fp32, every wave splits the fragments it reads in
// registers in the shadow of the MFMAs (tile_gemm_rsplit), the weight fragments stream from L2.
// ------------------------------------------------------------------------------------
// the GEMM part: acc = A0[row0 .. row0+MT) W0 (+ A1 W1).  The A operand lies in LDS as three bf16 planes [MT][136] of one
// half of the k range at a time (tile_gemm_planes): the thread that loads an element splits it once; the next half's rows
// are requested before this half's MFMAs start.  Ends with a barrier (the planes may be overwritten).
template <int MT, int NPC>
__device__ __forceinline__ void dgrad_tile_gemm(unsigned short* planes, int M, int row0, const float* __restrict__ A0,
                                                const void* __restrict__ W0, const float* __restrict__ A1,
                                                const void* __restrict__ W1, sf32x16 (&acc)[MT / 32][2]) {
    constexpr int HH = 256, PLDA = SPLIT_PLANE_LDA(HH / 2), PE = MT * PLDA, NP = MT / 8;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int c4 = tid & 31, rsub = tid >> 5;            // 32 lanes x 16 bytes = one half row, 8 rows per pass
    SCarry carry;
    const int nstage = A1 ? 4 : 2;                        // (source, half)
    auto frag_of = [&](int st) { return sfrag_ptr((st >> 1) ? W1 : W0, HH / 16, (st & 1) * (HH / 32), wave); };
    split_prefetch<NPC>(frag_of(0), carry);
    float4 v[NP];
    auto fetch = [&](int st) {
        const float* A = (st >> 1) ? A1 : A0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = row0 + p * 8 + rsub;
            v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < M) v[p] = *reinterpret_cast<const float4*>(A + (size_t)r * HH + (st & 1) * (HH / 2) + 4 * c4);
        }
    };
    fetch(0);
#pragma unroll
    for (int m = 0; m < MT / 32; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
#pragma unroll 1
    for (int st = 0; st < nstage; ++st) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if constexpr (NPC == 3) split_store4(planes, PE, (p * 8 + rsub) * PLDA + 4 * c4, v[p]);
            else *reinterpret_cast<uint2*>(planes + (p * 8 + rsub) * PLDA + 4 * c4) = make_uint2(cvt_pk_bf16(v[p].x, v[p].y), cvt_pk_bf16(v[p].z, v[p].w));
        }
        __syncthreads();
        if (st + 1 < nstage) fetch(st + 1);
        tile_gemm_planes<MT, HH / 32, NPC>(planes, PE, PLDA, frag_of(st), frag_of(st + 1 < nstage ? st + 1 : st), acc, carry);
        __syncthreads();
    }
}

template <int MT, int NPC>
__global__ __launch_bounds__(256, 2) void k_dgrad_split(int M, const float* __restrict__ A0, const void* __restrict__ W0,
                                                        const float* __restrict__ A1, const void* __restrict__ W1,
                                                        float* Y, int accumulate, float div,
                                                        const float* __restrict__ pre, const void* __restrict__ W0b,
                                                        float* __restrict__ Yb, int accumulate_b, float div_b, const float* Yin,
                                                        const float* __restrict__ rowdiv_b) {
    // gridDim.y = 2: a second product of the same A0 (W0b -> Yb; the two halves of node_mlp.0 share dpre3) in the same launch
    // Yin: the running value an accumulating product adds to (Y itself, or another buffer: Y = Yin + product, out of place)
    // rowdiv_b: per-row divisor of the second product instead of div_b (aggregation 'mean': the receiver's edge count)
    const float* rowdiv = nullptr;
    if (blockIdx.y) { W0 = W0b; Y = Yb; Yin = Yb; accumulate = accumulate_b; div = div_b; rowdiv = rowdiv_b; }
    constexpr int HH = 256, PLDA = SPLIT_PLANE_LDA(HH / 2), PE = MT * PLDA;
    constexpr int LDO = HH + 4, SMEM = 3 * PE * 2 > 32 * LDO * 4 ? 3 * PE * 2 : 32 * LDO * 4;   // planes / 32-row fp32 output image
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * MT;
    sf32x16 acc[MT / 32][2];
    dgrad_tile_gemm<MT, NPC>(reinterpret_cast<unsigned short*>(smem), M, row0, A0, W0, A1, W1, acc);
    // Epilogue through LDS, 32 rows at a time (the fp32 image of 32 rows, 33 KB, aliases the planes): whole 1 KB rows leave
    // as 16-byte pieces, and SiLU'(pre) / the running value of Y are read the same way.  Accumulator element r of tile
    // (m, n) is row m*32 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column wave*64 + n*32 + (lane & 31).
    float* obuf = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int m = 0; m < MT / 32; ++m) {
        if (m) __syncthreads();
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                obuf[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDO + wave * 64 + n * 32 + (lane & 31)] = acc[m][n][r];
        __syncthreads();
        const int q4 = tid & 63, rs = tid >> 6;          // a wave writes one row per pass
        float4 pv[8], yv[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = row0 + m * 32 + p * 4 + rs;
            if (row < M) {
                if (pre) pv[p] = reinterpret_cast<const float4*>(pre + (size_t)row * HH)[q4];
                if (accumulate) yv[p] = reinterpret_cast<const float4*>(Yin + (size_t)row * HH)[q4];
            }
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int rl = p * 4 + rs, row = row0 + m * 32 + rl;
            if (row < M) {
                float4 x = *reinterpret_cast<const float4*>(obuf + rl * LDO + 4 * q4);
                if (rowdiv) { const float dvr = rowdiv[row]; x.x = x.x / dvr; x.y = x.y / dvr; x.z = x.z / dvr; x.w = x.w / dvr; }
                else if (div != 1.0f) { x.x = x.x / div; x.y = x.y / div; x.z = x.z / div; x.w = x.w / div; }
                if (pre) { x.x *= dsilu(pv[p].x); x.y *= dsilu(pv[p].y); x.z *= dsilu(pv[p].z); x.w *= dsilu(pv[p].w); }
                if (accumulate) { x.x += yv[p].x; x.y += yv[p].y; x.z += yv[p].z; x.w += yv[p].w; }
                reinterpret_cast<float4*>(Y + (size_t)row * HH)[q4] = x;
            }
        }
    }
}

// split fragment packs of transposed weight sub-blocks: dst = pack of Wt, Wt[o'][k] = theta[src_off + k * ld + o'] (o', k < 256)
struct RepackSplitT { int src_off, ld; void* dst; int transpose; };      // transpose = 0: the pack of W itself (forward: Y = X W^T)
__global__ void k_repack_split_t(const float* __restrict__ theta, const RepackSplitT* __restrict__ tab) {
    const RepackSplitT f = tab[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;         // (nt * 16 + kb) * 64 + lane, nt < 8, kb < 16
    if (idx >= 8 * 16 * 64) return;
    const int lane = idx & 63, kb = (idx >> 6) & 15, nt = idx >> 10;
    const int o = 32 * nt + (lane & 31), k = 16 * kb + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = f.transpose ? theta[f.src_off + (size_t)(k + j) * f.ld + o] : theta[f.src_off + (size_t)o * f.ld + k + j];
    sbf16x8 p0, p1, p2;
    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p0, p1, p2);
    sbf16x8* d = reinterpret_cast<sbf16x8*>(f.dst) + (size_t)((nt * 16 + kb) * 3) * 64 + lane;
    d[0] = p0; d[64] = p1; d[128] = p2;
}
void tr_repack_split_t(const float* theta, const void* tab, int n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_repack_split_t, dim3(8 * 16 * 64 / 256, n), dim3(256), 0, s, theta, (const RepackSplitT*)tab);
}
// Half-engine packs of the training forward's two edge kernels (cmdgen_split.h, "half" engine), re-made every step like the split packs:
// two fp16 pieces of (w * 2^e) per weight in v_mfma_f32_32x32x16_f16 fragment order (the layout of pack_half, cmdgen_api.hip), e chosen ON
// THE DEVICE so that the largest |w| of the matrix lands in [2^11, 2^12) - the parameters move every step, no host value can be trusted -
// and sc = {2^e, 2^-e} left beside the pack for the kernel's epilogue (WPack::wh_dev).  One workgroup of 1024 threads per 256 x 256 matrix:
// the 64 weights a thread packs stay in its registers between the maximum and the split.
struct RepackHalf { int src_off, ld; void* dst; float* sc; int transpose; };     // transpose: the pack of W^T (data gradients: dX = dY W)
template <bool TR>      // TR: the entries are transposed blocks (a strided gather; its own instantiation: sharing one cost the plain packs 20 us per step)
__global__ __launch_bounds__(1024) void k_repack_half(const float* __restrict__ theta, const RepackHalf* __restrict__ tab) {
    const RepackHalf f = tab[blockIdx.x];
    __shared__ float red[16];
    const int tid = threadIdx.x;
    float4 va[8], vb[8];
    float mx = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = q * 1024 + tid;                                // (nt * 16 + kb) * 64 + lane, nt < 8, kb < 16
        const int lane = idx & 63, kb = (idx >> 6) & 15, nt = idx >> 10;
        const int o = 32 * nt + (lane & 31), k = 16 * kb + 8 * (lane >> 5);
        if constexpr (TR) {                                            // Wt[o][k] = W[k][o]
            const float* src = theta + f.src_off + (size_t)k * f.ld + o;
            va[q] = make_float4(src[0], src[(size_t)f.ld], src[2 * (size_t)f.ld], src[3 * (size_t)f.ld]);
            vb[q] = make_float4(src[4 * (size_t)f.ld], src[5 * (size_t)f.ld], src[6 * (size_t)f.ld], src[7 * (size_t)f.ld]);
        } else {
            const float4* src = reinterpret_cast<const float4*>(theta + f.src_off + (size_t)o * f.ld + k);
            va[q] = src[0]; vb[q] = src[1];
        }
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(va[q].x), fabsf(va[q].y)), fmaxf(fabsf(va[q].z), fabsf(va[q].w))));
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(vb[q].x), fabsf(vb[q].y)), fmaxf(fabsf(vb[q].z), fabsf(vb[q].w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) e = 12 - ((int)((__float_as_uint(mx) >> 23) & 0xffu) - 126);     // mx = m 2^ex, m in [0.5, 1): mx 2^e in [2^11, 2^12)
    e = max(-40, min(40, e));
    const float sc = __uint_as_float((unsigned)(127 + e) << 23);
    if (tid == 0) { f.sc[0] = sc; f.sc[1] = __uint_as_float((unsigned)(127 - e) << 23); }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = q * 1024 + tid;
        const int lane = idx & 63;
        const float w[8] = {va[q].x * sc, va[q].y * sc, va[q].z * sc, va[q].w * sc, vb[q].x * sc, vb[q].y * sc, vb[q].z * sc, vb[q].w * sc};
        union { _Float16 h[8]; uint4 u; } p0, p1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { p0.h[j] = (_Float16)w[j]; p1.h[j] = (_Float16)(w[j] - (float)p0.h[j]); }
        uint4* d = reinterpret_cast<uint4*>(f.dst) + (size_t)((idx >> 6) * 2) * 64 + lane;
        d[0] = p0.u; d[64] = p1.u;
    }
}
// entries [0, n_plain) are packs of W itself, [n_plain, n) of transposed blocks (the table's order, cmdgen_train.hip)
void tr_repack_half(const float* theta, const void* tab, int n_plain, int n, hipStream_t s) {
    if (n_plain) hipLaunchKernelGGL(k_repack_half<false>, dim3(n_plain), dim3(1024), 0, s, theta, (const RepackHalf*)tab);
    if (n > n_plain) hipLaunchKernelGGL(k_repack_half<true>, dim3(n - n_plain), dim3(1024), 0, s, theta, (const RepackHalf*)tab + n_plain);
}
// The same for the node kernel of the training forward (k_node16w<true>): the 16-row half packs (v_mfma_f32_16x16x32_f16 fragment order,
// the layout of pack_half16, cmdgen_api.hip) of node_mlp.0 [H][2H], node_mlp.2 [H][H] and the stacked projections [2H][H] of coord_mlp.0 /
// edge_mlp.0 (RepackFrag's row_split / col_shift).  A pack has ONE scale (its products share accumulators), so the maximum is a launch of
// its own: k_wmax16 (one workgroup per pack) leaves {2^e, 2^-e}, k_repack_half16 splits.
struct RepackHalf16 { int src_off, ld, out, in, row_split, col_shift; void* dst; float* sc; };
__device__ __forceinline__ const float* rh16_src(const float* theta, const RepackHalf16& f, int row, int k) {
    int r = row, c = k;
    if (f.row_split && row >= f.row_split) { r = row - f.row_split; c = k + f.col_shift; }
    return theta + f.src_off + (size_t)r * f.ld + c;
}
__global__ __launch_bounds__(1024) void k_wmax16(const float* __restrict__ theta, const RepackHalf16* __restrict__ tab) {
    const RepackHalf16 f = tab[blockIdx.x];
    __shared__ float red[16];
    const int tid = threadIdx.x, n4 = f.out * f.in / 4, k4 = f.in / 4;
    float mx = 0.f;
    const int sh = 31 - __clz(k4);                                      // in is 256 or 512: k4 a power of two (no integer division per element)
#pragma unroll 4
    for (int i = tid; i < n4; i += 1024) {
        const float2* src = reinterpret_cast<const float2*>(rh16_src(theta, f, i >> sh, 4 * (i & (k4 - 1))));      // (every tensor starts 16-byte aligned, rows are an even number of floats)
        const float2 a = src[0], b = src[1];
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(b.x), fabsf(b.y))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        mx = red[0];
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f) e = 12 - ((int)((__float_as_uint(mx) >> 23) & 0xffu) - 126);
        e = max(-40, min(40, e));
        f.sc[0] = __uint_as_float((unsigned)(127 + e) << 23); f.sc[1] = __uint_as_float((unsigned)(127 - e) << 23);
    }
}
__global__ void k_repack_half16(const float* __restrict__ theta, const RepackHalf16* __restrict__ tab) {
    const RepackHalf16 f = tab[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;         // (nt * KB + kb) * 64 + lane, nt < out / 16, kb < in / 32
    if (idx >= f.out * f.in / 8) return;
    const int KB = f.in / 32, lane = idx & 63, kb = (idx >> 6) % KB, nt = (idx >> 6) / KB, g = lane >> 4;
    const float sc = f.sc[0];
    const float* lo = rh16_src(theta, f, 16 * nt + (lane & 15), 32 * kb + 4 * g);           // k = 4g .. 4g + 3, then 16 + 4g .. 16 + 4g + 3
    const float w[8] = {lo[0] * sc, lo[1] * sc, lo[2] * sc, lo[3] * sc, lo[16] * sc, lo[17] * sc, lo[18] * sc, lo[19] * sc};
    union { _Float16 h[8]; uint4 u; } p0, p1;
#pragma unroll
    for (int j = 0; j < 8; ++j) { p0.h[j] = (_Float16)w[j]; p1.h[j] = (_Float16)(w[j] - (float)p0.h[j]); }
    uint4* d = reinterpret_cast<uint4*>(f.dst) + (size_t)((idx >> 6) * 2) * 64 + lane;
    d[0] = p0.u; d[64] = p1.u;
}
void tr_repack_half16(const float* theta, const void* tab, int n, int max8, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_wmax16, dim3(n), dim3(1024), 0, s, theta, (const RepackHalf16*)tab);
    hipLaunchKernelGGL(k_repack_half16, dim3((max8 + 255) / 256, n), dim3(256), 0, s, theta, (const RepackHalf16*)tab);
}
// pieces = 3: fp32-accurate (split engine); 1: the operands' leading bf16 piece only (= operands rounded to nearest-even
// bf16, fp32 accumulation: cmdgen_train_set_precision(1))
void cmdgen_dgrad_split(int M, const float* A0, const void* W0, const float* A1, const void* W1, float* Y, bool accumulate, float div,
                        const float* pre, hipStream_t s, int pieces = 3, const void* W0b = nullptr, float* Yb = nullptr,
                        bool accumulate_b = false, float div_b = 1.0f, int force_mt = 0, const float* Yin = nullptr, const float* rowdiv_b = nullptr) {
    if (M <= 0) return;
    if (!Yin) Yin = Y;
    const int mt = g_train_tune.dgrad_mt;
    const bool big = force_mt ? force_mt == 64 : (mt ? mt == 64 : M >= 24576);     // force_mt: cmdgen_debug_dgrad
    const int acc = accumulate ? 1 : 0, ny = W0b ? 2 : 1;
#define DG(MT_, NP_) hipLaunchKernelGGL((k_dgrad_split<MT_, NP_>), dim3((M + MT_ - 1) / MT_, ny), dim3(256), 0, s, M, A0, W0, A1, W1, Y, acc, div, pre, \
                                        W0b, Yb, accumulate_b ? 1 : 0, div_b, Yin, rowdiv_b)
    if (pieces == 3) { if (big) DG(64, 3); else DG(32, 3); }
    else { if (big) DG(64, 1); else DG(32, 1); }
#undef DG
}

// g <- g * SiLU'(pre)
__global__ void k_silu_bwd(float* __restrict__ g, const float* __restrict__ pre, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] *= dsilu(pre[i]);
}
__global__ void k_silu_bwd4(float4* __restrict__ g, const float4* __restrict__ pre, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 p = pre[i]; float4 v = g[i];
    v.x *= dsilu(p.x); v.y *= dsilu(p.y); v.z *= dsilu(p.z); v.w *= dsilu(p.w);
    g[i] = v;
}
// x[row][:] /= div[row]  (aggregation 'mean': the receiver's edge count)
__global__ void k_scale_rows(float* __restrict__ x, const float* __restrict__ div, int H, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] / div[i / H];
}
__global__ void k_scale(float* __restrict__ x, float d, size_t n) {      // x /= d (a true division, as the reference)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] /= d;
}



__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}



// ------------------------------------------------------------------------------------
// The same adjoint in ONE pass over pre2, together with the two reductions that used to follow it (k_colsum4 over m2
// with weights dz, k_sum over dz): m2 = SiLU(pre2) is recomputed with the forward's own silu_f instead of being read
// (the forward no longer stores it), d att_mlp.weight[c] += sum_e dz_e m2[e][c] and d att_mlp.bias += sum_e dz_e are
// accumulated per workgroup (32 consecutive edges, 8 per wave, a lane owns four consecutive columns) and written to a
// scratch row; k_partial_reduce adds the rows up.  H <= 256, H % 4 == 0.
// ------------------------------------------------------------------------------------
#define GATE_EPW 8
__global__ __launch_bounds__(256) void k_gate_bwd(int E, int H, const int* __restrict__ row, const float* __restrict__ pre2,
                                                  const float* __restrict__ wa, const float* __restrict__ z, int attention,
                                                  const float* __restrict__ dagg, float* __restrict__ dpre2,
                                                  float* __restrict__ scratch /* [workgroups][H + 4] */,
                                                  float4* __restrict__ zero, size_t zero_n4) {
    __shared__ float red[4][260];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // side duty: clear the buffer the NEXT kernel of the pass accumulates into (dP | dQ) - one memset launch less per list
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < zero_n4; i += (size_t)gridDim.x * 256) zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool on = 4 * lane < H;
    const int e0 = blockIdx.x * (4 * GATE_EPW) + wave * GATE_EPW, e1 = min(E, e0 + GATE_EPW);
    float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f), cs = w4;
    if (on && attention) w4 = reinterpret_cast<const float4*>(wa)[lane];
    float sdz = 0.f;
    float4 pv[GATE_EPW], gv[GATE_EPW];
    float zv[GATE_EPW];
#pragma unroll
    for (int k = 0; k < GATE_EPW; ++k) {            // all rows of the wave requested before the first is used
        const int e = e0 + k;
        pv[k] = gv[k] = make_float4(0.f, 0.f, 0.f, 0.f); zv[k] = 0.f;
        if (e < e1) {
            if (on) {
                pv[k] = reinterpret_cast<const float4*>(pre2 + (size_t)e * H)[lane];
                gv[k] = reinterpret_cast<const float4*>(dagg + (size_t)row[e] * H)[lane];
            }
            if (attention) zv[k] = z[e];
        }
    }
#pragma unroll
    for (int k = 0; k < GATE_EPW; ++k) {
        const int e = e0 + k;
        if (e < e1) {                                // wave-uniform
            const float4 p = pv[k], g = gv[k];
            float att = 1.0f, dz = 0.f;
            if (attention) {
                const float4 m = make_float4(silu_f(p.x), silu_f(p.y), silu_f(p.z), silu_f(p.w));
                att = 1.0f / (1.0f + expf(-zv[k]));
                dz = wave_sum(g.x * m.x + g.y * m.y + g.z * m.z + g.w * m.w) * att * (1.0f - att);
                cs.x += dz * m.x; cs.y += dz * m.y; cs.z += dz * m.z; cs.w += dz * m.w;
                sdz += dz;
            }
            if (on) {
                float4 o;
                o.x = (g.x * att + dz * w4.x) * dsilu(p.x); o.y = (g.y * att + dz * w4.y) * dsilu(p.y);
                o.z = (g.z * att + dz * w4.z) * dsilu(p.z); o.w = (g.w * att + dz * w4.w) * dsilu(p.w);
                reinterpret_cast<float4*>(dpre2 + (size_t)e * H)[lane] = o;
            }
        }
    }
    if (!attention) return;
    *reinterpret_cast<float4*>(&red[wave][4 * lane]) = cs;
    if (lane == 0) red[wave][256] = sdz;
    __syncthreads();
    float* out = scratch + (size_t)blockIdx.x * (H + 4);
    for (int c = threadIdx.x; c < H; c += 256) out[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (threadIdx.x == 0) out[H] = (red[0][256] + red[1][256]) + (red[2][256] + red[3][256]);
}

// the inputs of k_coord_out_bwd, for the form of k_head_bwd that computes dphi (and writes dcd) itself: one launch less per block
struct CoordOutArgs { const int* row; const int* col; const float4* X; const float* phi; int use_tanh; float range, norm_constant;
                      const float* dacc; float dacc_div; const float* adiv; float4* dcd_out; };
// dpre7[e][c] = dphi_e w5[c] SiLU'(pre7[e][c]) and, in the same pass, the partial sums of d coord_mlp.4.weight[c] =
// sum_e dphi_e SiLU(pre7[e][c]) (c2 is recomputed, not read); same workgroup shape and scratch layout as k_gate_bwd.
__global__ __launch_bounds__(256) void k_head_bwd(int E, int H, const float* __restrict__ dphi, const float* __restrict__ w5,
                                                  const float* __restrict__ pre7, float* __restrict__ dpre7,
                                                  float* __restrict__ scratch /* [workgroups][H + 4] */,
                                                  float4* __restrict__ zero, size_t zero_n4, CoordOutArgs co) {
    __shared__ float red[4][260];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < zero_n4; i += (size_t)gridDim.x * 256) zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool on = 4 * lane < H;
    const int e0 = blockIdx.x * (4 * GATE_EPW) + wave * GATE_EPW, e1 = min(E, e0 + GATE_EPW);
    // co.row: dphi is not read but formed here, by lane k for the wave's k-th edge (the arithmetic of k_coord_out_bwd), and dcd written
    float my_dphi = 0.f;
    if (co.row && lane < GATE_EPW && e0 + lane < e1) {
        const int e = e0 + lane;
        const int i = co.row[e], j = co.col[e];
        const float4 a = co.X[i], b = co.X[j];
        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
        const float r = dx * dx + dy * dy + dz * dz;
        const float den = sqrtf(r + 1e-8f) + co.norm_constant;
        const float cx = dx / den, cy = dy / den, cz = dz / den;
        const float4 dv = *reinterpret_cast<const float4*>(co.dacc + (size_t)i * 4);
        const float dd = co.adiv ? co.adiv[i] : co.dacc_div;
        const float da[3] = {dv.x / dd, dv.y / dd, dv.z / dd};
        const float p = co.phi[e];
        const float th = co.use_tanh ? tanhf(p) : 0.f;
        const float g = co.use_tanh ? th * co.range : p;
        const float dg = cx * da[0] + cy * da[1] + cz * da[2];
        my_dphi = co.use_tanh ? dg * co.range * (1.0f - th * th) : dg;
        co.dcd_out[e] = make_float4(g * da[0], g * da[1], g * da[2], 0.f);
    }
    float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f), cs = w4;
    if (on) w4 = reinterpret_cast<const float4*>(w5)[lane];
    float4 pv[GATE_EPW]; float sv[GATE_EPW];
#pragma unroll
    for (int k = 0; k < GATE_EPW; ++k) {
        const int e = e0 + k;
        pv[k] = make_float4(0.f, 0.f, 0.f, 0.f); sv[k] = 0.f;
        if (e < e1 && on) pv[k] = reinterpret_cast<const float4*>(pre7 + (size_t)e * H)[lane];
        if (!co.row && e < e1) sv[k] = dphi[e];
    }
    if (co.row) {
#pragma unroll
        for (int k = 0; k < GATE_EPW; ++k) sv[k] = __shfl(my_dphi, k);
    }
#pragma unroll
    for (int k = 0; k < GATE_EPW; ++k) {
        const int e = e0 + k;
        if (e < e1 && on) {
            const float4 p = pv[k]; const float se = sv[k];
            cs.x += se * silu_f(p.x); cs.y += se * silu_f(p.y); cs.z += se * silu_f(p.z); cs.w += se * silu_f(p.w);
            reinterpret_cast<float4*>(dpre7 + (size_t)e * H)[lane] =
                make_float4(se * w4.x * dsilu(p.x), se * w4.y * dsilu(p.y), se * w4.z * dsilu(p.z), se * w4.w * dsilu(p.w));
        }
    }
    *reinterpret_cast<float4*>(&red[wave][4 * lane]) = cs;
    __syncthreads();
    float* out = scratch + (size_t)blockIdx.x * (H + 4);
    for (int c = threadIdx.x; c < H; c += 256) out[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// out_w[c] += sum over workgroups of scratch[wg][c] (c < H), out_b[0] += sum of scratch[wg][H] (when out_b).  grid
// ((H + 1 + 63) / 64, slices): every workgroup sums one slice of the rows for 64 columns and adds its result with one atomic.
__global__ __launch_bounds__(256) void k_partial_reduce(int nwg, int H, const float* __restrict__ scratch, float* __restrict__ out_w,
                                                        float* __restrict__ out_b) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int ncol = H + (out_b ? 1 : 0);
    const int per = (nwg + gridDim.y - 1) / gridDim.y;
    const int w0 = blockIdx.y * per, w1 = min(nwg, w0 + per);
    float sum = 0.f;
    if (c < ncol) {
#pragma unroll 4
        for (int w = w0 + part; w < w1; w += 4) sum += scratch[(size_t)w * (H + 4) + c];
    }
    red[part][threadIdx.x & 63] = sum;
    __syncthreads();
    if (part == 0 && c < ncol) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        atomicAdd(c < H ? out_w + c : out_b, t);
    }
}


// adjoint of k_coord_out + coord2diff: given dacc [N][4] (gradient of the per-node coordinate sums),
//   dphi = (cd . dacc[row]) * range * (1 - tanh^2 phi);  dcd = g * dacc[row]
// writes dphi[e] and accumulates the geometry part (through cd, and through r when dr != null) into dX.
// dc2[e][c] = dphi * w5[c] is formed by the caller's k_outer_silu_bwd.
__global__ void k_coord_out_bwd(int E, const int* __restrict__ row, const int* __restrict__ col,
                                const float4* __restrict__ X, const float* __restrict__ phi, int use_tanh, float range,
                                float norm_constant, const float* __restrict__ dacc, float dacc_div, int n_moving,
                                float* __restrict__ dphi_out, float4* __restrict__ dcd_out, const float* __restrict__ adiv) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int i = row[e], j = col[e];
    const float4 a = X[i], b = X[j];
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    const float r = dx * dx + dy * dy + dz * dz;
    const float den = sqrtf(r + 1e-8f) + norm_constant;
    const float cx = dx / den, cy = dy / den, cz = dz / den;
    const float4 dv = *reinterpret_cast<const float4*>(dacc + (size_t)i * 4);
    const float dd = adiv ? adiv[i] : dacc_div;                  // aggregation 'mean': the receiver's edge count (egnn_new.py:288-292), else normalization_factor
    const float da[3] = {dv.x / dd, dv.y / dd, dv.z / dd};      // dL/d acc = dL/dx_{l+1} / divisor
    const float p = phi[e];
    const float th = use_tanh ? tanhf(p) : 0.f;
    const float g = use_tanh ? th * range : p;
    const float dg = cx * da[0] + cy * da[1] + cz * da[2];
    dphi_out[e] = use_tanh ? dg * range * (1.0f - th * th) : dg;
    dcd_out[e] = make_float4(g * da[0], g * da[1], g * da[2], 0.f);
    (void)n_moving;
}


// ------------------------------------------------------------------------------------
// k_edge_tail_bwd: everything the backward pass does with g = dL/d pre1 [E][H] of an edge list (pre1 = P[row] + Q[col]
// + w_r r + w_d d0, the first layer of the edge / coordinate MLP), in ONE pass over g instead of seven launches:
//   dP[row] += g (adjoint of the gather of P; rows are sorted, so a wave sums a receiver's run in registers and adds
//   once per run), dQ[col] += g (atomics), dW[:, 2H] += sum_e r_e g_e and dW[:, 2H+1] += sum_e d0_e g_e (the radial /
//   d0 columns of the Linear, strided by ldw), dr_e = g_e . w_r, and the geometry adjoint of k_geom_bwd for that dr_e
//   (plus dcd_e for the coordinate list) into dX.  A wave takes TAIL_EPW consecutive edges (a workgroup 4 x that).
// ------------------------------------------------------------------------------------
#define TAIL_EPW 8              // edges per wave: short serial runs, thousands of waves in flight
__global__ __launch_bounds__(256) void k_edge_tail_bwd(int E, int H, const int* __restrict__ row, const int* __restrict__ col,
                                                       const float* __restrict__ g, const float* __restrict__ d0,
                                                       const float* __restrict__ Wcol /* W + 2H, stride ldw */, int ldw,
                                                       const float4* __restrict__ X, float norm_constant,
                                                       const float4* __restrict__ dcd, int n_moving,
                                                       float* __restrict__ dP, float* __restrict__ dQ,
                                                       float* __restrict__ scratch /* [workgroups][2][H] */, float* __restrict__ dX) {
    __shared__ float red[2][4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a lane owns columns lane, lane + 64, lane + 128, lane + 192: every load / atomic of the wave is one contiguous
    // 256-byte piece of a row (the shape float atomics run at full rate in, MI355X_MICROARCH.md "Global float atomics")
    const int NC = (H + 63) / 64;
    const int e0 = blockIdx.x * (4 * TAIL_EPW) + wave * TAIL_EPW, e1 = min(E, e0 + TAIL_EPW);
    const int ne = max(e1 - e0, 0);
    float wr[4], accR[4], accD[4], run[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + 64 * q;
        wr[q] = (q < NC && c < H) ? Wcol[(size_t)c * ldw] : 0.f;
        accR[q] = accD[q] = run[q] = 0.f;
    }
    // lane l < ne owns edge e0 + l for the per-edge scalars: indices, geometry, and later the geometry adjoint
    int my_i = -1, my_j = -1; float my_r = 0.f, my_d0 = 0.f, dxl = 0.f, dyl = 0.f, dzl = 0.f;
    if (lane < ne) {
        my_i = row[e0 + lane]; my_j = col[e0 + lane]; my_d0 = d0[e0 + lane];
        const float4 a = X[my_i], b = X[my_j];
        dxl = a.x - b.x; dyl = a.y - b.y; dzl = a.z - b.z;
        my_r = dxl * dxl + dyl * dyl + dzl * dzl;
    }
    float my_gr = 0.f;
    int cur = __shfl(my_i, 0);
#pragma unroll
    for (int k = 0; k < TAIL_EPW; ++k) {
        if (k < ne) {                                        // wave-uniform
            const int e = e0 + k;
            const int i = __shfl(my_i, k), j = __shfl(my_j, k);
            const float r = __shfl(my_r, k), dd = __shfl(my_d0, k);
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int c = lane + 64 * q; v[q] = (c < H) ? g[(size_t)e * H + c] : 0.f; }
            if (i != cur) {                                  // the receiver's run ended: one add per run
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int c = lane + 64 * q; if (c < H) atomicAdd(dP + (size_t)cur * H + c, run[q]); run[q] = 0.f; }
                cur = i;
            }
            float dot = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = lane + 64 * q;
                accR[q] += r * v[q]; accD[q] += dd * v[q]; run[q] += v[q]; dot += v[q] * wr[q];
                if (c < H) atomicAdd(dQ + (size_t)j * H + c, v[q]);
            }
            const float gr = wave_sum(dot);                  // dL/d radial of this edge
            if (lane == k) my_gr = gr;
        }
    }
    if (ne > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = lane + 64 * q; if (c < H) atomicAdd(dP + (size_t)cur * H + c, run[q]); }
    }
    if (lane < ne && my_i != my_j) {                         // geometry adjoint, one lane per edge (as k_geom_bwd)
        const float sq = sqrtf(my_r + 1e-8f), den = sq + norm_constant;
        float gx = 0.f, gy = 0.f, gz = 0.f, gr = my_gr;
        if (dcd) {
            const float4 d = dcd[e0 + lane];
            gx = d.x / den; gy = d.y / den; gz = d.z / den;
            const float dden = -(d.x * dxl + d.y * dyl + d.z * dzl) / (den * den);
            gr += dden * 0.5f / sq;
        }
        gx += 2.0f * dxl * gr; gy += 2.0f * dyl * gr; gz += 2.0f * dzl * gr;
        if (my_i < n_moving) { float* p = dX + (size_t)my_i * 4; atomicAdd(p, gx); atomicAdd(p + 1, gy); atomicAdd(p + 2, gz); }
        if (my_j < n_moving) { float* p = dX + (size_t)my_j * 4; atomicAdd(p, -gx); atomicAdd(p + 1, -gy); atomicAdd(p + 2, -gz); }
    }
    // column sums: per-workgroup partials go to a scratch row (plain stores); k_tail_colsum_reduce adds them up.  (Hundreds
    // of workgroups adding into the same 2H addresses with float atomics serialise at the memory side.)
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[0][wave][lane + 64 * q] = accR[q]; red[1][wave][lane + 64 * q] = accD[q]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * H; idx += 256) {
        const int which = idx / H, c = idx - which * H;
        scratch[((size_t)blockIdx.x * 2 + which) * H + c] = (red[which][0][c] + red[which][1][c]) + (red[which][2][c] + red[which][3][c]);
    }
}
// ------------------------------------------------------------------------------------
// k_dgrad_tail: the data gradient through the second layer of an edge / coordinate MLP AND everything k_edge_tail_bwd does
// with its result, in one kernel: g = (dY W2) * SiLU'(pre1) of a 32-edge tile never leaves the chip (one [E,256] tensor
// less written and read per list and block).  GEMM as k_dgrad_split<32>; the accumulators go to LDS, every wave applies
// SiLU'(pre1) to the eight rows it owns (whole-row reads of pre1) and then runs the tail on the same rows: receiver runs
// into dP, dQ atomics, the radial / d0 column partial sums, dL/d radial and the geometry adjoint - lanes own columns
// lane + 64 q exactly as in k_edge_tail_bwd (contiguous 256-byte atomics).  Scratch layout and reduce kernel are shared.
// ------------------------------------------------------------------------------------
#ifndef CMDGEN_TAIL_EXP
#define CMDGEN_TAIL_EXP 0
#endif
#ifndef CMDGEN_TAIL_STAMPS
#define CMDGEN_TAIL_STAMPS 0
#endif
struct TailArgs {
    const int* row; const int* col; const float* d0; const float* Wcol; int ldw;
    const float4* X; float norm_constant; const float4* dcd; int n_moving;
    float* dP; float* dQ; float* scratch; float* dX;
    unsigned long long* dbg;      // diagnostic build (-DCMDGEN_TAIL_STAMPS=1): per-phase cycle sums [wave][8], [32 + wave] lifetimes, [40] workgroups; else unused
};
template <int NPC>
__global__ __launch_bounds__(256, 3) void k_dgrad_tail(int M, const float* __restrict__ A0, const void* __restrict__ W0,
                                                       const float* __restrict__ pre, TailArgs ta) {
    constexpr int MT = 32, HH = 256, PLDA = SPLIT_PLANE_LDA(HH / 2), PE = MT * PLDA;
    constexpr int LDO = HH + 4, SMEM = 3 * PE * 2 > 32 * LDO * 4 ? 3 * PE * 2 : 32 * LDO * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * MT;
#if CMDGEN_TAIL_STAMPS == 1
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_t;
#define TSTAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_t; st_t = n_; } while (0)
#else
#define TSTAMP(i) do {} while (0)
#endif
    // the tail's per-edge scalars: requested before the GEMM, used after it
    const int e0 = row0 + wave * 8, ne = max(0, min(8, M - e0));
    // (the positions - a second, dependent round trip - are requested after the GEMM, together with the pre1 rows: in front of the GEMM
    // their difference stood in every tile's critical path; profiles/r05_ah_tail_stamps.txt)
    int my_i = -1, my_j = -1; float my_r = 0.f, my_d0 = 0.f, dxl = 0.f, dyl = 0.f, dzl = 0.f;
    if (lane < ne) { my_i = ta.row[e0 + lane]; my_j = ta.col[e0 + lane]; my_d0 = ta.d0[e0 + lane]; }
    float wr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wr[q] = ta.Wcol[(size_t)(lane + 64 * q) * ta.ldw];
    sf32x16 acc[1][2];
    TSTAMP(0);
    dgrad_tile_gemm<MT, NPC>(reinterpret_cast<unsigned short*>(smem), M, row0, A0, W0, nullptr, nullptr, acc);
    TSTAMP(1);
    float* obuf = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            obuf[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDO + wave * 64 + n * 32 + (lane & 31)] = acc[0][n][r];
    __syncthreads();
    TSTAMP(2);
    // SiLU'(pre1) on this wave's rows 8 wave .. 8 wave + 7 (the same rows its tail walks: no barrier in between)
    {
        float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa;
        if (lane < ne) { xa = ta.X[my_i]; xb = ta.X[my_j]; }
        float4 pv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < ne) pv[k] = reinterpret_cast<const float4*>(pre + (size_t)(e0 + k) * HH)[lane];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < ne) {
                float4* cell = reinterpret_cast<float4*>(obuf + (wave * 8 + k) * LDO + 4 * lane);
                float4 x = *cell;
                x.x *= dsilu(pv[k].x); x.y *= dsilu(pv[k].y); x.z *= dsilu(pv[k].z); x.w *= dsilu(pv[k].w);
                *cell = x;
            }
        dxl = xa.x - xb.x; dyl = xa.y - xb.y; dzl = xa.z - xb.z;
        my_r = dxl * dxl + dyl * dyl + dzl * dzl;
    }
    TSTAMP(3);
    float accR[4] = {0.f, 0.f, 0.f, 0.f}, accD[4] = {0.f, 0.f, 0.f, 0.f}, run[4] = {0.f, 0.f, 0.f, 0.f};
    float my_gr = 0.f;
    int cur = __shfl(my_i, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k < ne) {                                        // wave-uniform
            const int i = __shfl(my_i, k), j = __shfl(my_j, k);
            const float r = __shfl(my_r, k), dd = __shfl(my_d0, k);
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = obuf[(wave * 8 + k) * LDO + lane + 64 * q];
            if (i != cur) {                                  // the receiver's run ended: one add per run
#pragma unroll
                for (int q = 0; q < 4; ++q) { atomicAdd(ta.dP + (size_t)cur * HH + lane + 64 * q, run[q]); run[q] = 0.f; }
                cur = i;
            }
            float dot = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                accR[q] += r * v[q]; accD[q] += dd * v[q]; run[q] += v[q]; dot += v[q] * wr[q];
#if CMDGEN_TAIL_EXP != 1
                atomicAdd(ta.dQ + (size_t)j * HH + lane + 64 * q, v[q]);
#endif
            }
            const float gr = wave_sum(dot);                  // dL/d radial of this edge
            if (lane == k) my_gr = gr;
        }
    }
    TSTAMP(4);
    if (ne > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(ta.dP + (size_t)cur * HH + lane + 64 * q, run[q]);
    }
    if (lane < ne && my_i != my_j) {                         // geometry adjoint, one lane per edge (as k_geom_bwd)
        const float sq = sqrtf(my_r + 1e-8f), den = sq + ta.norm_constant;
        float gx = 0.f, gy = 0.f, gz = 0.f, gr = my_gr;
        if (ta.dcd) {
            const float4 d = ta.dcd[e0 + lane];
            gx = d.x / den; gy = d.y / den; gz = d.z / den;
            const float dden = -(d.x * dxl + d.y * dyl + d.z * dzl) / (den * den);
            gr += dden * 0.5f / sq;
        }
        gx += 2.0f * dxl * gr; gy += 2.0f * dyl * gr; gz += 2.0f * dzl * gr;
        if (my_i < ta.n_moving) { float* p = ta.dX + (size_t)my_i * 4; atomicAdd(p, gx); atomicAdd(p + 1, gy); atomicAdd(p + 2, gz); }
        if (my_j < ta.n_moving) { float* p = ta.dX + (size_t)my_j * 4; atomicAdd(p, -gx); atomicAdd(p + 1, -gy); atomicAdd(p + 2, -gz); }
    }
    TSTAMP(5);
    // column sums: per-workgroup partials to the scratch row (k_tail_colsum_reduce adds them up)
    __syncthreads();                                         // every wave is done with the output image
    float* red = obuf;                                       // [2][4][256]
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[(0 * 4 + wave) * 256 + lane + 64 * q] = accR[q]; red[(1 * 4 + wave) * 256 + lane + 64 * q] = accD[q]; }
    __syncthreads();
    for (int idx = tid; idx < 2 * HH; idx += 256) {
        const int which = idx / HH, c = idx - which * HH;
        ta.scratch[((size_t)blockIdx.x * 2 + which) * HH + c] =
            (red[(which * 4 + 0) * 256 + c] + red[(which * 4 + 1) * 256 + c]) + (red[(which * 4 + 2) * 256 + c] + red[(which * 4 + 3) * 256 + c]);
    }
    TSTAMP(6);
#if CMDGEN_TAIL_STAMPS == 1
    if (ta.dbg && lane == 0 && (blockIdx.x & 3) == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&ta.dbg[wave * 8 + i], st_[i]);
        atomicAdd(&ta.dbg[32 + wave], __builtin_amdgcn_s_memtime() - st_begin);
        if (wave == 0) atomicAdd(&ta.dbg[40], 1ull);
    }
#endif
#undef TSTAMP
}


// dWcol[which + c * ldw] += sum over workgroups of scratch[wg][which][c].  grid (H / 64, 2, slices): every workgroup sums
// one slice of the partial rows for 64 columns (4 rows in flight per column) and adds its result with one atomic.
__global__ __launch_bounds__(256) void k_tail_colsum_reduce(int nwg, int H, const float* __restrict__ scratch,
                                                            float* __restrict__ dWcol, int ldw) {
    __shared__ float red[4][64];
    const int which = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int per = (nwg + gridDim.z - 1) / gridDim.z;
    const int w0 = blockIdx.z * per, w1 = min(nwg, w0 + per);
    float sum = 0.f;
    if (c < H) {
#pragma unroll 4
        for (int w = w0 + part; w < w1; w += 4) sum += scratch[((size_t)w * 2 + which) * H + c];
    }
    red[part][threadIdx.x & 63] = sum;
    __syncthreads();
    if (part == 0 && c < H) atomicAdd(dWcol + which + (size_t)c * ldw, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}


// out[c * ldo] += sum_e s[e] * X[e][c]   (s may be null = 1): bias gradients, radial / d0 column gradients, att and
// coordinate-head weight gradients.  One workgroup per 256-row chunk, one column per thread (coalesced rows).
__global__ void k_colsum(int E, int ncols, const float* __restrict__ X, int ldx, const float* __restrict__ s,
                         float* __restrict__ out, int ldo) {
    const int c = threadIdx.x;
    if (c >= ncols) return;
    const int e0 = blockIdx.x * 32, e1 = min(E, e0 + 32);
    float acc = 0.f;
    for (int e = e0; e < e1; ++e) acc += (s ? s[e] : 1.0f) * X[(size_t)e * ldx + c];
    atomicAdd(out + (size_t)c * ldo, acc);
}
// aligned fast path (ncols % 4 == 0, ldx % 4 == 0, 16-byte aligned X): 128 rows per workgroup, a thread owns four
// columns of every fourth row (16-byte loads, independent accumulators), the four row groups are combined in LDS, so a
// workgroup issues ncols atomics for 128 rows instead of ncols per 32 rows (the atomics all hit the same few lines).
__global__ __launch_bounds__(256) void k_colsum4(int E, int ncols, const float* __restrict__ X, int ldx,
                                                 const float* __restrict__ s, float* __restrict__ out, int ldo) {
    __shared__ float4 red[4][64];
    const int cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int e0 = blockIdx.x * 128, e1 = min(E, e0 + 128);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (4 * cg < ncols) {
#pragma unroll 4
        for (int e = e0 + rg; e < e1; e += 4) {
            const float4 v = *reinterpret_cast<const float4*>(X + (size_t)e * ldx + 4 * cg);
            const float w = s ? s[e] : 1.0f;
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
    }
    red[rg][cg] = acc;
    __syncthreads();
    if (rg == 0 && 4 * cg < ncols) {
        const float4 a = red[0][cg], b = red[1][cg], c = red[2][cg], d = red[3][cg];
        float* o = out + (size_t)(4 * cg) * ldo;
        atomicAdd(o, (a.x + b.x) + (c.x + d.x));
        atomicAdd(o + ldo, (a.y + b.y) + (c.y + d.y));
        atomicAdd(o + 2 * (size_t)ldo, (a.z + b.z) + (c.z + d.z));
        atomicAdd(o + 3 * (size_t)ldo, (a.w + b.w) + (c.w + d.w));
    }
}

// v[n] -= mean over the nodes of n's sample (one wave per sample; phar rows then pocket rows).  The projection is
// symmetric, so the same kernel is its own adjoint (applied to the incoming velocity gradient in backward).
__global__ __launch_bounds__(64) void k_center_per_sample(Layout lay, float* __restrict__ v /* [N][4] */) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nl = lay.num_phar[b], np = lay.num_pocket[b], pb = lay.phar_base[b], qb = lay.Nl + lay.pocket_base[b];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = lane; i < nl + np; i += 64) {
        const float* p = v + (size_t)(i < nl ? pb + i : qb + i - nl) * 4;
        sx += p[0]; sy += p[1]; sz += p[2];
    }
    sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
    const float c = fmaxf((float)(nl + np), 1.0f);
    sx /= c; sy /= c; sz /= c;
    for (int i = lane; i < nl + np; i += 64) {
        float* p = v + (size_t)(i < nl ? pb + i : qb + i - nl) * 4;
        p[0] -= sx; p[1] -= sy; p[2] -= sz;
    }
}
// split d_eps [n_rows][3+F] into dvel [N][4] (rows row0..) and ddec [n_rows][F]
__global__ void k_eps_bwd(int n_rows, int F, int row0, const float* __restrict__ deps, float* __restrict__ dvel,
                          float* __restrict__ ddec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * (3 + F)) return;
    const int n = i / (3 + F), k = i - n * (3 + F);
    if (k < 3) dvel[(size_t)(row0 + n) * 4 + k] = deps[i]; else ddec[(size_t)n * F + k - 3] = deps[i];
}

// ------------------------------------------------------------------------------------
// optimizer: AdamW with amsgrad (torch.optim.AdamW semantics, lightning_modules.py:141-143) on the flat buffers,
// with the norm clipping coefficient folded in (clip_grad_norm_: g *= clip when clip < 1).
// ------------------------------------------------------------------------------------
__global__ void k_adamw(size_t n, float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m,
                        float* __restrict__ v, float* __restrict__ vmax, float lr, float beta1, float beta2, float eps,
                        float weight_decay, float bias1, float bias2_sqrt, float clip, const float* __restrict__ sqnorm,
                        float max_norm, int skip_nonfinite) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (sqnorm) {
        const float sq = sqnorm[0];
        // a step whose forward ran on the half engine and whose gradient is not finite (an activation beyond fp16's range) leaves parameters and
        // moments untouched: the host sees the norm, switches the forward to the bf16 engine and repeats the batch (training.HipTrainer)
        if (skip_nonfinite && !(sq <= 3.0e38f)) return;
        if (max_norm > 0.f) clip = fminf(1.0f, max_norm / (sqrtf(sq) + 1e-6f));  // clip_grad_norm_'s coefficient from the device-side norm
    }
    const float g = grad[i] * clip;
    float p = theta[i];
    p *= 1.0f - lr * weight_decay;
    const float mi = m[i] + (g - m[i]) * (1.0f - beta1);           // lerp, as torch
    const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
    const float vm = fmaxf(vmax[i], vi);
    m[i] = mi; v[i] = vi; vmax[i] = vm;
    const float denom = sqrtf(vm) / bias2_sqrt + eps;
    theta[i] = p - (lr / bias1) * (mi / denom);
}
// after a forward on the half engine: a NaN reset of that forward (the evaluation's guard zeroed the velocities - the fp32 reference would not have
// seen a NaN there) makes the step's norm non-finite, so that k_adamw skips the update and the host repeats the batch on the bf16 engine
__global__ void k_norm_guard(float* __restrict__ sq, const int* __restrict__ nan_flag) {
    if (*nan_flag) *sq = __int_as_float(0x7fc00000);
}
void tr_norm_guard(float* sq, const int* nan_flag, hipStream_t s) { hipLaunchKernelGGL(k_norm_guard, dim3(1), dim3(1), 0, s, sq, nan_flag); }
__global__ __launch_bounds__(256) void k_sqsum(size_t n, const float* __restrict__ x, float* __restrict__ out) {
    // one atomic per workgroup: thousands of waves adding into the same address serialise (44 us for 3M elements before)
    __shared__ float red[4];
    float v = 0.f;
    const size_t n4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n / 4 : 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 q = reinterpret_cast<const float4*>(x)[i];
        v += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) v += x[i] * x[i];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

// ------------------------------------------------------------------------------------
// The loss side of the conditional training step (conditional_model.py:198-320 + lightning_modules.py:188-239 of the
// reference) as three launches instead of ~270 small tensor ops: the noising of a batch (normalize, remove_mean_batch,
// noised_representation), the per-sample loss terms together with dL/d eps, and their batch means.  One workgroup per
// sample (a sample has tens of phar nodes and tens to hundreds of pocket nodes); the per-sample scalars that depend
// only on t and on the node counts (alpha_t, sigma_t, SNR weight, the normalisation constants, log p(N)) come from the
// host in `tab`, column-major [TT_COLS][B].
// ------------------------------------------------------------------------------------
enum { TT_ALPHA_T = 0, TT_SIGMA_T, TT_T0, TT_SNRW, TT_ALPHA_TT, TT_SIGMA_TT, TT_NEGLOGC, TT_DLOGPX, TT_LOGPN, TT_TINT, TT_T, TT_S0CAT, TT_COLS };
enum { TS_NLL = 0, TS_ERR_T, TS_LOSS_0, TS_KL, TS_ABS_X, TS_ABS_H, TS_LOSS_0X, TS_LOSS_0H, TS_LOSS_T, TS_COLS = 12 };

// sum of up to three per-thread values over a 256-thread workgroup; every thread gets the totals
__device__ __forceinline__ void wg_sum3(float& a, float& b, float& c, float* red /* [12] */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[w] = a; red[4 + w] = b; red[8 + w] = c; }
    __syncthreads();
    a = (red[0] + red[1]) + (red[2] + red[3]); b = (red[4] + red[5]) + (red[6] + red[7]); c = (red[8] + red[9]) + (red[10] + red[11]);
}

// z_t, the centred pocket and the two sums of the prior KL term of every sample
__global__ __launch_bounds__(256) void k_train_noise(Layout lay, Dims d, const float* __restrict__ px, const float* __restrict__ poh,
                                                     const float* __restrict__ qx, const float* __restrict__ qoh,
                                                     const float* __restrict__ tab, const float* __restrict__ eps,
                                                     float* __restrict__ z_t, float* __restrict__ xh_pocket, float* __restrict__ klsum) {
    __shared__ float red[12];
    const int b = blockIdx.x, B = lay.B, tid = threadIdx.x;
    const int n = lay.num_phar[b], base = lay.phar_base[b], m = lay.num_pocket[b], qbase = lay.pocket_base[b];
    const int P = d.P, R = d.R, ldp = 3 + P, ldq = 3 + R;
    const float alpha = tab[TT_ALPHA_T * B + b], sigma = tab[TT_SIGMA_T * B + b], alphaT = tab[TT_ALPHA_TT * B + b];
    const float cnt = (float)max(n, 1);
    // phar centre of mass of the normalised coordinates (remove_mean_batch #1); SimpleConditionalDDPM (d.no_com): the POCKET's centre
    // of mass of the raw coordinates, subtracted before normalisation (conditional_model.py:481-525), and no projection afterwards
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (d.no_com) {
        for (int i = tid; i < m; i += 256) { const float* x = qx + (size_t)(qbase + i) * 3; sx += x[0]; sy += x[1]; sz += x[2]; }
    } else {
        for (int i = tid; i < n; i += 256) {
            const float* x = px + (size_t)(base + i) * 3;
            sx += x[0] / d.norm_x; sy += x[1] / d.norm_x; sz += x[2] / d.norm_x;
        }
    }
    wg_sum3(sx, sy, sz, red);
    const float cq = (float)max(m, 1);
    const float m1x = d.no_com ? sx / cq : sx / cnt, m1y = d.no_com ? sy / cq : sy / cnt, m1z = d.no_com ? sz / cq : sz / cnt;
    // (conditional: m1 is in normalised units and subtracted after the division; simple: raw units, subtracted before it)
#define TN_X(v, mm) (d.no_com ? ((v) - (mm)) / d.norm_x : (v) / d.norm_x - (mm))
    // z_t = alpha xh0 + sigma eps (x part still with its mean), the KL sums of alpha_T xh0
    float zx = 0.f, zy = 0.f, zz = 0.f, klx = 0.f, klh = 0.f, dummy = 0.f;
    for (int i = tid; i < n; i += 256) {
        const size_t g = (size_t)(base + i);
        const float* x = px + g * 3; const float* e = eps + g * ldp; float* z = z_t + g * ldp;
        const float x0 = TN_X(x[0], m1x), x1 = TN_X(x[1], m1y), x2 = TN_X(x[2], m1z);
        const float a0 = alphaT * x0, a1 = alphaT * x1, a2 = alphaT * x2;
        klx += a0 * a0 + a1 * a1 + a2 * a2;
        const float z0 = alpha * x0 + sigma * e[0], z1 = alpha * x1 + sigma * e[1], z2 = alpha * x2 + sigma * e[2];
        z[0] = z0; z[1] = z1; z[2] = z2; zx += z0; zy += z1; zz += z2;
        for (int c = 0; c < P; ++c) {
            const float h = (poh[g * P + c] - d.bias_h) / d.norm_h, ah = alphaT * h;
            klh += ah * ah;
            z[3 + c] = alpha * h + sigma * e[3 + c];
        }
    }
    wg_sum3(zx, zy, zz, red);
    wg_sum3(klx, klh, dummy, red);
    const float m2x = d.no_com ? 0.f : zx / cnt, m2y = d.no_com ? 0.f : zy / cnt, m2z = d.no_com ? 0.f : zz / cnt;
    if (!d.no_com)
        for (int i = tid; i < n; i += 256) {        // remove_mean_batch #2 (each thread revisits the rows it wrote)
            float* z = z_t + (size_t)(base + i) * ldp;
            z[0] -= m2x; z[1] -= m2y; z[2] -= m2z;
        }
    for (int i = tid; i < m; i += 256) {
        const size_t g = (size_t)(qbase + i);
        const float* x = qx + g * 3; float* o = xh_pocket + g * ldq;
        o[0] = TN_X(x[0], m1x) - m2x; o[1] = TN_X(x[1], m1y) - m2y; o[2] = TN_X(x[2], m1z) - m2z;
        for (int c = 0; c < R; ++c) o[3 + c] = (qoh[g * R + c] - d.bias_h) / d.norm_h;
    }
    if (tid == 0) { klsum[2 * b] = klx; klsum[2 * b + 1] = klh; }
#undef TN_X
}

__device__ __forceinline__ float cdf_std_gauss(float x) { return 0.5f * (1.0f + erff(x / 1.41421356237309515f)); }

// per-sample loss terms and dL/d net_out.  l2: the 'l2' training objective (lightning_modules.py:198-205), else the vlb
// weighting (:206-212); both in training mode (t = 0 handled by the t_is_zero masks, conditional_model.py:265-272).
__global__ __launch_bounds__(256) void k_train_loss(Layout lay, Dims d, int l2, float T, const float* __restrict__ net,
                                                    const float* __restrict__ eps, const float* __restrict__ z_t,
                                                    const float* __restrict__ poh, const float* __restrict__ tab,
                                                    const float* __restrict__ klsum, float* __restrict__ terms,
                                                    float* __restrict__ d_eps) {
    __shared__ float red[12];
    const int b = blockIdx.x, B = lay.B, tid = threadIdx.x;
    const int n = lay.num_phar[b], base = lay.phar_base[b];
    const int P = d.P, ldp = 3 + P;
    const float t0 = tab[TT_T0 * B + b], snrw = tab[TT_SNRW * B + b], s0cat = tab[TT_S0CAT * B + b];
    const float nf = (float)n;
    const float s_t = l2 ? 1.0f / ((float)(3 + P) * nf) : -T * snrw, s_0 = l2 ? 1.0f / (3.0f * nf) : 1.0f;
    const float w_t = (1.0f - t0) * s_t / (float)B, w_0 = t0 * s_0 / (float)B;
    float ex = 0.f, eh = 0.f, lph = 0.f, ax = 0.f, ah = 0.f, dummy = 0.f;
    for (int i = tid; i < n; i += 256) {
        const size_t g = (size_t)(base + i);
        const float* o = net + g * ldp; const float* e = eps + g * ldp; float* de = d_eps + g * ldp;
        float sabs = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float df = o[c] - e[c];
            ex += df * df; sabs += fabsf(o[c]);
            de[c] = df * w_t + df * w_0;
        }
        ax += sabs / 3.0f;
        sabs = 0.f;
        for (int c = 0; c < P; ++c) {
            const float df = o[3 + c] - e[3 + c];
            eh += df * df; sabs += fabsf(o[3 + c]);
            de[3 + c] = df * w_t;
        }
        ah += sabs / (float)P;
        // log p(h | z_0): integrate the normal around each class over the unit bin, normalise over the classes
        const float* z = z_t + g * ldp + 3;
        float mx = -INFINITY;
        for (int c = 0; c < P; ++c) {
            const float ctr = z[c] * d.norm_h + d.bias_h - 1.0f;
            const float lp = logf(cdf_std_gauss((ctr + 0.5f) / s0cat) - cdf_std_gauss((ctr - 0.5f) / s0cat) + 1e-10f);
            mx = fmaxf(mx, lp);
        }
        float se = 0.f, dot = 0.f, ohs = 0.f;
        for (int c = 0; c < P; ++c) {
            const float ctr = z[c] * d.norm_h + d.bias_h - 1.0f;
            const float lp = logf(cdf_std_gauss((ctr + 0.5f) / s0cat) - cdf_std_gauss((ctr - 0.5f) / s0cat) + 1e-10f);
            se += expf(lp - mx);
            const float oh = ((poh[g * P + c] - d.bias_h) / d.norm_h) * d.norm_h + d.bias_h;
            dot += lp * oh; ohs += oh;
        }
        lph += dot - (mx + logf(se)) * ohs;
    }
    wg_sum3(ex, eh, lph, red);
    wg_sum3(ax, ah, dummy, red);
    if (tid == 0) {
        const float alphaT = tab[TT_ALPHA_TT * B + b], sigT = tab[TT_SIGMA_TT * B + b];
        (void)alphaT;
        const float dsub = (d.no_com ? nf : nf - 1.0f) * 3.0f;      // subspace_dimensionality (SimpleConditionalDDPM: no projection)
        // gaussian_KL(|mu|^2, sigma_T, 1, dim) = dim log(1 / sigma_T) + 0.5 (dim sigma_T^2 + |mu|^2) - 0.5 dim
        const float kl_h = logf(1.0f / sigT) + 0.5f * (sigT * sigT + klsum[2 * b + 1]) - 0.5f;
        const float kl_x = dsub * logf(1.0f / sigT) + 0.5f * (dsub * sigT * sigT + klsum[2 * b]) / 1.0f - 0.5f * dsub;
        const float kl = kl_x + kl_h;
        float err_t = (ex + eh) * (1.0f - t0);
        const float loss_0x = 0.5f * ex * t0, loss_0h = -lph * t0;
        float loss_t, loss_0, nll;
        if (l2) {
            err_t = err_t / ((float)(3 + P) * nf);
            loss_t = 0.5f * err_t;
            loss_0 = loss_0x / (3.0f * nf) + loss_0h;
            nll = loss_t + loss_0 + kl;
        } else {
            loss_t = -T * 0.5f * snrw * err_t;
            loss_0 = loss_0x + loss_0h + tab[TT_NEGLOGC * B + b];
            nll = loss_t + loss_0 + kl - tab[TT_DLOGPX * B + b] - tab[TT_LOGPN * B + b];
        }
        const float cnt = (float)max(n, 1);
        float* o = terms + (size_t)b * TS_COLS;
        o[TS_NLL] = nll; o[TS_ERR_T] = err_t; o[TS_LOSS_0] = loss_0; o[TS_KL] = kl; o[TS_ABS_X] = ax / cnt; o[TS_ABS_H] = ah / cnt;
        o[TS_LOSS_0X] = loss_0x; o[TS_LOSS_0H] = loss_0h; o[TS_LOSS_T] = loss_t;
    }
}

// ------------------------------------------------------------------------------------
// The same two launches for the JOINT model (EnVariationalDiffusion.forward in training mode, en_diffusion.py:332-465, and the
// joint branch of lightning_modules.py:198-217): pocket nodes are noised and denoised too.  One workgroup per sample.
//   k_train_noise_joint: normalize; the x-part of the draw loses its centre of mass over ALL nodes of the sample
//     (sample_combined_position_feature_noise, :555-574); z = alpha_t xh + sigma_t eps for both parts; the prior-KL sums over both.
//   k_train_loss_joint: error_t, L0 (x of both parts, the categorical likelihood of both one-hot blocks), kl_prior with
//     (n_phar + n_pocket - 1) * 3 degrees of freedom, the 'l2' / vlb combination, dL/d net_out for both outputs.
// terms columns 0..8 as in k_train_loss (error_t / |eps_hat| of the phar part), 9 error_t of the pocket part, 10 / 11 mean
// |eps_hat| of the pocket's x / h.
// ------------------------------------------------------------------------------------
enum { TS_ERR_T_Q = 9, TS_ABS_X_Q = 10, TS_ABS_H_Q = 11 };

__global__ __launch_bounds__(256) void k_train_noise_joint(Layout lay, Dims d, const float* __restrict__ px, const float* __restrict__ poh,
                                                           const float* __restrict__ qx, const float* __restrict__ qoh,
                                                           const float* __restrict__ tab, const float* __restrict__ raw_l,
                                                           const float* __restrict__ raw_q, float* __restrict__ z_l, float* __restrict__ z_q,
                                                           float* __restrict__ e_l, float* __restrict__ e_q, float* __restrict__ klsum) {
    __shared__ float red[12];
    const int b = blockIdx.x, B = lay.B, tid = threadIdx.x;
    const int n = lay.num_phar[b], base = lay.phar_base[b], m = lay.num_pocket[b], qbase = lay.pocket_base[b];
    const int P = d.P, R = d.R, ldp = 3 + P, ldq = 3 + R;
    const float alpha = tab[TT_ALPHA_T * B + b], sigma = tab[TT_SIGMA_T * B + b], alphaT = tab[TT_ALPHA_TT * B + b];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = tid; i < n; i += 256) { const float* e = raw_l + (size_t)(base + i) * ldp; sx += e[0]; sy += e[1]; sz += e[2]; }
    for (int i = tid; i < m; i += 256) { const float* e = raw_q + (size_t)(qbase + i) * ldq; sx += e[0]; sy += e[1]; sz += e[2]; }
    wg_sum3(sx, sy, sz, red);
    const float cnt = (float)max(n + m, 1);
    const float mx = sx / cnt, my = sy / cnt, mz = sz / cnt;
    float klx = 0.f, klh = 0.f, dummy = 0.f;
    auto part = [&](int rows, int row0, int C, int ld, const float* x3, const float* oh, const float* raw, float* z, float* eo) {
        for (int i = tid; i < rows; i += 256) {
            const size_t g = (size_t)(row0 + i);
            const float* x = x3 + g * 3; const float* e = raw + g * ld; float* zo = z + g * ld; float* ee = eo + g * ld;
            const float en[3] = {e[0] - mx, e[1] - my, e[2] - mz};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float xn = x[c] / d.norm_x, a = alphaT * xn;
                klx += a * a;
                ee[c] = en[c];
                zo[c] = alpha * xn + sigma * en[c];
            }
            for (int c = 0; c < C; ++c) {
                const float hn = (oh[g * C + c] - d.bias_h) / d.norm_h, a = alphaT * hn;
                klh += a * a;
                ee[3 + c] = e[3 + c];
                zo[3 + c] = alpha * hn + sigma * e[3 + c];
            }
        }
    };
    part(n, base, P, ldp, px, poh, raw_l, z_l, e_l);
    part(m, qbase, R, ldq, qx, qoh, raw_q, z_q, e_q);
    wg_sum3(klx, klh, dummy, red);
    if (tid == 0) { klsum[2 * b] = klx; klsum[2 * b + 1] = klh; }
}

// log p(h | z_0) of one node (en_diffusion.py:298-326): the normal around each class integrated over the unit bin, normalised over the
// classes, dotted with the node's one-hot row.  z: the node's normalised feature block; oh: its raw one-hot row.
__device__ __forceinline__ float log_ph_row(const float* z, const float* oh, int C, const Dims& d, float s0cat) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) {
        const float ctr = z[c] * d.norm_h + d.bias_h - 1.0f;
        mx = fmaxf(mx, logf(cdf_std_gauss((ctr + 0.5f) / s0cat) - cdf_std_gauss((ctr - 0.5f) / s0cat) + 1e-10f));
    }
    float se = 0.f, dot = 0.f, ohs = 0.f;
    for (int c = 0; c < C; ++c) {
        const float ctr = z[c] * d.norm_h + d.bias_h - 1.0f;
        const float lp = logf(cdf_std_gauss((ctr + 0.5f) / s0cat) - cdf_std_gauss((ctr - 0.5f) / s0cat) + 1e-10f);
        se += expf(lp - mx);
        const float o = ((oh[c] - d.bias_h) / d.norm_h) * d.norm_h + d.bias_h;
        dot += lp * o; ohs += o;
    }
    return dot - (mx + logf(se)) * ohs;
}

__global__ __launch_bounds__(256) void k_train_loss_joint(Layout lay, Dims d, int l2, float T, const float* __restrict__ net_l,
                                                          const float* __restrict__ net_q, const float* __restrict__ e_l,
                                                          const float* __restrict__ e_q, const float* __restrict__ z_l,
                                                          const float* __restrict__ z_q, const float* __restrict__ poh,
                                                          const float* __restrict__ qoh, const float* __restrict__ tab,
                                                          const float* __restrict__ klsum, float* __restrict__ terms,
                                                          float* __restrict__ d_l, float* __restrict__ d_q) {
    __shared__ float red[12];
    const int b = blockIdx.x, B = lay.B, tid = threadIdx.x;
    const int n = lay.num_phar[b], base = lay.phar_base[b], m = lay.num_pocket[b], qbase = lay.pocket_base[b];
    const int P = d.P, R = d.R;
    const float t0 = tab[TT_T0 * B + b], snrw = tab[TT_SNRW * B + b], s0cat = tab[TT_S0CAT * B + b];
    const float nf = (float)n, mf = (float)m;
    float sums[2][5];                      // [part][x error, h error, log p(h), |eps_hat_x|, |eps_hat_h|]
    auto part = [&](int which, int rows, int row0, int C, float cntf, const float* net, const float* eps, const float* z, const float* oh, float* de) {
        const int ld = 3 + C;
        const float s_t = l2 ? 1.0f / ((float)(3 + C) * cntf) : -T * snrw, s_0 = l2 ? 1.0f / (3.0f * cntf) : 1.0f;
        const float w_t = (1.0f - t0) * s_t / (float)B, w_0 = t0 * s_0 / (float)B;
        float ex = 0.f, eh = 0.f, lph = 0.f, ax = 0.f, ah = 0.f, dummy = 0.f;
        for (int i = tid; i < rows; i += 256) {
            const size_t g = (size_t)(row0 + i);
            const float* o = net + g * ld; const float* e = eps + g * ld; float* dd = de + g * ld;
            float sabs = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float df = o[c] - e[c];
                ex += df * df; sabs += fabsf(o[c]);
                dd[c] = df * w_t + df * w_0;
            }
            ax += sabs / 3.0f;
            sabs = 0.f;
            for (int c = 0; c < C; ++c) {
                const float df = o[3 + c] - e[3 + c];
                eh += df * df; sabs += fabsf(o[3 + c]);
                dd[3 + c] = df * w_t;
            }
            ah += sabs / (float)C;
            lph += log_ph_row(z + g * ld + 3, oh + g * C, C, d, s0cat);
        }
        wg_sum3(ex, eh, lph, red);
        wg_sum3(ax, ah, dummy, red);
        sums[which][0] = ex; sums[which][1] = eh; sums[which][2] = lph; sums[which][3] = ax; sums[which][4] = ah;
    };
    part(0, n, base, P, nf, net_l, e_l, z_l, poh, d_l);
    part(1, m, qbase, R, mf, net_q, e_q, z_q, qoh, d_q);
    if (tid == 0) {
        const float sigT = tab[TT_SIGMA_TT * B + b];
        const float dsub = (nf + mf - 1.0f) * 3.0f;
        const float kl_h = logf(1.0f / sigT) + 0.5f * (sigT * sigT + klsum[2 * b + 1]) - 0.5f;
        const float kl_x = dsub * logf(1.0f / sigT) + 0.5f * (dsub * sigT * sigT + klsum[2 * b]) / 1.0f - 0.5f * dsub;
        const float kl = kl_x + kl_h;
        float err_l = (sums[0][0] + sums[0][1]) * (1.0f - t0), err_q = (sums[1][0] + sums[1][1]) * (1.0f - t0);
        const float l0x_l = 0.5f * sums[0][0] * t0, l0x_q = 0.5f * sums[1][0] * t0, l0h = -(sums[0][2] + sums[1][2]) * t0;
        float loss_t, loss_0, nll;
        if (l2) {
            err_l = err_l / ((float)(3 + P) * nf); err_q = err_q / ((float)(3 + R) * mf);
            loss_t = 0.5f * (err_l + err_q);
            loss_0 = l0x_l / (3.0f * nf) + l0x_q / (3.0f * mf) + l0h;
            nll = loss_t + loss_0 + kl;
        } else {
            loss_t = -T * 0.5f * snrw * (err_l + err_q);
            loss_0 = l0x_l + l0x_q + l0h + tab[TT_NEGLOGC * B + b];
            nll = loss_t + loss_0 + kl - tab[TT_DLOGPX * B + b] - tab[TT_LOGPN * B + b];
        }
        float* o = terms + (size_t)b * TS_COLS;
        o[TS_NLL] = nll; o[TS_ERR_T] = err_l; o[TS_LOSS_0] = loss_0; o[TS_KL] = kl;
        o[TS_ABS_X] = sums[0][3] / (float)max(n, 1); o[TS_ABS_H] = sums[0][4] / (float)max(n, 1);
        o[TS_LOSS_0X] = l0x_l + l0x_q; o[TS_LOSS_0H] = l0h; o[TS_LOSS_T] = loss_t;
        o[TS_ERR_T_Q] = err_q; o[TS_ABS_X_Q] = sums[1][3] / (float)max(m, 1); o[TS_ABS_H_Q] = sums[1][4] / (float)max(m, 1);
    }
}

// means over the batch of every per-sample term column (one workgroup; B is at most a few thousand)
__global__ __launch_bounds__(256) void k_train_means(int B, const float* __restrict__ terms, float* __restrict__ means) {
    __shared__ float red[12];
    for (int c = 0; c < TS_COLS; c += 3) {
        float a = 0.f, b2 = 0.f, c2 = 0.f;
        for (int i = threadIdx.x; i < B; i += 256) {
            const float* o = terms + (size_t)i * TS_COLS + c;
            a += o[0]; b2 += o[1]; c2 += o[2];
        }
        wg_sum3(a, b2, c2, red);
        if (threadIdx.x == 0) { means[c] = a / (float)B; means[c + 1] = b2 / (float)B; means[c + 2] = c2 / (float)B; }
    }
}

void tr_noise(const Layout& lay, const Dims& d, const float* px, const float* poh, const float* qx, const float* qoh, const float* tab,
              const float* eps, float* z_t, float* xh_pocket, float* klsum, hipStream_t s) {
    hipLaunchKernelGGL(k_train_noise, dim3(lay.B), dim3(256), 0, s, lay, d, px, poh, qx, qoh, tab, eps, z_t, xh_pocket, klsum);
}
void tr_loss(const Layout& lay, const Dims& d, int l2, float T, const float* net, const float* eps, const float* z_t, const float* poh,
             const float* tab, const float* klsum, float* terms, float* d_eps, float* means, hipStream_t s) {
    hipLaunchKernelGGL(k_train_loss, dim3(lay.B), dim3(256), 0, s, lay, d, l2, T, net, eps, z_t, poh, tab, klsum, terms, d_eps);
    hipLaunchKernelGGL(k_train_means, dim3(1), dim3(256), 0, s, lay.B, terms, means);
}

void tr_noise_joint(const Layout& lay, const Dims& d, const float* px, const float* poh, const float* qx, const float* qoh, const float* tab,
                    const float* raw_l, const float* raw_q, float* z_l, float* z_q, float* e_l, float* e_q, float* klsum, hipStream_t s) {
    hipLaunchKernelGGL(k_train_noise_joint, dim3(lay.B), dim3(256), 0, s, lay, d, px, poh, qx, qoh, tab, raw_l, raw_q, z_l, z_q, e_l, e_q, klsum);
}
void tr_loss_joint(const Layout& lay, const Dims& d, int l2, float T, const float* net_l, const float* net_q, const float* e_l, const float* e_q,
                   const float* z_l, const float* z_q, const float* poh, const float* qoh, const float* tab, const float* klsum, float* terms,
                   float* d_l, float* d_q, float* means, hipStream_t s) {
    hipLaunchKernelGGL(k_train_loss_joint, dim3(lay.B), dim3(256), 0, s, lay, d, l2, T, net_l, net_q, e_l, e_q, z_l, z_q, poh, qoh, tab, klsum,
                       terms, d_l, d_q);
    hipLaunchKernelGGL(k_train_means, dim3(1), dim3(256), 0, s, lay.B, terms, means);
}

// ------------------------------------------------------------------------------------
// launch helpers (C++ linkage, used by cmdgen_train.hip)
// ------------------------------------------------------------------------------------
#define EW_GRID(n) dim3((unsigned)(((size_t)(n) + 255) / 256)), dim3(256)
#define ROW_GRID(E) dim3((unsigned)(((E) + 3) / 4)), dim3(256)        // one wave per row, 4 rows per workgroup

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
void tr_silu_bwd(float* g, const float* pre, size_t n, hipStream_t s) {
    if (!n) return;
    if (n % 4 == 0 && al16(g) && al16(pre)) hipLaunchKernelGGL(k_silu_bwd4, EW_GRID(n / 4), 0, s, (float4*)g, (const float4*)pre, n / 4);
    else hipLaunchKernelGGL(k_silu_bwd, EW_GRID(n), 0, s, g, pre, n);
}
void tr_scale(float* x, float d, size_t n, hipStream_t s) { if (n) hipLaunchKernelGGL(k_scale, EW_GRID(n), 0, s, x, d, n); }
void tr_scale_rows(float* x, const float* div, int H, size_t n, hipStream_t s) { if (n) hipLaunchKernelGGL(k_scale_rows, EW_GRID(n), 0, s, x, div, H, n); }
// The two reductions of an edge list in one launch: blockIdx.y = 0 is k_partial_reduce's job (gate / head partial sums,
// scratch rows of H + 4 floats), blockIdx.y = 1, 2 are k_tail_colsum_reduce's (radial / d0 column partials, rows of 2 H).
__global__ __launch_bounds__(256) void k_reduce_pair(int nwg_a, int H, const float* __restrict__ scratch_a, float* __restrict__ out_w,
                                                     float* __restrict__ out_b, int nwg_t, const float* __restrict__ scratch_t,
                                                     float* __restrict__ dWcol, int ldw) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const bool job_a = blockIdx.y == 0;
    if (job_a && !out_w) return;                 // (a model without attention has no gate parameters)
    const int which = (int)blockIdx.y - 1;
    const int nwg = job_a ? nwg_a : nwg_t;
    const int ncol = job_a ? H + (out_b ? 1 : 0) : H;
    const int per = (nwg + gridDim.z - 1) / gridDim.z;
    const int w0 = blockIdx.z * per, w1 = min(nwg, w0 + per);
    float sum = 0.f;
    if (c < ncol) {
        if (job_a) {
#pragma unroll 4
            for (int w = w0 + part; w < w1; w += 4) sum += scratch_a[(size_t)w * (H + 4) + c];
        } else {
#pragma unroll 4
            for (int w = w0 + part; w < w1; w += 4) sum += scratch_t[((size_t)w * 2 + which) * H + c];
        }
    }
    red[part][threadIdx.x & 63] = sum;
    __syncthreads();
    if (part == 0 && c < ncol) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (job_a) atomicAdd(c < H ? out_w + c : out_b, t);
        else atomicAdd(dWcol + which + (size_t)c * ldw, t);
    }
}
void tr_reduce_pair(int E, int H, const float* scratch_a, float* out_w, float* out_b, const float* scratch_t, float* dWcol, int ldw, hipStream_t s) {
    if (E <= 0) return;
    const int nwg = (E + 31) / 32;               // both producers take 32 edges per workgroup
    hipLaunchKernelGGL(k_reduce_pair, dim3((H + 1 + 63) / 64, 3, min(32, (nwg + 15) / 16)), dim3(256), 0, s, nwg, H, scratch_a, out_w, out_b,
                       nwg, scratch_t, dWcol, ldw);
}
size_t tr_partial_scratch_floats(size_t E, size_t H) { return ((E + 4 * GATE_EPW - 1) / (4 * GATE_EPW)) * (H + 4); }
// the attention gate's adjoint with its two parameter gradients (d_wa [H], d_ba [1]; ignored without attention)
void tr_gate_bwd(int E, int H, const int* row, const float* pre2, const float* wa, const float* z, int attention, const float* dagg,
                 float* dpre2, float* scratch, float* d_wa, float* d_ba, float* zero, size_t zero_floats, hipStream_t s,
                 bool defer_reduce = false) {       // defer_reduce: the caller adds the partial sums up later (tr_reduce_pair)
    if (!E) { if (zero_floats) hipMemsetAsync(zero, 0, zero_floats * sizeof(float), s); return; }
    const int nwg = (E + 4 * GATE_EPW - 1) / (4 * GATE_EPW);
    hipLaunchKernelGGL(k_gate_bwd, dim3(nwg), dim3(256), 0, s, E, H, row, pre2, wa, z, attention, dagg, dpre2, scratch, (float4*)zero, zero_floats / 4);
    if (attention && !defer_reduce) hipLaunchKernelGGL(k_partial_reduce, dim3((H + 1 + 63) / 64, min(32, (nwg + 15) / 16)), dim3(256), 0, s, nwg, H, scratch, d_wa, d_ba);
}
// dpre7 from dphi, and d coord_mlp.4.weight
void tr_head_bwd(int E, int H, const float* dphi, const float* w5, const float* pre7, float* dpre7, float* scratch, float* d_w5,
                 float* zero, size_t zero_floats, hipStream_t s, bool defer_reduce = false, const CoordOutArgs* co = nullptr) {
    if (!E) { if (zero_floats) hipMemsetAsync(zero, 0, zero_floats * sizeof(float), s); return; }
    const int nwg = (E + 4 * GATE_EPW - 1) / (4 * GATE_EPW);
    hipLaunchKernelGGL(k_head_bwd, dim3(nwg), dim3(256), 0, s, E, H, dphi, w5, pre7, dpre7, scratch, (float4*)zero, zero_floats / 4, co ? *co : CoordOutArgs{});
    if (!defer_reduce) hipLaunchKernelGGL(k_partial_reduce, dim3((H + 63) / 64, min(32, (nwg + 15) / 16)), dim3(256), 0, s, nwg, H, scratch, d_w5, (float*)nullptr);
}
void tr_coord_out_bwd(int E, const int* row, const int* col, const float4* X, const float* phi, int use_tanh, float range,
                      float nc, const float* dacc, float dacc_div, int n_moving, float* dphi, float4* dcd, hipStream_t s, const float* adiv = nullptr) {
    if (E) hipLaunchKernelGGL(k_coord_out_bwd, EW_GRID(E), 0, s, E, row, col, X, phi, use_tanh, range, nc, dacc, dacc_div, n_moving, dphi, dcd, adiv);
}
void tr_edge_tail_bwd(int E, int H, const int* row, const int* col, const float* g, const float* d0, const float* Wcol, int ldw,
                      const float4* X, float nc, const float4* dcd, int n_moving, float* dP, float* dQ, float* dWcol, float* dX,
                      float* scratch, hipStream_t s) {
    if (!E) return;
    const int nwg = (E + 4 * TAIL_EPW - 1) / (4 * TAIL_EPW);
    hipLaunchKernelGGL(k_edge_tail_bwd, dim3(nwg), dim3(256), 0, s, E, H, row, col, g, d0, Wcol, ldw, X, nc, dcd, n_moving, dP, dQ,
                       scratch, dX);
    hipLaunchKernelGGL(k_tail_colsum_reduce, dim3((H + 63) / 64, 2, min(32, (nwg + 15) / 16)), dim3(256), 0, s, nwg, H, scratch, dWcol, ldw);
}
// fused: g = (dY W2^T-pack) * SiLU'(pre1) and its whole tail (H = 256; W = split pack of the transposed weight)
void cmdgen_dgrad_tail(int E, const float* dY, const void* Wt, const float* pre1, const int* row, const int* col, const float* d0,
                       const float* Wcol, int ldw, const float4* X, float nc, const float4* dcd, int n_moving, float* dP, float* dQ,
                       float* dWcol, float* dX, float* scratch, int pieces, hipStream_t s, bool defer_reduce = false) {
    if (E <= 0) return;
    const int nwg = (E + 31) / 32;
    const TailArgs ta{row, col, d0, Wcol, ldw, X, nc, dcd, n_moving, dP, dQ, scratch, dX, g_train_tune.dbg};
    if (pieces == 3) hipLaunchKernelGGL((k_dgrad_tail<3>), dim3(nwg), dim3(256), 0, s, E, dY, Wt, pre1, ta);
    else hipLaunchKernelGGL((k_dgrad_tail<1>), dim3(nwg), dim3(256), 0, s, E, dY, Wt, pre1, ta);
    if (!defer_reduce) hipLaunchKernelGGL(k_tail_colsum_reduce, dim3(4, 2, min(32, (nwg + 15) / 16)), dim3(256), 0, s, nwg, 256, scratch, dWcol, ldw);
}
size_t tr_edge_tail_scratch_floats(size_t E, size_t H) { return ((E + 4 * TAIL_EPW - 1) / (4 * TAIL_EPW)) * 2 * H; }
void tr_colsum(int E, int ncols, const float* X, int ldx, const float* sv, float* out, int ldo, hipStream_t s) {
    if (!E) return;
    if (ncols % 4 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && ncols <= 256)
        hipLaunchKernelGGL(k_colsum4, dim3((E + 127) / 128), dim3(256), 0, s, E, ncols, X, ldx, sv, out, ldo);
    else
        hipLaunchKernelGGL(k_colsum, dim3((E + 31) / 32), dim3(256), 0, s, E, ncols, X, ldx, sv, out, ldo);
}
void tr_center_per_sample(const Layout& lay, float* v, hipStream_t s) {
    hipLaunchKernelGGL(k_center_per_sample, dim3(lay.B), dim3(64), 0, s, lay, v);
}
void tr_eps_bwd(int n_rows, int F, int row0, const float* deps, float* dvel, float* ddec, hipStream_t s) {
    if (n_rows) hipLaunchKernelGGL(k_eps_bwd, EW_GRID((size_t)n_rows * (3 + F)), 0, s, n_rows, F, row0, deps, dvel, ddec);
}
// start of the conditional model's backward pass in one launch instead of three fills and k_eps_bwd: dX <- the velocity part of d_eps on the phar
// rows and zero elsewhere (pocket rows do not move), ddec <- its feature part, dhfin <- 0
__global__ void k_bwd_init(int Nl, int N, int P, int dyn, const float* __restrict__ deps, float* __restrict__ dX, float* __restrict__ ddec,
                           float* __restrict__ dhfin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * 4) { const int n = i >> 2, k = i & 3; dX[i] = (n < Nl && k < 3) ? deps[(size_t)n * (3 + P) + k] : 0.f; }
    if (i < Nl * P) { const int n = i / P, k = i - n * P; ddec[i] = deps[(size_t)n * (3 + P) + 3 + k]; }
    if (i < N * dyn) dhfin[i] = 0.f;
}
void tr_bwd_init(int Nl, int N, int P, int dyn, const float* deps, float* dX, float* ddec, float* dhfin, hipStream_t s) {
    const size_t n = (size_t)N * (dyn > 4 ? dyn : 4) > (size_t)Nl * P ? (size_t)N * (dyn > 4 ? dyn : 4) : (size_t)Nl * P;
    if (n) hipLaunchKernelGGL(k_bwd_init, EW_GRID(n), 0, s, Nl, N, P, dyn, deps, dX, ddec, dhfin);
}
void tr_adamw(size_t n, float* theta, const float* grad, float* m, float* v, float* vmax, float lr, float b1, float b2,
              float eps, float wd, float bias1, float bias2_sqrt, float clip, hipStream_t s, const float* sqnorm = nullptr,
              float max_norm = 0.f, int skip_nonfinite = 0) {
    if (n) hipLaunchKernelGGL(k_adamw, EW_GRID(n), 0, s, n, theta, grad, m, v, vmax, lr, b1, b2, eps, wd, bias1, bias2_sqrt, clip, sqnorm, max_norm, skip_nonfinite);
}
void tr_sqsum(size_t n, const float* x, float* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_sqsum, dim3((unsigned)(n / 8192 + 1 > 512 ? 512 : n / 8192 + 1)), dim3(256), 0, s, n, x, out);
}
