// kernels_joint.hip - the sampler arithmetic of the JOINT model around the network evaluation:
// EnVariationalDiffusion.sample (en_diffusion.py:576-647) and .inpaint (RePaint, :672-831).
// Differences from the conditional sampler (kernels_ddpm.hip): pocket nodes are noised and denoised
// too (x and h), every centre of mass is taken over ALL nodes of a sample, a combined draw has three
// parts (x of all nodes, h_phar, h_pocket; sample_combined_position_feature_noise :555-574), and
// inpainting re-noises the known part from the input at every step and merges by the fixed masks.
// One workgroup of JT threads per sample; the sums the reference forms with scatter_add/scatter_mean run in index
// order (phar rows, then pocket rows) on three threads, everything element-wise is spread over the workgroup, and a
// Philox call fills the four columns it generates.  Compiled with -ffp-contract=off (no FMA re-association).
#include "cmdgen_dev.h"

namespace {

constexpr int JT = 256;         // threads per sample

struct SampleView {             // sample b of the flat batch
    int nl, np, pb, qb, ldp, ldq;
};

__device__ __forceinline__ SampleView view_of(const Layout& lay, const Dims& d, int b) {
    return SampleView{lay.num_phar[b], lay.num_pocket[b], lay.phar_base[b], lay.pocket_base[b], 3 + d.P, 3 + d.R};
}

// max that keeps a NaN, and the bit-ordered atomic maximum that keeps one too (see kernels_ddpm.hip: the reference's
// assertions fail on NaN, so it must reach the host)
__device__ __forceinline__ float max_nan(float a, float b) { return (a != a || b != b) ? __uint_as_float(0x7fc00000u) : fmaxf(a, b); }
__device__ __forceinline__ void atomic_max_pos(unsigned int* slot, float v) {
    atomicMax(slot, __float_as_uint(fabsf(v)) & 0x7fffffffu);
}

// raw standard normals of combined draw `draw_idx` for node `i` (phar: 0..nl-1, pocket: 0..np-1), columns 4g .. 4g+3
// (one Philox counter per group of four columns; columns past the row's width are not used)
__device__ __forceinline__ void jdraw4(const JointBuf& c, const Layout& lay, const SampleView& v, int draw_idx, int b,
                                       bool pocket, int i, int g, float (&z)[4]) {
    if (c.noise) {
        const size_t row = (size_t)lay.Nl * v.ldp + (size_t)lay.Np * v.ldq;
        const float* base = c.noise + (size_t)draw_idx * row;
        const float* src = pocket ? base + (size_t)lay.Nl * v.ldp + (size_t)(v.qb + i) * v.ldq : base + (size_t)(v.pb + i) * v.ldp;
        const int ld = pocket ? v.ldq : v.ldp;
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = 4 * g + j < ld ? src[4 * g + j] : 0.f;
        return;
    }
    const int local = pocket ? v.nl + i : i;            // node index inside the sample
    philox_normal4(c.seed, (uint32_t)lay.pocket_gid[b], (uint32_t)(lay.pocket_gid[b] >> 32),
                   (uint32_t)draw_idx | 0x40000000u,     // joint draws never collide with the conditional sampler's
                   (uint32_t)(local * 8 + g), z);
}

// Workgroup scratch: sm = broadcast slots; xs / fx = the sample's x columns ([3][n], phar rows then pocket rows) and fixed flags staged for the
// ordered sums below (null when the sample is too large for the buffer the launcher sized: the sums then read global memory)
struct JScratch { float* sm; float* xs; float* fx; };
#ifndef CMDGEN_J_STAGE_MAX_N
#define CMDGEN_J_STAGE_MAX_N 3800      // (8 + 4 n) floats <= 60 KB of dynamic LDS; larger samples (the layout allows 5705 nodes) sum from global memory.
#endif                                  // (-DCMDGEN_J_STAGE_MAX_N=4 puts every sample on that path: how tests/test_hip_joint.py was run over it once, profiles/r06_m)
constexpr int J_STAGE_MAX_N = CMDGEN_J_STAGE_MAX_N;

__device__ __forceinline__ JScratch scratch_of(float* jsm, const Layout& lay) {
    const bool st = lay.max_n <= J_STAGE_MAX_N;
    return JScratch{jsm, st ? jsm + 8 : nullptr, st ? jsm + 8 + 3 * lay.max_n : nullptr};
}
__device__ __forceinline__ void stage_fixed(const JointBuf& c, const SampleView& v, int tid, const JScratch& S) {
    if (!S.fx || !c.fix_phar) return;
    for (int i = tid; i < v.nl + v.np; i += JT) S.fx[i] = i < v.nl ? c.fix_phar[v.pb + i] : c.fix_pocket[v.qb + i - v.nl];
    __syncthreads();
}

// The sums of the three x columns over the sample's nodes IN INDEX ORDER (phar rows, then pocket rows - the order scatter_add runs in), on
// threads 0..2; `known`: over the fixed nodes only, cnt = how many.  The workgroup first copies the columns to LDS in one pass, so the ordered
// sum is a chain of LDS reads instead of dependent global loads (59 nodes: ~40 us -> ~2 us per sum).  The caller's barrier after its use of
// the result separates this staging from the next.
__device__ __forceinline__ void ordered_sums(const float* zp, const float* zq, const JointBuf& c, const SampleView& v, int tid, const JScratch& S,
                                             bool known, float& s, float& cnt) {
    const int n = v.nl + v.np;
    if (S.xs) {
        for (int idx = tid; idx < 3 * n; idx += JT) {
            const int k = idx / n, i = idx - k * n;
            S.xs[idx] = i < v.nl ? zp[(size_t)(v.pb + i) * v.ldp + k] : zq[(size_t)(v.qb + i - v.nl) * v.ldq + k];
        }
        __syncthreads();
    }
    s = 0.f; cnt = 0.f;
    if (tid >= 3) return;
    if (S.xs) {
        const float* x = S.xs + tid * n;
        if (known) for (int i = 0; i < n; ++i) { const bool f = S.fx[i] != 0.f; s += f ? x[i] : 0.f; cnt += f ? 1.f : 0.f; }   // (s is never -0: adding +0 changes nothing)
        else for (int i = 0; i < n; ++i) s += x[i];
    } else if (known) {
        for (int i = 0; i < v.nl; ++i) if (c.fix_phar[v.pb + i] != 0.f) { s += zp[(size_t)(v.pb + i) * v.ldp + tid]; cnt += 1.f; }
        for (int i = 0; i < v.np; ++i) if (c.fix_pocket[v.qb + i] != 0.f) { s += zq[(size_t)(v.qb + i) * v.ldq + tid]; cnt += 1.f; }
    } else {
        for (int i = 0; i < v.nl; ++i) s += zp[(size_t)(v.pb + i) * v.ldp + tid];
        for (int i = 0; i < v.np; ++i) s += zq[(size_t)(v.qb + i) * v.ldq + tid];
    }
}

// mean over all nodes of the sample of the three x columns; result broadcast to the workgroup through sm[0..2] (the caller's barriers
// separate two uses of the same slots)
__device__ __forceinline__ void mean_all(const float* zp, const float* zq, const JointBuf& c, const SampleView& v, int tid, const JScratch& S,
                                         float& m0, float& m1, float& m2) {
    float s, cnt;
    ordered_sums(zp, zq, c, v, tid, S, false, s, cnt);
    if (tid < 3) S.sm[tid] = s / fmaxf((float)(v.nl + v.np), 1.0f);
    __syncthreads();
    m0 = S.sm[0]; m1 = S.sm[1]; m2 = S.sm[2];
}

__device__ __forceinline__ void sub_all(float* zp, float* zq, const SampleView& v, int tid, float m0, float m1, float m2) {
    for (int i = tid; i < v.nl; i += JT) { float* p = zp + (size_t)(v.pb + i) * v.ldp; p[0] -= m0; p[1] -= m1; p[2] -= m2; }
    for (int i = tid; i < v.np; i += JT) { float* p = zq + (size_t)(v.qb + i) * v.ldq; p[0] -= m0; p[1] -= m1; p[2] -= m2; }
}

// remove_mean_batch over the concatenated (phar, pocket) x columns (en_diffusion.py:914-917)
__device__ __forceinline__ void remove_mean_all(float* zp, float* zq, const JointBuf& c, const SampleView& v, int tid, const JScratch& S) {
    float m0, m1, m2;
    mean_all(zp, zq, c, v, tid, S, m0, m1, m2);
    sub_all(zp, zq, v, tid, m0, m1, m2);
}

// e_phar / e_pocket <- combined draw `draw_idx` with the x part COM-projected (:555-574, :927-937)
__device__ __forceinline__ void fill_noise(const JointBuf& c, const Layout& lay, const SampleView& v, int draw_idx, int b, int tid, const JScratch& S) {
    const int gp = (v.ldp + 3) >> 2, gq = (v.ldq + 3) >> 2;
    for (int idx = tid; idx < v.nl * gp; idx += JT) {
        const int i = idx / gp, g = idx - i * gp;
        float z[4];
        jdraw4(c, lay, v, draw_idx, b, false, i, g, z);
        float* e = c.e_phar + (size_t)(v.pb + i) * v.ldp + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * g + j < v.ldp) e[j] = z[j];
    }
    for (int idx = tid; idx < v.np * gq; idx += JT) {
        const int i = idx / gq, g = idx - i * gq;
        float z[4];
        jdraw4(c, lay, v, draw_idx, b, true, i, g, z);
        float* e = c.e_pocket + (size_t)(v.qb + i) * v.ldq + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * g + j < v.ldq) e[j] = z[j];
    }
    __syncthreads();
    remove_mean_all(c.e_phar, c.e_pocket, c, v, tid, S);
    __syncthreads();
}

// mean of the x columns over the FIXED nodes of the sample (phar rows, then pocket rows); 0 when none is fixed.  slot: which sm triple
__device__ __forceinline__ void mean_known(const float* zp, const float* zq, const JointBuf& c, const SampleView& v, int tid, const JScratch& S, int slot,
                                           float& m0, float& m1, float& m2) {
    float s, cnt;
    ordered_sums(zp, zq, c, v, tid, S, true, s, cnt);
    if (tid < 3) S.sm[slot + tid] = s / fmaxf(cnt, 1.0f);
    __syncthreads();
    m0 = S.sm[slot]; m1 = S.sm[slot + 1]; m2 = S.sm[slot + 2];
}

// the two maxima assert_mean_zero_with_mask compares (en_diffusion.py:919-924), over all nodes; returns the second (|sum| of the x columns,
// valid on thread 0)
__device__ __forceinline__ float record_check(unsigned int* slot2, const float* zp, const float* zq, const JointBuf& c, const SampleView& v, int tid,
                                              const JScratch& S) {
    float mx = 0.f;
    for (int i = tid; i < v.nl; i += JT) {
        const float* p = zp + (size_t)(v.pb + i) * v.ldp;
        mx = max_nan(mx, max_nan(fabsf(p[0]), max_nan(fabsf(p[1]), fabsf(p[2]))));
    }
    for (int i = tid; i < v.np; i += JT) {
        const float* p = zq + (size_t)(v.qb + i) * v.ldq;
        mx = max_nan(mx, max_nan(fabsf(p[0]), max_nan(fabsf(p[1]), fabsf(p[2]))));
    }
    for (int o = 32; o > 0; o >>= 1) mx = max_nan(mx, __shfl_xor(mx, o));
    float s, cnt;
    ordered_sums(zp, zq, c, v, tid, S, false, s, cnt);
    s = fabsf(s);
    s = max_nan(s, max_nan(__shfl(s, 1), __shfl(s, 2)));        // (threads 0..2 sit in the first wave)
    if ((tid & 63) == 0) atomic_max_pos(slot2, mx);             // one maximum per wave: the order of a maximum does not matter
    if (tid == 0) atomic_max_pos(slot2 + 1, s);
    return s;
}

}  // namespace

// z_T ~ combined noise; inpainting also centres the known input on the COM of its fixed nodes (:703-713)
__global__ __launch_bounds__(JT) void k_joint_init(Layout lay, Dims d, JointBuf c,
                                                   const float* __restrict__ phar_x, const float* __restrict__ phar_oh,
                                                   const float* __restrict__ pocket_x, const float* __restrict__ pocket_oh) {
    const int b = blockIdx.x, tid = threadIdx.x;
    extern __shared__ float jsm[];
    const JScratch S = scratch_of(jsm, lay);
    const SampleView v = view_of(lay, d, b);
    stage_fixed(c, v, tid, S);
    if (c.fix_phar) {
        for (int idx = tid; idx < v.nl * v.ldp; idx += JT) {
            const int i = idx / v.ldp, k = idx - i * v.ldp;
            c.x0_phar[(size_t)(v.pb + i) * v.ldp + k] = k < 3 ? phar_x[(size_t)(v.pb + i) * 3 + k] : phar_oh[(size_t)(v.pb + i) * d.P + k - 3];
        }
        for (int idx = tid; idx < v.np * v.ldq; idx += JT) {
            const int i = idx / v.ldq, k = idx - i * v.ldq;
            c.x0_pocket[(size_t)(v.qb + i) * v.ldq + k] = k < 3 ? pocket_x[(size_t)(v.qb + i) * 3 + k] : pocket_oh[(size_t)(v.qb + i) * d.R + k - 3];
        }
        __syncthreads();
        float m0, m1, m2;
        mean_known(c.x0_phar, c.x0_pocket, c, v, tid, S, 0, m0, m1, m2);
        sub_all(c.x0_phar, c.x0_pocket, v, tid, m0, m1, m2);
        __syncthreads();
    }
    fill_noise(c, lay, v, 0, b, tid, S);
    for (int idx = tid; idx < v.nl * v.ldp; idx += JT) c.z_phar[(size_t)v.pb * v.ldp + idx] = c.e_phar[(size_t)v.pb * v.ldp + idx];
    for (int idx = tid; idx < v.np * v.ldq; idx += JT) c.z_pocket[(size_t)v.qb * v.ldq + idx] = c.e_pocket[(size_t)v.qb * v.ldq + idx];
    __syncthreads();
    record_check(c.check, c.z_phar, c.z_pocket, c, v, tid, S);
}

// one denoising step t -> s (sample_p_zs_given_zt :499-553); inpainting: noised known part (:735-740),
// COM alignment and merge (:757-782) and, where the schedule says so, the jump back (:796-811, :475-497).
// eps_* already carry the NaN reset and the velocity COM removal (k_vel_com).
__global__ __launch_bounds__(JT) void k_joint_step(Layout lay, Dims d, JointBuf c,
                                                   const float* __restrict__ eps_phar, const float* __restrict__ eps_pocket) {
    const int b = blockIdx.x, tid = threadIdx.x;
    extern __shared__ float jsm[];
    const JScratch S = scratch_of(jsm, lay);
    const SampleView v = view_of(lay, d, b);
    const int step = c.state->step - 1;      // k_readout has counted the evaluation (ChainState)
    const float4 cf = c.coef[step], cf2 = c.coef2[step];
    const int4 io = c.iop[step];
    const bool inpaint = c.fix_phar != nullptr;
    int draw_idx = io.y;
    stage_fixed(c, v, tid, S);
    record_check(c.check + 2 * (1 + step), c.z_phar, c.z_pocket, c, v, tid, S);        // z_t, the step's input
    __syncthreads();
    if (inpaint) {      // known nodes from the input: q(z_s | x), its own combined draw, taken first
        fill_noise(c, lay, v, draw_idx++, b, tid, S);
        for (int idx = tid; idx < v.nl * v.ldp; idx += JT) {
            const size_t o = (size_t)v.pb * v.ldp + idx;
            c.zk_phar[o] = cf2.x * c.x0_phar[o] + cf2.y * c.e_phar[o];
        }
        for (int idx = tid; idx < v.np * v.ldq; idx += JT) {
            const size_t o = (size_t)v.qb * v.ldq + idx;
            c.zk_pocket[o] = cf2.x * c.x0_pocket[o] + cf2.y * c.e_pocket[o];
        }
        __syncthreads();
    }
    fill_noise(c, lay, v, draw_idx++, b, tid, S);
    for (int idx = tid; idx < v.nl * v.ldp; idx += JT) {
        const size_t o = (size_t)v.pb * v.ldp + idx;
        const float mu = c.z_phar[o] / cf.x - cf.y * eps_phar[o];
        c.z_phar[o] = mu + cf.z * c.e_phar[o];
    }
    for (int idx = tid; idx < v.np * v.ldq; idx += JT) {
        const size_t o = (size_t)v.qb * v.ldq + idx;
        const float mu = c.z_pocket[o] / cf.x - cf.y * eps_pocket[o];
        c.z_pocket[o] = mu + cf.z * c.e_pocket[o];
    }
    __syncthreads();
    remove_mean_all(c.z_phar, c.z_pocket, c, v, tid, S);
    __syncthreads();
    if (inpaint) {
        float n0, n1, n2, u0, u1, u2;
        mean_known(c.zk_phar, c.zk_pocket, c, v, tid, S, 0, n0, n1, n2);       // com_noised
        mean_known(c.z_phar, c.z_pocket, c, v, tid, S, 4, u0, u1, u2);         // com_denoised
        const float s0 = u0 - n0, s1 = u1 - n1, s2 = u2 - n2;
        for (int idx = tid; idx < v.nl * v.ldp; idx += JT) {
            const int i = idx / v.ldp, k = idx - i * v.ldp;
            const size_t o = (size_t)v.pb * v.ldp + idx;
            float zk = c.zk_phar[o];
            if (k < 3) zk = zk + (k == 0 ? s0 : k == 1 ? s1 : s2);
            const float f = c.fix_phar[v.pb + i];
            c.z_phar[o] = zk * f + c.z_phar[o] * (1.0f - f);
        }
        for (int idx = tid; idx < v.np * v.ldq; idx += JT) {
            const int i = idx / v.ldq, k = idx - i * v.ldq;
            const size_t o = (size_t)v.qb * v.ldq + idx;
            float zk = c.zk_pocket[o];
            if (k < 3) zk = zk + (k == 0 ? s0 : k == 1 ? s1 : s2);
            const float f = c.fix_pocket[v.qb + i];
            c.z_pocket[o] = zk * f + c.z_pocket[o] * (1.0f - f);
        }
        __syncthreads();
    }
    if (c.z_steps) {
        float* dst = c.z_steps + (size_t)step * ((size_t)lay.Nl * v.ldp + (size_t)lay.Np * v.ldq);
        for (int idx = tid; idx < v.nl * v.ldp; idx += JT) dst[(size_t)v.pb * v.ldp + idx] = c.z_phar[(size_t)v.pb * v.ldp + idx];
        for (int idx = tid; idx < v.np * v.ldq; idx += JT)
            dst[(size_t)lay.Nl * v.ldp + (size_t)v.qb * v.ldq + idx] = c.z_pocket[(size_t)v.qb * v.ldq + idx];
    }
    if (io.x & 1) {     // jump back: z_t ~ q(z_t | z_s), then the COM projection
        fill_noise(c, lay, v, draw_idx++, b, tid, S);
        for (int idx = tid; idx < v.nl * v.ldp; idx += JT) {
            const size_t o = (size_t)v.pb * v.ldp + idx;
            c.z_phar[o] = cf2.z * c.z_phar[o] + cf2.w * c.e_phar[o];
        }
        for (int idx = tid; idx < v.np * v.ldq; idx += JT) {
            const size_t o = (size_t)v.qb * v.ldq + idx;
            c.z_pocket[o] = cf2.z * c.z_pocket[o] + cf2.w * c.e_pocket[o];
        }
        __syncthreads();
        remove_mean_all(c.z_phar, c.z_pocket, c, v, tid, S);
    }
}

// p(x, h | z_0) for both node types (sample_p_xh_given_z0 :259-284), un-normalise, one-hot; records the
// CoG drift of the un-normalised coordinates (:634-641)
__global__ __launch_bounds__(JT) void k_joint_final(Layout lay, Dims d, JointBuf c,
                                                    const float* __restrict__ eps_phar, const float* __restrict__ eps_pocket,
                                                    float* __restrict__ xh_phar_out, float* __restrict__ xh_pocket_out,
                                                    unsigned int* cog_slot) {
    const int b = blockIdx.x, tid = threadIdx.x;
    extern __shared__ float jsm[];
    const JScratch S = scratch_of(jsm, lay);
    const SampleView v = view_of(lay, d, b);
    const int step = c.state->step - 1;      // k_readout has counted the evaluation (ChainState)
    const float4 cf = c.coef[step];                     // (sigma_0, alpha_0, sigma_x, 0)
    const int draw_idx = c.iop[step].y;
    for (int i = tid; i < v.nl; i += JT) {             // types from z_0 itself: argmax of the un-normalised h
        const float* z = c.z_phar + (size_t)(v.pb + i) * v.ldp;
        int best = 0; float bv = z[3] * d.norm_h + d.bias_h;
        for (int k = 1; k < d.P; ++k) { const float x = z[3 + k] * d.norm_h + d.bias_h; if (x > bv) { bv = x; best = k; } }
        float* o = xh_phar_out + (size_t)(v.pb + i) * v.ldp;
        for (int k = 0; k < d.P; ++k) o[3 + k] = (k == best) ? 1.0f : 0.0f;
    }
    for (int i = tid; i < v.np; i += JT) {
        const float* z = c.z_pocket + (size_t)(v.qb + i) * v.ldq;
        int best = 0; float bv = z[3] * d.norm_h + d.bias_h;
        for (int k = 1; k < d.R; ++k) { const float x = z[3 + k] * d.norm_h + d.bias_h; if (x > bv) { bv = x; best = k; } }
        float* o = xh_pocket_out + (size_t)(v.qb + i) * v.ldq;
        for (int k = 0; k < d.R; ++k) o[3 + k] = (k == best) ? 1.0f : 0.0f;
    }
    fill_noise(c, lay, v, draw_idx, b, tid, S);
    for (int idx = tid; idx < v.nl * 3; idx += JT) {
        const int i = idx / 3, k = idx - 3 * i;
        const size_t o = (size_t)(v.pb + i) * v.ldp + k;
        const float mu = (1.0f / cf.y) * (c.z_phar[o] - cf.x * eps_phar[o]);
        xh_phar_out[o] = (mu + cf.z * c.e_phar[o]) * d.norm_x;
    }
    for (int idx = tid; idx < v.np * 3; idx += JT) {
        const int i = idx / 3, k = idx - 3 * i;
        const size_t o = (size_t)(v.qb + i) * v.ldq + k;
        const float mu = (1.0f / cf.y) * (c.z_pocket[o] - cf.x * eps_pocket[o]);
        xh_pocket_out[o] = (mu + cf.z * c.e_pocket[o]) * d.norm_x;
    }
    __syncthreads();
    const float s = record_check(c.check + 2 * (1 + step), xh_phar_out, xh_pocket_out, c, v, tid, S);     // (its second maximum is the CoG drift)
    if (tid == 0) atomic_max_pos(cog_slot, s);
}

// batch-wide CoG drift above 5e-2: every sample is re-centred over all its nodes (:636-641)
__global__ __launch_bounds__(JT) void k_joint_drift_fix(Layout lay, Dims d, float* __restrict__ xh_phar_out,
                                                        float* __restrict__ xh_pocket_out, const unsigned int* cog_slot) {
    if (!(__uint_as_float(*cog_slot) > 5e-2f)) return;
    const SampleView v = view_of(lay, d, blockIdx.x);
    extern __shared__ float jsm[];
    const JointBuf none{};
    remove_mean_all(xh_phar_out, xh_pocket_out, none, v, threadIdx.x, scratch_of(jsm, lay));
}

static size_t joint_lds(const Layout& lay) { return sizeof(float) * (size_t)(lay.max_n <= J_STAGE_MAX_N ? 8 + 4 * lay.max_n : 8); }

void cmdgen_launch_joint_init(const Layout& lay, const Dims& d, const JointBuf& c, const float* phx, const float* phoh,
                              const float* px, const float* poh, hipStream_t s) {
    hipLaunchKernelGGL(k_joint_init, dim3(lay.B), dim3(JT), joint_lds(lay), s, lay, d, c, phx, phoh, px, poh);
}
void cmdgen_launch_joint_step(const Layout& lay, const Dims& d, const JointBuf& c, const float* ep, const float* eq, hipStream_t s) {
    hipLaunchKernelGGL(k_joint_step, dim3(lay.B), dim3(JT), joint_lds(lay), s, lay, d, c, ep, eq);
}
void cmdgen_launch_joint_final(const Layout& lay, const Dims& d, const JointBuf& c, const float* ep, const float* eq,
                               float* xo, float* po, unsigned int* cog, hipStream_t s) {
    hipLaunchKernelGGL(k_joint_final, dim3(lay.B), dim3(JT), joint_lds(lay), s, lay, d, c, ep, eq, xo, po, cog);
    hipLaunchKernelGGL(k_joint_drift_fix, dim3(lay.B), dim3(JT), joint_lds(lay), s, lay, d, xo, po, (const unsigned int*)cog);
}
