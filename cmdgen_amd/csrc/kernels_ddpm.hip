// kernels_ddpm.hip - the ancestral-sampling arithmetic around the network evaluation:
// ConditionalDDPM.sample_given_pocket (conditional_model.py:388-465).  One wave per sample;
// sums that the reference forms with scatter_add/scatter_mean in index order are formed in
// the same order here (a lane walks the sample's nodes) so results agree to the last bit
// given the same inputs.
#include "cmdgen_dev.h"

__device__ __forceinline__ float draw(const ChainBuf& c, const Layout& lay, int draw_idx, int b,
                                      int local, int node, int comp, int ld) {
    if (c.noise) return c.noise[((size_t)draw_idx * lay.Nl + node) * ld + comp];
    float z[4];
    philox_normal4(c.seed, (uint32_t)lay.pocket_gid[b], (uint32_t)(lay.pocket_gid[b] >> 32),
                   (uint32_t)draw_idx, (uint32_t)(local * 4 + (comp >> 2)), z);
    return z[comp & 3];
}

// subtract the phar centre of mass from phar and pocket coordinates of sample b
// (remove_mean_batch, conditional_model.py:467-475); returns nothing, works in place.
__device__ __forceinline__ void remove_com(float* zx, int ld, int pb, int nl, float* px, int ldq,
                                           int qb, int np, int lane) {
    float mean = 0.f;
    if (lane < 3) {
        float s = 0.f;
        for (int i = 0; i < nl; ++i) s += zx[(size_t)(pb + i) * ld + lane];   // index order, as index_add_
        mean = s / fmaxf((float)nl, 1.0f);
    }
    const float m0 = __shfl(mean, 0), m1 = __shfl(mean, 1), m2 = __shfl(mean, 2);
    for (int i = lane; i < nl; i += 64) {
        float* p = zx + (size_t)(pb + i) * ld;
        p[0] -= m0; p[1] -= m1; p[2] -= m2;
    }
    for (int i = lane; i < np; i += 64) {
        float* p = px + (size_t)(qb + i) * ldq;
        p[0] -= m0; p[1] -= m1; p[2] -= m2;
    }
}

__device__ __forceinline__ void atomic_max_pos(unsigned int* slot, float v) {
    atomicMax(slot, __float_as_uint(fabsf(v)) & 0x7fffffffu);     // non-negative floats order like their bits (a NaN's sign bit is cleared: any NaN ranks above +Inf)
}

// max that keeps a NaN (fmaxf drops it): torch's x.abs().max() returns NaN as soon as one element is NaN, and the
// reference's assertion then fails (NaN < 1e-2 is False) - so a NaN must survive into the recorded maxima
__device__ __forceinline__ float max_nan(float a, float b) { return (a != a || b != b) ? __uint_as_float(0x7fc00000u) : fmaxf(a, b); }

// records the two maxima assert_mean_zero_with_mask compares (en_diffusion.py:919-924).  Non-negative floats order like
// their bits and the quiet NaN 0x7fc00000 lies above +Inf, so atomicMax on the bits keeps a NaN once one sample has it.
__device__ __forceinline__ void record_com_check(unsigned int* slot2, const float* zx, int ld, int pb,
                                                 int nl, float scale, int lane) {
    float mx = 0.f;
    for (int i = lane; i < nl; i += 64) {
        const float* p = zx + (size_t)(pb + i) * ld;
        mx = max_nan(mx, max_nan(fabsf(p[0] * scale), max_nan(fabsf(p[1] * scale), fabsf(p[2] * scale))));
    }
    for (int o = 32; o > 0; o >>= 1) mx = max_nan(mx, __shfl_xor(mx, o));
    float s = 0.f;
    if (lane < 3) for (int i = 0; i < nl; ++i) s += zx[(size_t)(pb + i) * ld + lane] * scale;
    s = fabsf(s);
    s = max_nan(s, max_nan(__shfl(s, 1), __shfl(s, 2)));
    if (lane == 0) { atomic_max_pos(slot2, mx); atomic_max_pos(slot2 + 1, s); }
}

// z_T = [pocket COM, 0] + noise, then COM projection (conditional_model.py:402-420)
__global__ __launch_bounds__(64) void k_chain_init(Layout lay, Dims d, ChainBuf c,
                                                   const float* __restrict__ pocket_x,
                                                   const float* __restrict__ pocket_onehot) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nl = lay.num_phar[b], np = lay.num_pocket[b];
    const int pb = lay.phar_base[b], qb = lay.pocket_base[b];
    const int ld = 3 + d.P, ldq = 3 + d.R;
    // normalize (en_diffusion.py:874-889)
    for (int i = lane; i < np; i += 64) {
        float* o = c.xh_pocket + (size_t)(qb + i) * ldq;
        for (int k = 0; k < 3; ++k) o[k] = pocket_x[(size_t)(qb + i) * 3 + k] / d.norm_x;
        for (int k = 0; k < d.R; ++k) o[3 + k] = (pocket_onehot[(size_t)(qb + i) * d.R + k] - d.bias_h) / d.norm_h;
    }
    __syncthreads();
    if (d.no_com) {      // SimpleConditionalDDPM.sample_given_pocket :512-521: subtract the pocket COM once (un-normalised x; norm_x divides both)
        float pm = 0.f;
        if (lane < 3) {
            float s = 0.f;
            for (int i = 0; i < np; ++i) s += pocket_x[(size_t)(qb + i) * 3 + lane];
            pm = s / fmaxf((float)np, 1.0f);
        }
        const float p0 = __shfl(pm, 0), p1 = __shfl(pm, 1), p2 = __shfl(pm, 2);
        for (int i = lane; i < np; i += 64) {
            float* o = c.xh_pocket + (size_t)(qb + i) * ldq;
            o[0] = (pocket_x[(size_t)(qb + i) * 3 + 0] - p0) / d.norm_x;
            o[1] = (pocket_x[(size_t)(qb + i) * 3 + 1] - p1) / d.norm_x;
            o[2] = (pocket_x[(size_t)(qb + i) * 3 + 2] - p2) / d.norm_x;
        }
        __syncthreads();
    }
    float mu = 0.f;
    if (lane < 3) {
        float s = 0.f;
        for (int i = 0; i < np; ++i) s += c.xh_pocket[(size_t)(qb + i) * ldq + lane];
        mu = s / fmaxf((float)np, 1.0f);
    }
    const float m0 = __shfl(mu, 0), m1 = __shfl(mu, 1), m2 = __shfl(mu, 2);
    for (int idx = lane; idx < nl * ld; idx += 64) {
        const int i = idx / ld, k = idx % ld;
        const float m = k == 0 ? m0 : k == 1 ? m1 : k == 2 ? m2 : 0.f;
        c.z_phar[(size_t)(pb + i) * ld + k] = m + 1.0f * draw(c, lay, 0, b, i, pb + i, k, ld);
    }
    __syncthreads();
    if (d.no_com) return;                       // no projection, and the mean-zero assertion is a no-op (:503-505)
    remove_com(c.z_phar, ld, pb, nl, c.xh_pocket, ldq, qb, np, lane);
    __syncthreads();
    record_com_check(c.check, c.z_phar, ld, pb, nl, 1.0f, lane);
}

// one posterior step z_t -> z_s (sample_p_zs_given_zt, conditional_model.py:342-374).
// The sample's z (<= a few hundred floats) and eps are pulled into LDS with all loads in flight, the index-order sums
// (COM, checks) then walk LDS instead of paying one global round trip per node; every sum keeps its order, so the
// results are bit-identical to the straightforward version.
__global__ __launch_bounds__(64) void k_ddpm_step(Layout lay, Dims d, ChainBuf c, Work w,
                                                  const float* __restrict__ eps) {
    extern __shared__ float s_z[];                  // [nl * ld] z of this sample
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nl = lay.num_phar[b], np = lay.num_pocket[b];
    const int pb = lay.phar_base[b], qb = lay.pocket_base[b];
    const int ld = 3 + d.P, ldq = 3 + d.R;
    const int step = c.state->step - 1;             // 0-based index of this posterior step (k_readout has counted the evaluation)
    const float4 cf = c.coef[step];
    const bool nan_reset = *w.nan_flag != 0;
    float* zg = c.z_phar + (size_t)pb * ld;
    const float* eg = eps + (size_t)pb * ld;
    const int cnt = nl * ld;
    for (int idx = lane; idx < cnt; idx += 64) s_z[idx] = zg[idx];
    __syncthreads();
    // the reference checks z_t (the step's input) after the update; same numbers, recorded first
    if (!d.no_com) record_com_check(c.check + 2 * (1 + step), s_z, ld, 0, nl, 1.0f, lane);
    __syncthreads();
    for (int idx = lane; idx < cnt; idx += 64) {
        const int i = idx / ld, k = idx - i * ld;
        float e = eg[idx];
        if (nan_reset && k < 3) e = 0.f;
        const float mu = s_z[idx] / cf.x - cf.y * e;
        s_z[idx] = mu + cf.z * draw(c, lay, 1 + step, b, i, pb + i, k, ld);
    }
    __syncthreads();
    if (!d.no_com) remove_com(s_z, ld, 0, nl, c.xh_pocket, ldq, qb, np, lane);
    __syncthreads();
    for (int idx = lane; idx < cnt; idx += 64) {
        const float v = s_z[idx];
        zg[idx] = v;
        if (c.z_steps) c.z_steps[(size_t)step * lay.Nl * ld + (size_t)pb * ld + idx] = v;
    }
    if (c.pocket_steps)
        for (int idx = lane; idx < np * 3; idx += 64) {
            const int i = idx / 3, k = idx - 3 * i;
            c.pocket_steps[((size_t)step * lay.Np + qb + i) * 3 + k] = c.xh_pocket[(size_t)(qb + i) * ldq + k];
        }
    if (b == 0 && lane == 0 && nan_reset) atomicAdd(&w.counters[4], 1ull);
}

// p(x, h | z_0): sample_p_xh_given_z0 (conditional_model.py:108-131) + unnormalize + one-hot
__global__ __launch_bounds__(64) void k_chain_final(Layout lay, Dims d, ChainBuf c, Work w,
                                                    const float* __restrict__ eps,
                                                    float* __restrict__ xh_phar_out,
                                                    float* __restrict__ xh_pocket_out,
                                                    unsigned int* cog_slot) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nl = lay.num_phar[b], np = lay.num_pocket[b];
    const int pb = lay.phar_base[b], qb = lay.pocket_base[b];
    const int ld = 3 + d.P, ldq = 3 + d.R;
    const int K = c.state->K;
    const float4 cf = c.coef[K];                    // (sigma_0, alpha_0, sigma_x = exp(gamma_0/2), 0)
    const bool nan_reset = *w.nan_flag != 0;
    // types come from z_0 itself (not from the sampled xh): argmax of the un-normalised h
    for (int i = lane; i < nl; i += 64) {
        const float* z = c.z_phar + (size_t)(pb + i) * ld;
        int best = 0; float bv = z[3] * d.norm_h + d.bias_h;
        for (int k = 1; k < d.P; ++k) { const float v = z[3 + k] * d.norm_h + d.bias_h; if (v > bv) { bv = v; best = k; } }
        float* o = xh_phar_out + (size_t)(pb + i) * ld;
        for (int k = 0; k < d.P; ++k) o[3 + k] = (k == best) ? 1.0f : 0.0f;
    }
    __syncthreads();
    // mu_x = 1/alpha_0 * (z_0 - sigma_0 * eps)  (compute_x_pred, en_diffusion.py:153-165); + sigma_x * noise
    for (int idx = lane; idx < nl * ld; idx += 64) {
        const int i = idx / ld, k = idx % ld;
        const size_t o = (size_t)(pb + i) * ld + k;
        float e = eps[o];
        if (nan_reset && k < 3) e = 0.f;
        const float mu = (1.0f / cf.y) * (c.z_phar[o] - cf.x * e);
        c.z_phar[o] = mu + cf.z * draw(c, lay, 1 + K, b, i, pb + i, k, ld);
    }
    __syncthreads();
    if (!d.no_com) remove_com(c.z_phar, ld, pb, nl, c.xh_pocket, ldq, qb, np, lane);
    __syncthreads();
    // unnormalize (en_diffusion.py:891-895)
    for (int i = lane; i < nl; i += 64) {
        const float* z = c.z_phar + (size_t)(pb + i) * ld;
        float* o = xh_phar_out + (size_t)(pb + i) * ld;
        o[0] = z[0] * d.norm_x; o[1] = z[1] * d.norm_x; o[2] = z[2] * d.norm_x;
    }
    for (int i = lane; i < np; i += 64) {
        const float* q = c.xh_pocket + (size_t)(qb + i) * ldq;
        float* o = xh_pocket_out + (size_t)(qb + i) * ldq;
        o[0] = q[0] * d.norm_x; o[1] = q[1] * d.norm_x; o[2] = q[2] * d.norm_x;
        for (int k = 0; k < d.R; ++k) o[3 + k] = q[3 + k] * d.norm_h + d.bias_h;
    }
    __syncthreads();
    if (!d.no_com) record_com_check(c.check + 2 * (1 + K), xh_phar_out, ld, pb, nl, 1.0f, lane);
    // CoG drift of the un-normalised coordinates (conditional_model.py:451-452)
    float s = 0.f;
    if (lane < 3) for (int i = 0; i < nl; ++i) s += xh_phar_out[(size_t)(pb + i) * ld + lane];
    s = fabsf(s);
    s = max_nan(s, max_nan(__shfl(s, 1), __shfl(s, 2)));
    if (lane == 0) atomic_max_pos(cog_slot, s);
    if (b == 0 && lane == 0 && nan_reset) atomicAdd(&w.counters[4], 1ull);
}

// if the batch-wide max drift exceeds 5e-2 every sample is re-centred (conditional_model.py:453-457)
__global__ __launch_bounds__(64) void k_chain_drift_fix(Layout lay, Dims d, float* __restrict__ xh_phar_out,
                                                        float* __restrict__ xh_pocket_out,
                                                        const unsigned int* cog_slot) {
    if (d.no_com || !(__uint_as_float(*cog_slot) > 5e-2f)) return;       // the simple variant's re-centring is the identity (:500-501)
    const int b = blockIdx.x, lane = threadIdx.x;
    remove_com(xh_phar_out, 3 + d.P, lay.phar_base[b], lay.num_phar[b], xh_pocket_out, 3 + d.R,
               lay.pocket_base[b], lay.num_pocket[b], lane);
}

// ------------------------------------------------------------------------------------------------------------
// k_step_count: the posterior step z_t -> z_s (as k_ddpm_step) FUSED with pass 1 of the next evaluation's radius
// graph (k_edge_count, kernels_egnn.hip): both are one-workgroup-per-sample, and the new positions are already in
// LDS when the step is done - one launch and one global round trip of the positions less per denoising step.
// 256 threads per sample.  Arithmetic and summation order of the step are those of k_ddpm_step (bit-identical
// results); the count pass is k_edge_count's (same dist2, same ballots).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_step_count(Layout lay, Dims d, ChainBuf c, Work w,
                                                    const float* __restrict__ eps) {
    extern __shared__ float4 s_pos[];               // [max_n] positions of the sample (phar first), then int sdeg[max_n], then z
    int* sdeg = reinterpret_cast<int*>(s_pos + lay.max_n);
    float* s_z = reinterpret_cast<float*>(sdeg + lay.max_n);      // [nl * ld]
    __shared__ float s_mean[3];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int nl = lay.num_phar[b], np = lay.num_pocket[b], n = nl + np;
    const int pb = lay.phar_base[b], qb = lay.pocket_base[b];
    const int ld = 3 + d.P, ldq = 3 + d.R;
    const int step = c.state->step - 1;
    const float4 cf = c.coef[step];
    const bool nan_reset = *w.nan_flag != 0;
    float* zg = c.z_phar + (size_t)pb * ld;
    const float* eg = eps + (size_t)pb * ld;
    const int cnt = nl * ld;
    for (int idx = tid; idx < cnt; idx += blockDim.x) s_z[idx] = zg[idx];
    // pocket coordinates of this sample: in flight while the step is computed
    for (int i = tid; i < np; i += blockDim.x) {
        const float* q = c.xh_pocket + (size_t)(qb + i) * ldq;
        s_pos[nl + i] = make_float4(q[0], q[1], q[2], 0.f);
    }
    __syncthreads();
    if (wave == 0 && !d.no_com) record_com_check(c.check + 2 * (1 + step), s_z, ld, 0, nl, 1.0f, lane);   // z_t, the step's input
    __syncthreads();
    for (int idx = tid; idx < cnt; idx += blockDim.x) {
        const int i = idx / ld, k = idx - i * ld;
        float e = eg[idx];
        if (nan_reset && k < 3) e = 0.f;
        const float mu = s_z[idx] / cf.x - cf.y * e;
        s_z[idx] = mu + cf.z * draw(c, lay, 1 + step, b, i, pb + i, k, ld);
    }
    __syncthreads();
    if (tid < 3) {                                  // phar centre of mass, index order (remove_mean_batch :467-475)
        float sum = 0.f;
        if (!d.no_com) {
            for (int i = 0; i < nl; ++i) sum += s_z[i * ld + tid];
            sum = sum / fmaxf((float)nl, 1.0f);
        }
        s_mean[tid] = sum;
    }
    __syncthreads();
    const float m0 = s_mean[0], m1 = s_mean[1], m2 = s_mean[2];
    for (int i = tid; i < n; i += blockDim.x) {
        float4 p;
        if (i < nl) {
            float* z = s_z + i * ld;
            if (!d.no_com) { z[0] -= m0; z[1] -= m1; z[2] -= m2; }
            p = make_float4(z[0], z[1], z[2], 0.f);
            w.X0[pb + i] = p;
            for (int l = 0; l < d.L; ++l) w.ACC[(size_t)l * lay.Nm + pb + i] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            p = s_pos[i];
            if (!d.no_com) {
                p.x -= m0; p.y -= m1; p.z -= m2;
                float* q = c.xh_pocket + (size_t)(qb + i - nl) * ldq;
                q[0] = p.x; q[1] = p.y; q[2] = p.z;
            }
            w.XP[qb + i - nl] = p;
            if (c.pocket_steps) {
                float* o = c.pocket_steps + ((size_t)step * lay.Np + qb + i - nl) * 3;
                o[0] = p.x; o[1] = p.y; o[2] = p.z;
            }
        }
        s_pos[i] = p;
    }
    __syncthreads();
    for (int idx = tid; idx < cnt; idx += blockDim.x) {
        const float v = s_z[idx];
        zg[idx] = v;
        if (c.z_steps) c.z_steps[(size_t)step * lay.Nl * ld + (size_t)pb * ld + idx] = v;
    }
    // ---- pass 1 of the radius graph of the NEXT evaluation (as k_edge_count)
    for (int i = wave; i < n; i += nwaves) {
        const float4 pi = s_pos[i];
        int deg = 0, self = 0;
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            bool ok = false;
            if (j < n) {
                const float r2 = dist2(pi, s_pos[j]);
                ok = (d.cutoff2 < 0.f) || (r2 <= d.cutoff2);
            }
            const unsigned long long m = __ballot(ok);
            deg += __popcll(m);
            if (i >= j0 && i < j0 + 64) self = (int)((m >> (i - j0)) & 1ull);
        }
        if (lane == 0) { sdeg[i] = deg | (self << 30); w.degL[pb + qb + i] = deg | (self << 30); }
    }
    __syncthreads();
    if (wave == 0) {
        int e = 0, eph = 0, ens = 0, ensq = 0;
        for (int i = lane; i < n; i += 64) {
            const int dg = sdeg[i] & 0x3fffffff; e += dg;
            if (i < nl) { eph += dg; ens += dg - ((sdeg[i] >> 30) & 1); }
            else ensq += dg - ((sdeg[i] >> 30) & 1);
        }
        for (int o = 32; o > 0; o >>= 1) {
            e += __shfl_xor(e, o); eph += __shfl_xor(eph, o); ens += __shfl_xor(ens, o); ensq += __shfl_xor(ensq, o);
        }
        if (lane == 0) { w.pocketE[b] = e; w.pocketEph[b] = eph; w.pocketEns[b] = ens; w.pocketEnsQ[b] = ensq; }
    }
    if (b == 0 && tid == 0) {
        if (nan_reset) atomicAdd(&w.counters[4], 1ull);
        atomicAdd(&w.counters[0], 1ull);                       // evaluations (the one about to run)
        atomicAdd(&w.counters[3], (unsigned long long)lay.N);  // nodes
    }
}

// (The NaN flag this kernel reads is cleared for the next evaluation by k_edge_write, which runs after every reader
// of the old value and before k_readout can set it again.)

void cmdgen_launch_chain_init(const Layout& lay, const Dims& d, const ChainBuf& c, const float* px,
                              const float* poh, hipStream_t s) {
    hipLaunchKernelGGL(k_chain_init, dim3(lay.B), dim3(64), 0, s, lay, d, c, px, poh);
}
void cmdgen_launch_ddpm_step(const Layout& lay, const Dims& d, const ChainBuf& c, const Work& w,
                             const float* eps, hipStream_t s) {
    // dynamic LDS: the largest sample's z.  (max_n bounds nl; 3 + P floats per node)
    hipLaunchKernelGGL(k_ddpm_step, dim3(lay.B), dim3(64), (size_t)lay.max_n * (3 + d.P) * sizeof(float), s, lay, d, c, w, eps);
}
void cmdgen_launch_step_count(const Layout& lay, const Dims& d, const ChainBuf& c, const Work& w,
                              const float* eps, hipStream_t s) {
    const size_t shm = (size_t)lay.max_n * (sizeof(float4) + sizeof(int)) + (size_t)lay.max_n * (3 + d.P) * sizeof(float);
    hipLaunchKernelGGL(k_step_count, dim3(lay.B), dim3(lay.max_n > 128 ? 1024 : 256), shm, s, lay, d, c, w, eps);     // 16 waves for big samples (one receiver per wave at a time)
}
void cmdgen_launch_chain_final(const Layout& lay, const Dims& d, const ChainBuf& c, const Work& w,
                               const float* eps, float* xo, float* po, unsigned int* cog, hipStream_t s) {
    hipLaunchKernelGGL(k_chain_final, dim3(lay.B), dim3(64), 0, s, lay, d, c, w, eps, xo, po, cog);
    hipLaunchKernelGGL(k_chain_drift_fix, dim3(lay.B), dim3(64), 0, s, lay, d, xo, po, (const unsigned int*)cog);
}

// same generator as draw() above, exposed for statistical tests
__global__ void k_debug_noise(unsigned long long seed, long long pocket_id, int draw_idx, int n_nodes, int width,
                              float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_nodes * width) return;
    const int local = idx / width, comp = idx - local * width;
    float z[4];
    philox_normal4(seed, (uint32_t)pocket_id, (uint32_t)((unsigned long long)pocket_id >> 32), (uint32_t)draw_idx,
                   (uint32_t)(local * 4 + (comp >> 2)), z);
    out[idx] = z[comp & 3];
}
void cmdgen_launch_debug_noise(unsigned long long seed, long long pocket_id, int draw, int n_nodes, int width,
                               float* out, hipStream_t s) {
    const int n = n_nodes * width;
    hipLaunchKernelGGL(k_debug_noise, dim3((n + 255) / 256), dim3(256), 0, s, seed, pocket_id, draw, n_nodes, width, out);
}
