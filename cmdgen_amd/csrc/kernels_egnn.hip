// kernels_egnn.hip - one EGNNDynamics.forward (dynamics.py:75-139) as gfx950 kernels: the launch sequence, k_readout and the small per-node
// kernels.  The tile kernels are one translation unit per family (kernels_egnn_graph.hip: radius graph + k_embed; kernels_egnn_msg.hip;
// kernels_egnn_node.hip; kernels_egnn_coord.hip; shared device code: cmdgen_egnn_common.h).
//
// Launch sequence of one evaluation (host side: cmdgen_api.hip, launch_evaluation):
//   k_edge_count   radius graph, pass 1: degrees per receiver      (dynamics.py:141-147)
//   k_edge_write   pass 2: compact (row, col, d0) lists sorted like torch.where
//   k_embed        encoders + time column + embedding + P/Q of block 0
//   per block l:   k_edge_msg   GCL.edge_model + segment sum          (egnn_new.py:31-52)
//                  k_node       GCL.node_model, then P/Q for the coord MLP and block l+1
//                  k_edge_coord EquivariantUpdate.coord_model          (egnn_new.py:87-104)
//   k_readout      embedding_out, decoders, velocity, NaN flag       (dynamics.py:110-139)
//
// Tiling: a workgroup owns MT rows (edges or nodes; MT = 64, 32 or 16, chosen per launch so
// that even a 64-pocket batch spreads over all 256 CUs) and all H output columns; wave w owns
// columns [64w, 64w+64), so the block has H/64 waves.  The A operand (rows x H) lives in LDS
// with a 4-float row pad (conflict-free ds_read_b128); the B operand (weights) streams from
// L2 in MFMA fragment order (cmdgen_dev.h).
#include "cmdgen_egnn_common.h"

// ------------------------------------------------------------------------------------
// k_readout: embedding_out (drop the time column), decoders, velocity, NaN flag
// (egnn_new.py:205, dynamics.py:110-131).  8 nodes per workgroup, 32 threads per node; the
// node's h row is staged in LDS and the transposed weight is read coalesced.
// eps rows: [vel(3) | decoded features].  The batch-global NaN reset is applied by the
// consumer (k_nan_fix or the DDPM kernels) once the flag is complete.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_readout(Layout lay, Work w, Dims d, SmallW sw,
                                                 float* __restrict__ eps_phar, float* __restrict__ eps_pocket,
                                                 ChainState* chain, TrainSave sv) {
    extern __shared__ float s_hrow[];            // [8][H] node rows, then [H][dyn] embedding_out^T
    __shared__ float s_j[8][CMDGEN_MAX_SMALL + 1];
    __shared__ float s_h1[8][CMDGEN_MAX_SMALL];
    const int tid = threadIdx.x, g = tid >> 5, l32 = tid & 31;
    if (chain && blockIdx.x == 0 && tid == 0) chain->step += 1;      // this evaluation is done (see ChainState)
    const int nnodes = eps_pocket ? lay.N : lay.Nl;
    const int n = blockIdx.x * 8 + g;
    const bool live = n < nnodes;
    const bool ph = n < lay.Nl;
    const int H = d.H;
    float* s_wT = s_hrow + 8 * H;
    for (int i = tid; i < H * d.dyn; i += 256) s_wT[i] = sw.embo_wT[i];        // coalesced, all loads in flight
    if (live) for (int k = l32; k < H; k += 32) s_hrow[g * H + k] = w.h[(size_t)n * H + k];
    __syncthreads();
    if (live) {
        for (int j = l32; j < d.J; j += 32) {
            float s0 = sw.embo_b[j], s1 = 0.f, s2 = 0.f, s3 = 0.f;
            const float* hr = s_hrow + g * H;
            const float* wt = s_wT + j;
            for (int k = 0; k < H; k += 4) {
                s0 = fmaf(hr[k], wt[k * d.dyn], s0);
                s1 = fmaf(hr[k + 1], wt[(k + 1) * d.dyn], s1);
                s2 = fmaf(hr[k + 2], wt[(k + 2) * d.dyn], s2);
                s3 = fmaf(hr[k + 3], wt[(k + 3) * d.dyn], s3);
            }
            s_j[g][j] = (s0 + s1) + (s2 + s3);
            if (sv.hfin) sv.hfin[(size_t)n * d.dyn + j] = s_j[g][j];
        }
    }
    __syncthreads();
    if (live) {
        const int F = ph ? d.P : d.R;
        const float* W0 = ph ? sw.pd0_w : sw.rd0_w; const float* B0 = ph ? sw.pd0_b : sw.rd0_b;
        for (int o = l32; o < 2 * F; o += 32) {
            float s = B0[o];
            for (int k = 0; k < d.J; ++k) s = fmaf(s_j[g][k], W0[(size_t)o * d.J + k], s);
            const float a = silu_f(s);
            s_h1[g][o] = a;
            if (sv.dec1) {
                if (ph) { sv.dec1[(size_t)n * 2 * F + o] = s; sv.deca[(size_t)n * 2 * F + o] = a; }
                else if (sv.qdec1) { sv.qdec1[(size_t)(n - lay.Nl) * 2 * F + o] = s; sv.qdeca[(size_t)(n - lay.Nl) * 2 * F + o] = a; }
            }
        }
    }
    __syncthreads();
    if (live) {
        const int F = ph ? d.P : d.R;
        const float* W2 = ph ? sw.pd2_w : sw.rd2_w; const float* B2 = ph ? sw.pd2_b : sw.rd2_b;
        float* out = ph ? eps_phar + (size_t)n * (3 + d.P) : eps_pocket + (size_t)(n - lay.Nl) * (3 + d.R);
        for (int o = l32; o < F; o += 32) {
            float s = B2[o];
            for (int k = 0; k < 2 * F; ++k) s = fmaf(s_h1[g][k], W2[(size_t)o * 2 * F + k], s);
            out[3 + o] = s;
            if (sv.dec_out) {
                if (ph) sv.dec_out[(size_t)n * F + o] = s;
                else if (sv.qdec_out) sv.qdec_out[(size_t)(n - lay.Nl) * F + o] = s;
            }
        }
        if (l32 == 0) {
            float vx = 0.f, vy = 0.f, vz = 0.f;
            if (n < lay.Nm) {
                // x_final = X[L-1] + ACC[L-1]/nf ; vel = x_final - x_input
                const float4 p = (d.L == 1) ? w.X0[n] : w.XL[(size_t)(d.L - 1) * lay.Nm + n];
                const float4 a = w.ACC[(size_t)(d.L - 1) * lay.Nm + n];
                const float4 x0 = w.X0[n];
                const float dv = agg_div(w, d, n);
                vx = (p.x + a.x / dv) - x0.x;
                vy = (p.y + a.y / dv) - x0.y;
                vz = (p.z + a.z / dv) - x0.z;
                if (isnan(vx) || isnan(vy) || isnan(vz)) atomicOr(w.nan_flag, 1);
            }
            out[0] = vx; out[1] = vy; out[2] = vz;
        }
    }
}

// applies the reference's batch-global NaN reset to an evaluation's output (dynamics.py:129-131)
__global__ void k_nan_fix(Layout lay, Work w, Dims d, float* __restrict__ eps_phar) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0 && *w.nan_flag) atomicAdd(&w.counters[4], 1ull);
    if (n < lay.Nl && *w.nan_flag) {
        float* o = eps_phar + (size_t)n * (3 + d.P);
        o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
    }
}

// Joint mode tail of EGNNDynamics.forward (dynamics.py:129-136): the batch-global NaN reset, then
// remove_mean_batch(vel, mask) over ALL nodes of each sample.  One wave per sample; the sum runs in
// index order (phar rows, then pocket rows) like the reference's scatter_add.
__global__ __launch_bounds__(64) void k_vel_com(Layout lay, Work w, Dims d, float* __restrict__ eps_phar,
                                                float* __restrict__ eps_pocket) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nl = lay.num_phar[b], np = lay.num_pocket[b];
    const int ldp = 3 + d.P, ldq = 3 + d.R;
    float* vp = eps_phar + (size_t)lay.phar_base[b] * ldp;
    float* vq = eps_pocket + (size_t)lay.pocket_base[b] * ldq;
    const bool nan_reset = *w.nan_flag != 0;
    if (b == 0 && lane == 0 && nan_reset) atomicAdd(&w.counters[4], 1ull);
    if (nan_reset) {
        for (int i = lane; i < nl; i += 64) { vp[i * ldp] = 0.f; vp[i * ldp + 1] = 0.f; vp[i * ldp + 2] = 0.f; }
        for (int i = lane; i < np; i += 64) { vq[i * ldq] = 0.f; vq[i * ldq + 1] = 0.f; vq[i * ldq + 2] = 0.f; }
    }
    __syncthreads();
    float mean = 0.f;
    if (lane < 3) {
        float s = 0.f;
        for (int i = 0; i < nl; ++i) s += vp[i * ldp + lane];
        for (int i = 0; i < np; ++i) s += vq[i * ldq + lane];
        mean = s / fmaxf((float)(nl + np), 1.0f);
    }
    const float m0 = __shfl(mean, 0), m1 = __shfl(mean, 1), m2 = __shfl(mean, 2);
    for (int i = lane; i < nl; i += 64) { vp[i * ldp] -= m0; vp[i * ldp + 1] -= m1; vp[i * ldp + 2] -= m2; }
    for (int i = lane; i < np; i += 64) { vq[i * ldq] -= m0; vq[i * ldq + 1] -= m1; vq[i * ldq + 2] -= m2; }
}

// ------------------------------------------------------------------------------------
// host-callable launchers (C++ linkage)
// ------------------------------------------------------------------------------------
static void launch_eval(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket,
                          const float* t_arr, const float4* coef, ChainState* chain,
                          float* eps_phar, float* eps_pocket, hipStream_t s,
                          hipEvent_t* ev /* null or 2*(3+3L) events */) {
    const int B = a.lay.B, N = a.lay.N;
    const size_t shm = (size_t)a.lay.max_n * (sizeof(float4) + 3 * sizeof(int));
    int e = 0;
#define REC() do { if (ev) hipEventRecord(ev[e++], s); } while (0)
    // kernel profiling: the launch itself carries a start and a stop event (hipExtLaunchKernelGGL: the timestamps of the dispatch
    // packet, i.e. what rocprofv3 reports), not events recorded around it on the stream (those include the launch gap)
#define PROF_BEGIN(k) do { if (a.prof_events && !a.save) { hipEventCreate(&a.pe_start); hipEventCreate(&a.pe_stop); \
                           a.prof_events[k].push_back(a.pe_start); a.prof_events[k].push_back(a.pe_stop); } } while (0)
#define PROF_END() do { a.pe_start = nullptr; a.pe_stop = nullptr; } while (0)
    // k_embed's tile: the small one only where the pocket cache serves the pocket rows (inside a conditional chain)
    const int emt = (chain && !t_arr && a.pcache.c) ? a.embed_mt : a.node_mt;
    REC();
    // per-sample graph kernels: one wave scans one receiver at a time, so big samples (full-atom pockets: 381 nodes) get 16 waves
    const int gthr = a.lay.max_n > 128 ? 1024 : 256;
    if (!a.skip_count) cmdgen_launch_edge_count(a, xh_phar, xh_pocket, s);
    if (a.skip_count == 2) {          // training forward: the graph was built (and its size read back) before the activation store was sized
        REC(); REC();
        cmdgen_launch_embed_tiles(a, emt, xh_phar, xh_pocket, t_arr, coef, chain, s);
    } else if (a.d.H == 256 && !ev && !a.save && gthr == 256 && shm <= 64 * 1024 && (size_t)emt * 1812 + 1024 + (shm > 12288 ? shm : 12288) <= 160 * 1024 &&
               a.write_embed) {      // (static LDS of the embedding body is 1812 B per tile row; one launch must hold both bodies' LDS)
        cmdgen_launch_write_embed_tiles(a, emt, xh_phar, xh_pocket, t_arr, coef, chain, s);       // both in one launch
    } else {
        cmdgen_launch_edge_write(a, s);
        REC(); REC();
        cmdgen_launch_embed_tiles(a, emt, xh_phar, xh_pocket, t_arr, coef, chain, s);
    }
    REC();
    // dead work (conditional sampler, pocket output not asked for): see edge_msg_body / cmdgen_node_planes.h
    const bool live_last = a.dead_skip && !eps_pocket && !a.save && !a.d.joint && a.stop_block < 0 && a.w.need_qc != nullptr;
    for (int l = 0; l < a.d.L; ++l) {
        a.live_thr = live_last ? (a.dead_skip >= 2 ? a.d.L - l : (l == a.d.L - 1 ? 1 : 0)) : 0;     // dead_skip 1: the last block only; 2: every block (a kernel argument of its own: any n_layers)
        const int stop = a.stop_block == l ? a.stop_stage : 0;        // parity aid: leave intermediates in the workspace
        // the block's GCLs (inv_sublayers, egnn_new.py:152-154): message + node kernel per unit; only the last one projects P_c | Q_c.
        // (The per-stage events and the prefix stops belong to the block's last unit; cmdgen_profile_evaluation asks for S = 1.)
        for (int sub = 0; sub < a.d.S; ++sub) {
            const bool last = sub == a.d.S - 1;
            a.unit = l * a.d.S + sub; a.skip_pc = last ? 0 : 1;
            if (last) REC();
            PROF_BEGIN(0);
            if (!cmdgen_launch_msg128(a, l, s)) cmdgen_launch_msg_tiles(a, l, s);
            PROF_END();
            if (last) { REC(); REC(); }
            if (last && stop == 1) { a.unit = -1; a.skip_pc = 0; return; }
            PROF_BEGIN(1);
            if (!(a.node64 && cmdgen_launch_node64(a, l, s)) && !cmdgen_launch_node16w(a, l, s)) cmdgen_launch_node_tiles(a, l, s);
            PROF_END();
            if (last) { REC(); REC(); }
            if (last && stop == 2) { a.unit = -1; a.skip_pc = 0; return; }
        }
        PROF_BEGIN(2);
        if (!cmdgen_launch_coord128(a, l, s)) cmdgen_launch_coord_tiles(a, l, s);
        PROF_END();
        REC();
        a.unit = -1; a.skip_pc = 0;
        if (stop == 3) return;
    }
    REC();
    const int nn = eps_pocket ? N : a.lay.Nl;
    hipLaunchKernelGGL(k_readout, dim3((nn + 7) / 8), dim3(256), (8 + a.d.dyn) * a.d.H * sizeof(float), s, a.lay, a.w, a.d, a.sw,
                       eps_phar, eps_pocket, chain, a.save ? *a.save : TrainSave{});
    if (a.d.joint) hipLaunchKernelGGL(k_vel_com, dim3(B), dim3(64), 0, s, a.lay, a.w, a.d, eps_phar, eps_pocket);
    REC();
#undef REC
#undef PROF_BEGIN
#undef PROF_END
}

void cmdgen_launch_eval(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket,
                        const float* t_arr, const float4* coef, ChainState* chain,
                        float* eps_phar, float* eps_pocket, hipStream_t s, hipEvent_t* ev) {
    launch_eval(a, xh_phar, xh_pocket, t_arr, coef, chain, eps_phar, eps_pocket, s, ev);
}

// positions entering every block (and after the last) for ALL nodes, as the backward pass indexes them: X[l][n], l = 0..L
__global__ void k_save_positions(Layout lay, Work w, Dims d, float4* __restrict__ X) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= lay.N) return;
    for (int l = 0; l <= d.L; ++l) {
        float4 p;
        if (n >= lay.Nm) p = w.XP[n - lay.Nl];
        else if (l == 0) p = w.X0[n];
        else if (l < d.L) p = w.XL[(size_t)l * lay.Nm + n];
        else {
            const float4 q = (d.L == 1) ? w.X0[n] : w.XL[(size_t)(d.L - 1) * lay.Nm + n];
            const float4 a = w.ACC[(size_t)(d.L - 1) * lay.Nm + n];
            const float dv = agg_div(w, d, n);
            p = make_float4(q.x + a.x / dv, q.y + a.y / dv, q.z + a.z / dv, 0.f);
        }
        X[(size_t)l * lay.N + n] = p;
    }
}
void cmdgen_launch_save_positions(const EvalLaunch& a, float4* X, hipStream_t s) {
    hipLaunchKernelGGL(k_save_positions, dim3((a.lay.N + 255) / 256), dim3(256), 0, s, a.lay, a.w, a.d, X);
}

// k_readout stages 8 node rows + embedding_out^T in dynamic LDS: above the 64 KiB default (hidden_nf 512) the kernel needs the opt-in
void cmdgen_readout_allow_lds(size_t bytes) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_readout), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

void cmdgen_launch_nan_fix(const EvalLaunch& a, float* eps_phar, hipStream_t s) {
    hipLaunchKernelGGL(k_nan_fix, dim3((a.lay.Nl + 255) / 256), dim3(256), 0, s, a.lay, a.w, a.d, eps_phar);
}

