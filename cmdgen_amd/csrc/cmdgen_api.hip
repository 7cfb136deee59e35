// cmdgen_api.hip - the C ABI of libcmdgen_hip.so (include/cmdgen_hip.h): handle, weight
// packing, workspaces, the launch sequence of one evaluation and the denoising loop
// (eager or replayed as a hipGraph).
#include "cmdgen_dev.h"
#include "../../include/cmdgen_hip.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

// launchers implemented next to the kernels
void cmdgen_launch_eval(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket,
                        const float* t_arr, const float4* coef, ChainState* chain, float* eps_phar,
                        float* eps_pocket, hipStream_t s, hipEvent_t* ev);
void cmdgen_launch_nan_fix(const EvalLaunch& a, float* eps_phar, hipStream_t s);
void cmdgen_readout_allow_lds(size_t bytes);      // kernels_egnn.hip: k_readout's dynamic LDS above the 64 KiB default (hidden_nf 512)
void cmdgen_launch_edges(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket, hipStream_t s);
void cmdgen_build_pocket_cache(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket, const float* t01,
                               float* c, float* P0, float* Q0, float* dh, float* dP, float* dQ, hipStream_t s);
void cmdgen_launch_edge_msg_only(const EvalLaunch& a, int layer, hipStream_t s);
void cmdgen_launch_chain_init(const Layout& lay, const Dims& d, const ChainBuf& c, const float* px,
                              const float* poh, hipStream_t s);
void cmdgen_launch_ddpm_step(const Layout& lay, const Dims& d, const ChainBuf& c, const Work& w,
                             const float* eps, hipStream_t s);
void cmdgen_launch_step_count(const Layout& lay, const Dims& d, const ChainBuf& c, const Work& w,
                              const float* eps, hipStream_t s);
void cmdgen_launch_debug_noise(unsigned long long seed, long long pocket_id, int draw, int n_nodes, int width,
                               float* out, hipStream_t s);
void cmdgen_launch_chain_final(const Layout& lay, const Dims& d, const ChainBuf& c, const Work& w,
                               const float* eps, float* xo, float* po, unsigned int* cog, hipStream_t s);

void cmdgen_launch_joint_init(const Layout& lay, const Dims& d, const JointBuf& c, const float* phx, const float* phoh,
                              const float* px, const float* poh, hipStream_t s);
void cmdgen_launch_joint_step(const Layout& lay, const Dims& d, const JointBuf& c, const float* ep, const float* eq, hipStream_t s);
void cmdgen_launch_joint_final(const Layout& lay, const Dims& d, const JointBuf& c, const float* ep, const float* eq,
                               float* xo, float* po, unsigned int* cog, hipStream_t s);

#include "cmdgen_host.h"

std::string g_create_error;

extern "C" const char* cmdgen_version(void) { return "cmdgen_hip 0.1 (gfx950)"; }

extern "C" const char* cmdgen_last_error(const cmdgen_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int cmdgen_create(const cmdgen_config* cfg, int device, cmdgen_handle** out) {
    if (!cfg || !out) return fail(nullptr, CMDGEN_EINVAL, "null argument");
    const int H = cfg->hidden_nf;
    if (!(H == 64 || H == 128 || H == 256 || H == 512))
        return fail(nullptr, CMDGEN_EINVAL, "hidden_nf=%d unsupported: the gfx950 kernels tile 64 columns per wave and are built for 64, 128, 256 and 512", H);
    if (cfg->inv_sublayers < 1 || cfg->inv_sublayers > 8) return fail(nullptr, CMDGEN_EINVAL, "inv_sublayers=%d out of range [1, 8]", cfg->inv_sublayers);
    if (cfg->n_layers < 1 || cfg->n_layers > CMDGEN_MAX_LAYERS) return fail(nullptr, CMDGEN_EINVAL, "n_layers out of range");
    if (cfg->phar_nf < 1 || 2 * cfg->phar_nf > CMDGEN_MAX_SMALL || cfg->residue_nf < 1 ||
        2 * cfg->residue_nf > CMDGEN_MAX_SMALL || cfg->joint_nf < 1 || cfg->joint_nf + 1 > CMDGEN_MAX_SMALL)
        return fail(nullptr, CMDGEN_EINVAL, "feature sizes exceed the small-MLP bound %d", CMDGEN_MAX_SMALL);
    if (cfg->timesteps < 1) return fail(nullptr, CMDGEN_EINVAL, "timesteps < 1");
    if (cfg->update_pocket_coords && cfg->no_com_projection)
        return fail(nullptr, CMDGEN_EINVAL, "update_pocket_coords (joint model) and no_com_projection (SimpleConditionalDDPM) are exclusive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(nullptr, CMDGEN_EHIP, "no usable HIP device %d (found %d)", device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, CMDGEN_EHIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, CMDGEN_EINVAL, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, CMDGEN_EHIP, "hipSetDevice failed");
    cmdgen_handle* h = new cmdgen_handle();
    h->cfg = *cfg; h->device = device;
    Dims& d = h->dims;
    d.P = cfg->phar_nf; d.R = cfg->residue_nf; d.J = cfg->joint_nf; d.H = H; d.L = cfg->n_layers;
    d.condition_time = cfg->condition_time ? 1 : 0; d.dyn = d.J + d.condition_time;
    d.attention = cfg->attention ? 1 : 0; d.use_tanh = cfg->tanh ? 1 : 0; d.no_com = cfg->no_com_projection ? 1 : 0;
    d.joint = cfg->update_pocket_coords ? 1 : 0;
    d.cutoff2 = cfg->edge_cutoff < 0.f ? -1.f : cfg->edge_cutoff * cfg->edge_cutoff;
    d.norm_constant = cfg->norm_constant; d.norm_factor = cfg->normalization_factor; d.coords_range = cfg->coords_range;
    d.S = cfg->inv_sublayers; d.agg_mean = cfg->aggregation_mean ? 1 : 0;
    d.sin = cfg->sin_embedding ? 1 : 0;
    {   // SinusoidsEmbeddingNew.frequencies (egnn_new.py:252): 2 * pi * 4 ** arange(6) / 15 as torch evaluates it in fp32
        const float two_pi = (float)6.283185307179586;
        float p4 = 1.0f;
        for (int k = 0; k < 6; ++k) { d.sin_freq[k] = (two_pi * p4) / 15.0f; p4 *= 4.0f; }
    }
    d.norm_x = cfg->norm_x; d.norm_h = cfg->norm_h; d.bias_h = cfg->bias_h;
    h->n_cus = prop.multiProcessorCount;
    if ((size_t)(8 + d.dyn) * H * sizeof(float) > 64 * 1024) cmdgen_readout_allow_lds((size_t)(8 + d.dyn) * H * sizeof(float));
    h->edge_grid = 2 * prop.multiProcessorCount;      // two 66 KB-LDS workgroups per CU at 64-row tiles
    h->gemm_split = !d.sin;                           // matrix engine of the tiles of >= 32 rows (cmdgen_set_gemm_mode); sin_embedding: the fp32
                                                      // instruction (the split engine's plane builders carry the two scalar distance features only)
    *out = h;
    return CMDGEN_OK;
}

extern "C" void cmdgen_destroy(cmdgen_handle* h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->step_graph) hipGraphExecDestroy(h->step_graph);
    if (h->joint_graph) hipGraphExecDestroy(h->joint_graph);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    if (h->ev_in) hipEventDestroy(h->ev_in);
    if (h->ev_out) hipEventDestroy(h->ev_out);
    free_pool(h->weight_allocs); free_pool(h->layout_allocs); free_pool(h->chain_allocs); free_pool(h->joint_allocs);
    for (int i = 0; i < 2; ++i) { if (h->idx_stage[i]) hipHostFree(h->idx_stage[i]); if (h->idx_ev[i]) hipEventDestroy(h->idx_ev[i]); }
    if (h->h_norm) hipHostFree(h->h_norm);
    if (h->norm_ev) hipEventDestroy(h->norm_ev);
    cmdgen_train_free(h->train);
    delete h;
}

// ---------------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------------
extern "C" int cmdgen_load_weights(cmdgen_handle* h, const char* name, const float* host, size_t n) {
    if (!h || !name || !host) return fail(h, CMDGEN_EINVAL, "null argument");
    h->staged[name].assign(host, host + n);
    h->finalized = false;
    return CMDGEN_OK;
}

static int get_w(cmdgen_handle* h, const std::string& name, size_t n, const std::vector<float>** out) {
    auto it = h->staged.find(name);
    if (it == h->staged.end()) return fail(h, CMDGEN_ESTATE, "missing tensor '%s'", name.c_str());
    if (it->second.size() != n) return fail(h, CMDGEN_EINVAL, "tensor '%s' has %zu values, expected %zu", name.c_str(), it->second.size(), n);
    *out = &it->second;
    return 0;
}

static int upload(cmdgen_handle* h, const std::vector<float>& v, const float** dev) {
    void* p; int rc = dev_alloc(h, h->weight_allocs, &p, v.size() * sizeof(float), false);
    if (rc) return rc;
    if (hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(h, CMDGEN_EHIP, "hipMemcpy H2D failed");
    *dev = (const float*)p;
    return 0;
}

// W[out][ld] rows, columns [c0, c0+in) -> MFMA fragment order (see cmdgen_dev.h)
static std::vector<float> pack_frag(const float* W, int out, int ld, int c0, int in) {
    const int NT = out / 32, KB = in / 8;
    std::vector<float> p((size_t)NT * KB * 64 * 4);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j)
                    p[(((size_t)nt * KB + kb) * 64 + lane) * 4 + j] =
                        W[(size_t)(32 * nt + (lane & 31)) * ld + c0 + 8 * kb + 4 * (lane >> 5) + j];
    return p;
}

// same matrix in v_mfma_f32_16x16x4_f32 fragment order (16-row tiles)
static std::vector<float> pack_frag16(const float* W, int out, int ld, int c0, int in) {
    const int NT = out / 16, KB = in / 16;
    std::vector<float> p((size_t)NT * KB * 64 * 4);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j)
                    p[(((size_t)nt * KB + kb) * 64 + lane) * 4 + j] =
                        W[(size_t)(16 * nt + (lane & 15)) * ld + c0 + 16 * kb + 4 * (lane >> 4) + j];
    return p;
}

// round-to-nearest-even bf16 of a finite float (weights are finite: cmdgen_load_weights' callers check)
static inline unsigned short bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_val(unsigned short b) { const uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

// same matrix as three bf16 pieces per weight (w = w0 + w1 + w2 exactly up to 2^-24 |w|) in v_mfma_f32_32x32x16_bf16
// fragment order, the three pieces of a fragment contiguous (cmdgen_split.h)
static std::vector<unsigned short> pack_split(const float* W, int out, int ld, int c0, int in) {
    const int NT = out / 32, KB = in / 16;
    std::vector<unsigned short> p((size_t)NT * KB * 3 * 64 * 8);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const float w = W[(size_t)(32 * nt + (lane & 31)) * ld + c0 + 16 * kb + 8 * (lane >> 5) + j];
                    const unsigned short h0 = bf16_rne(w); const float r1 = w - bf16_val(h0);
                    const unsigned short h1 = bf16_rne(r1); const float r2 = r1 - bf16_val(h1);
                    const unsigned short h2 = bf16_rne(r2);
                    const size_t base = (((size_t)nt * KB + kb) * 3) * 64 * 8;
                    p[base + (0 * 64 + lane) * 8 + j] = h0; p[base + (1 * 64 + lane) * 8 + j] = h1; p[base + (2 * 64 + lane) * 8 + j] = h2;
                }
    return p;
}

// the same three pieces in v_mfma_f32_16x16x32_bf16 fragment order (16-row tiles; k order inside a block of 32 as the
// A-side reads it: lane group g holds k = 4g .. 4g+3 and 16+4g .. 16+4g+3, cmdgen_split.h)
static std::vector<unsigned short> pack_split16(const float* W, int out, int ld, int c0, int in) {
    const int NT = out / 16, KB = in / 32;
    std::vector<unsigned short> p((size_t)NT * KB * 3 * 64 * 8);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int g = lane >> 4;
                    const int k = 32 * kb + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
                    const float w = W[(size_t)(16 * nt + (lane & 15)) * ld + c0 + k];
                    const unsigned short h0 = bf16_rne(w); const float r1 = w - bf16_val(h0);
                    const unsigned short h1 = bf16_rne(r1); const float r2 = r1 - bf16_val(h1);
                    const unsigned short h2 = bf16_rne(r2);
                    const size_t base = (((size_t)nt * KB + kb) * 3) * 64 * 8;
                    p[base + (0 * 64 + lane) * 8 + j] = h0; p[base + (1 * 64 + lane) * 8 + j] = h1; p[base + (2 * 64 + lane) * 8 + j] = h2;
                }
    return p;
}

// same matrix as TWO fp16 pieces of (scale * w) per weight in v_mfma_f32_32x32x16_f16 fragment order, the two pieces of a fragment
// contiguous: Wh[((nt * KB16 + kb) * 2 + s) * 64 + lane] = 8 halves (cmdgen_split.h, "half" engine).  scale is a power of two chosen
// so that the largest weight lands in [2^11, 2^12): both pieces of every weight that matters are normal fp16 numbers.
static std::vector<unsigned short> pack_half(const float* W, int out, int ld, int c0, int in, float* scale) {
    float mx = 0.f;
    for (int o = 0; o < out; ++o) for (int k = 0; k < in; ++k) mx = std::max(mx, std::fabs(W[(size_t)o * ld + c0 + k]));
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) { int ex; std::frexp(mx, &ex); e = 12 - ex; }      // mx * 2^e in [2^11, 2^12)
    e = std::max(-40, std::min(40, e));
    const float sc = std::ldexp(1.0f, e);
    *scale = sc;
    const int NT = out / 32, KB = in / 16;
    std::vector<unsigned short> p((size_t)NT * KB * 2 * 64 * 8);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const float w = W[(size_t)(32 * nt + (lane & 31)) * ld + c0 + 16 * kb + 8 * (lane >> 5) + j] * sc;
                    const _Float16 h0 = (_Float16)w; const float r1 = w - (float)h0;
                    const _Float16 h1 = (_Float16)r1;
                    unsigned short u0, u1; memcpy(&u0, &h0, 2); memcpy(&u1, &h1, 2);
                    const size_t base = (((size_t)nt * KB + kb) * 2) * 64 * 8;
                    p[base + (0 * 64 + lane) * 8 + j] = u0; p[base + (1 * 64 + lane) * 8 + j] = u1;
                }
    return p;
}

// the same two fp16 pieces of (scale * w) in v_mfma_f32_16x16x32_f16 fragment order (16-row tiles; k order inside a block of 32 as pack_split16)
static std::vector<unsigned short> pack_half16(const float* W, int out, int ld, int c0, int in, float sc) {
    const int NT = out / 16, KB = in / 32;
    std::vector<unsigned short> p((size_t)NT * KB * 2 * 64 * 8);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int g = lane >> 4;
                    const int k = 32 * kb + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
                    const float w = W[(size_t)(16 * nt + (lane & 15)) * ld + c0 + k] * sc;
                    const _Float16 h0 = (_Float16)w; const float r1 = w - (float)h0;
                    const _Float16 h1 = (_Float16)r1;
                    unsigned short u0, u1; memcpy(&u0, &h0, 2); memcpy(&u1, &h1, 2);
                    const size_t base = (((size_t)nt * KB + kb) * 2) * 64 * 8;
                    p[base + (0 * 64 + lane) * 8 + j] = u0; p[base + (1 * 64 + lane) * 8 + j] = u1;
                }
    return p;
}

static int upload_pack(cmdgen_handle* h, const float* W, int out, int in, WPack* wp) {
    const float* dp;
    std::vector<float> p = pack_frag(W, out, in, 0, in);
    int r = upload(h, p, &dp); if (r) return r; wp->w32 = (const float4*)dp;
    p = pack_frag16(W, out, in, 0, in);
    r = upload(h, p, &dp); if (r) return r; wp->w16 = (const float4*)dp;
    const std::vector<unsigned short> ps = pack_split(W, out, in, 0, in);
    void* q; r = dev_alloc(h, h->weight_allocs, &q, ps.size() * sizeof(unsigned short), false); if (r) return r;
    if (hipMemcpy(q, ps.data(), ps.size() * sizeof(unsigned short), hipMemcpyHostToDevice) != hipSuccess) return fail(h, CMDGEN_EHIP, "hipMemcpy H2D failed");
    wp->ws = q;
    wp->ws16 = nullptr;
    {
        float sc = 1.0f;
        const std::vector<unsigned short> ph = pack_half(W, out, in, 0, in, &sc);
        r = dev_alloc(h, h->weight_allocs, &q, ph.size() * sizeof(unsigned short), false); if (r) return r;
        if (hipMemcpy(q, ph.data(), ph.size() * sizeof(unsigned short), hipMemcpyHostToDevice) != hipSuccess) return fail(h, CMDGEN_EHIP, "hipMemcpy H2D failed");
        wp->wh = q; wp->wh_scale = sc; wp->wh_inv = 1.0f / sc;
        wp->wh16 = nullptr;
        if (in % 128 == 0) {
            const std::vector<unsigned short> ph16 = pack_half16(W, out, in, 0, in, sc);
            r = dev_alloc(h, h->weight_allocs, &q, ph16.size() * sizeof(unsigned short), false); if (r) return r;
            if (hipMemcpy(q, ph16.data(), ph16.size() * sizeof(unsigned short), hipMemcpyHostToDevice) != hipSuccess) return fail(h, CMDGEN_EHIP, "hipMemcpy H2D failed");
            wp->wh16 = q;
        }
    }
    if (in % 128 == 0) {        // the 16-row split GEMM walks four k-blocks of 32 per iteration
        const std::vector<unsigned short> p16 = pack_split16(W, out, in, 0, in);
        r = dev_alloc(h, h->weight_allocs, &q, p16.size() * sizeof(unsigned short), false); if (r) return r;
        if (hipMemcpy(q, p16.data(), p16.size() * sizeof(unsigned short), hipMemcpyHostToDevice) != hipSuccess) return fail(h, CMDGEN_EHIP, "hipMemcpy H2D failed");
        wp->ws16 = q;
    }
    return 0;
}

extern "C" int cmdgen_finalize_weights(cmdgen_handle* h) {
    if (!h) return CMDGEN_EINVAL;
    hipSetDevice(h->device);
    if (h->step_graph) { hipGraphExecDestroy(h->step_graph); h->step_graph = nullptr; }
    if (h->joint_graph) { hipGraphExecDestroy(h->joint_graph); h->joint_graph = nullptr; }
    free_pool(h->weight_allocs);
    h->layers.clear();
    const Dims& d = h->dims;
    const int H = d.H, T = h->cfg.timesteps;
    const std::vector<float>* v; int rc;
#define GET(name, n) do { rc = get_w(h, name, (size_t)(n), &v); if (rc) return rc; } while (0)
#define UP(dst) do { rc = upload(h, *v, &(dst)); if (rc) return rc; } while (0)
    GET("gamma.gamma", T + 1); h->gamma = *v;
    const std::string dy = "dynamics.";
    SmallW& s = h->small;
    GET(dy + "phar_encoder.0.weight", 2 * d.P * d.P); UP(s.pe0_w); GET(dy + "phar_encoder.0.bias", 2 * d.P); UP(s.pe0_b);
    GET(dy + "phar_encoder.2.weight", d.J * 2 * d.P); UP(s.pe2_w); GET(dy + "phar_encoder.2.bias", d.J); UP(s.pe2_b);
    GET(dy + "phar_decoder.0.weight", 2 * d.P * d.J); UP(s.pd0_w); GET(dy + "phar_decoder.0.bias", 2 * d.P); UP(s.pd0_b);
    GET(dy + "phar_decoder.2.weight", d.P * 2 * d.P); UP(s.pd2_w); GET(dy + "phar_decoder.2.bias", d.P); UP(s.pd2_b);
    GET(dy + "residue_encoder.0.weight", 2 * d.R * d.R); UP(s.re0_w); GET(dy + "residue_encoder.0.bias", 2 * d.R); UP(s.re0_b);
    GET(dy + "residue_encoder.2.weight", d.J * 2 * d.R); UP(s.re2_w); GET(dy + "residue_encoder.2.bias", d.J); UP(s.re2_b);
    GET(dy + "residue_decoder.0.weight", 2 * d.R * d.J); UP(s.rd0_w); GET(dy + "residue_decoder.0.bias", 2 * d.R); UP(s.rd0_b);
    GET(dy + "residue_decoder.2.weight", d.R * 2 * d.R); UP(s.rd2_w); GET(dy + "residue_decoder.2.bias", d.R); UP(s.rd2_b);
    {   // the encoders' tensors once more, contiguous, in the order k_embed lays them out in LDS
        std::vector<float> pack;
        for (const char* nm : {"phar_encoder.0.weight", "phar_encoder.0.bias", "phar_encoder.2.weight", "phar_encoder.2.bias",
                               "residue_encoder.0.weight", "residue_encoder.0.bias", "residue_encoder.2.weight", "residue_encoder.2.bias"}) {
            auto it = h->staged.find(dy + nm);
            if (it == h->staged.end()) return fail(h, CMDGEN_ESTATE, "missing tensor '%s%s'", dy.c_str(), nm);
            pack.insert(pack.end(), it->second.begin(), it->second.end());
        }
        rc = upload(h, pack, &s.enc_pack); if (rc) return rc;
    }
    {   // embedding [H][dyn] -> transposed [dyn][H]
        GET(dy + "egnn.embedding.weight", H * d.dyn);
        std::vector<float> t((size_t)H * d.dyn);
        for (int c = 0; c < H; ++c) for (int k = 0; k < d.dyn; ++k) t[(size_t)k * H + c] = (*v)[(size_t)c * d.dyn + k];
        rc = upload(h, t, &s.emb_wT); if (rc) return rc;
        GET(dy + "egnn.embedding.bias", H); UP(s.emb_b);
    }
    {   // embedding_out [dyn][H] -> transposed [H][dyn]
        GET(dy + "egnn.embedding_out.weight", d.dyn * H);
        std::vector<float> t((size_t)H * d.dyn);
        for (int j = 0; j < d.dyn; ++j) for (int k = 0; k < H; ++k) t[(size_t)k * d.dyn + j] = (*v)[(size_t)j * H + k];
        rc = upload(h, t, &s.embo_wT); if (rc) return rc;
        GET(dy + "egnn.embedding_out.bias", d.dyn); UP(s.embo_b);
    }
    const int nfeat = d.sin ? 24 : 2;            // edge features behind [h_row | h_col]: radial + d0, or 12 + 12 sinusoids (egnn_new.py:174-176)
    const int ld1 = 2 * H + nfeat;
    // one LayerW per GCL ("unit" b * S + sub, egnn_new.py:127-131); the block's EquivariantUpdate rides with its LAST GCL (the node kernel of
    // that unit projects P_c | Q_c), earlier units of a block carry no coordinate weights
    for (int b = 0; b < d.L; ++b)
    for (int sub = 0; sub < d.S; ++sub) {
        LayerW lw{};
        const std::string g = dy + "egnn.e_block_" + std::to_string(b) + ".gcl_" + std::to_string(sub) + ".";
        const std::string c = dy + "egnn.e_block_" + std::to_string(b) + ".gcl_equiv.";
        auto split_first = [&](const std::string& wname, const std::string& bname, WPack* Wpq,
                               const float** bias, const float** wr, const float** wd, const float** we) -> int {
            const std::vector<float>* w; int r = get_w(h, wname, (size_t)H * ld1, &w); if (r) return r;
            // stack [A ; B] as a [2H][H] matrix: rows 0..H-1 = columns 0..H-1 (h_row), rows H.. = columns H..2H-1 (h_col)
            std::vector<float> AB((size_t)2 * H * H);
            std::vector<float> vr(H), vd(H);
            for (int o = 0; o < H; ++o) {
                for (int k = 0; k < H; ++k) {
                    AB[(size_t)o * H + k] = (*w)[(size_t)o * ld1 + k];
                    AB[(size_t)(H + o) * H + k] = (*w)[(size_t)o * ld1 + H + k];
                }
                vr[o] = (*w)[(size_t)o * ld1 + 2 * H]; vd[o] = (*w)[(size_t)o * ld1 + 2 * H + 1];
            }
            r = upload_pack(h, AB.data(), 2 * H, H, Wpq); if (r) return r;
            r = upload(h, vr, wr); if (r) return r; r = upload(h, vd, wd); if (r) return r;
            *we = nullptr;
            if (d.sin) {                   // the 24 feature columns, transposed [24][H]
                std::vector<float> wt((size_t)24 * H);
                for (int o = 0; o < H; ++o) for (int k = 0; k < 24; ++k) wt[(size_t)k * H + o] = (*w)[(size_t)o * ld1 + 2 * H + k];
                r = upload(h, wt, we); if (r) return r;
            }
            const std::vector<float>* bb; r = get_w(h, bname, H, &bb); if (r) return r;
            return upload(h, *bb, bias);
        };
        auto square = [&](const std::string& wname, int in, WPack* Wp) -> int {
            const std::vector<float>* w; int r = get_w(h, wname, (size_t)H * in, &w); if (r) return r;
            return upload_pack(h, w->data(), H, in, Wp);
        };
        rc = split_first(g + "edge_mlp.0.weight", g + "edge_mlp.0.bias", &lw.Wpq_e, &lw.b1, &lw.wr_e, &lw.wd_e, &lw.we_e); if (rc) return rc;
        rc = square(g + "edge_mlp.2.weight", H, &lw.W2); if (rc) return rc;
        GET(g + "edge_mlp.2.bias", H); UP(lw.b2);
        if (d.attention) {
            GET(g + "att_mlp.0.weight", H); UP(lw.wa);
            GET(g + "att_mlp.0.bias", 1); UP(lw.ba);
        } else { lw.wa = lw.b2; lw.ba = lw.b2; }
        rc = square(g + "node_mlp.0.weight", 2 * H, &lw.W3); if (rc) return rc;
        GET(g + "node_mlp.0.bias", H); UP(lw.b3);
        rc = square(g + "node_mlp.2.weight", H, &lw.W4); if (rc) return rc;
        GET(g + "node_mlp.2.bias", H); UP(lw.b4);
        if (sub == d.S - 1) {
            rc = split_first(c + "coord_mlp.0.weight", c + "coord_mlp.0.bias", &lw.Wpq_c, &lw.b6, &lw.wr_c, &lw.wd_c, &lw.we_c); if (rc) return rc;
            rc = square(c + "coord_mlp.2.weight", H, &lw.W7); if (rc) return rc;
            GET(c + "coord_mlp.2.bias", H); UP(lw.b7);
            GET(c + "coord_mlp.4.weight", H); UP(lw.w5);
        } else {                       // never multiplied (the node kernel skips the projection); valid pointers for the bias prefetches
            lw.Wpq_c = lw.Wpq_e; lw.b6 = lw.b1; lw.wr_c = lw.wr_e; lw.wd_c = lw.wd_e; lw.we_c = lw.we_e; lw.W7 = lw.W2; lw.b7 = lw.b2; lw.w5 = lw.b2;
        }
        h->layers.push_back(lw);
    }
#undef GET
#undef UP
    h->finalized = true;
    h->user_coef_K = -1; h->chain_K = -1;      // a new gamma table invalidates any step table
    h->joint_steps = -1; h->joint_key.clear();
    return CMDGEN_OK;
}

// ---------------------------------------------------------------------------------
// layout
// ---------------------------------------------------------------------------------
// dynamic LDS of k_edge_count / k_edge_write: float4 position + two ints per node (kernels_egnn.hip)
static inline size_t edge_lds_bytes(int max_n) { return (size_t)max_n * (sizeof(float4) + 3 * sizeof(int)); }
static const size_t kEdgeLdsMax = 156 * 1024;      // 160 KiB per CU minus k_edge_write's small static arrays
void cmdgen_edge_kernels_allow_lds(size_t bytes);  // kernels_egnn.hip: hipFuncSetAttribute above the 64 KiB default

// Rows per tile and grids of the evaluation's launches for the current layout (cmdgen_set_layout, and again after cmdgen_set_option /
// cmdgen_set_gemm_mode).
static void pick_tiles(cmdgen_handle* h) {
    // Rows per tile: the largest tile that still gives every CU a few workgroups.  Edge counts are only known on
    // the device, so they are estimated from the layout for the geometry a trained model holds and every chain starts
    // from - the phar points inside the pocket (measured on CrossDocked-shaped pockets, bench.py's steady_state_evaluation:
    // C-alpha 9.2 neighbours per node within 6 A and 15 coordinate edges per phar node; full-atom 36 and 54).  A chain of
    // an untrained model drifts to fewer edges; the persistent edge grids just find fewer tiles then.
    double e_est = 0.0, ec_est = 0.0;
    const int B = (int)h->cur_nphar.size();
    const Dims& d = h->dims;
    const int64_t N = (int64_t)h->lay.N;
    const int64_t* nph = h->cur_nphar.data(); const int64_t* npk = h->cur_npocket.data();
    for (int b = 0; b < B; ++b) {
        const double n = (double)(nph[b] + npk[b]);
        const bool full = h->cfg.edge_cutoff < 0.f;
        const double deg = full ? n : (n <= 128.0 ? 9.0 : 36.0);
        const double dnode = deg < n ? deg : n;
        double dphar = full ? n : 0.6 * (double)nph[b] + (n <= 128.0 ? 0.15 : 0.13) * (double)npk[b];   // receivers that move
        if (dphar > n) dphar = n;
        e_est += n * dnode;
        ec_est += d.joint ? n * dnode : (double)nph[b] * dphar;                   // joint: every receiver moves
    }
    // thresholds from sweeps on MI355X (fp32 engine: profiles/r01_tile_sweep.txt; split engine:
    // profiles/r02_o_tile_sweep_split.txt, r02_z_tiles_trained_geometry.txt)
    auto pick = [&](double rows) { return rows / 64.0 >= 3.0 * h->n_cus ? 64 : (rows / 32.0 >= 1.5 * h->n_cus ? 32 : 16); };
    h->node_mt = pick((double)N); h->edge_mt = pick(e_est); h->coord_mt = pick(ec_est);
    if (h->gemm_split) {
        // node kernel: 32-row tiles (two LDS images) as soon as they put a workgroup on 0.6 of the CUs (96 C-alpha pockets),
        // never 64 rows; coordinate kernel: 64-row tiles only for very long lists - its list shrinks to a few tiles when a
        // chain drifts, and a lone 64-row tile costs 15 us where a 32-row one costs 10
        h->node_mt = (double)N / 32.0 >= 0.6 * h->n_cus ? 32 : 16;
        h->coord_mt = ec_est / 64.0 >= 6.0 * h->n_cus ? 64 : (ec_est / 32.0 >= 1.5 * h->n_cus ? 32 : 16);
    }
    // long lists on the split engine: the 128-row kernels of kernels_edge128.hip (every workgroup owns one chunk of the list; same-box
    // A/B at 256 C-alpha pockets: messages -3 %, coordinate list -14 %; full-atom pockets: level; profiles/r04_d)
    // (the half engine as make_launch resolves it)
    const int he_opt = (int)opt_of(h, "half_engine", 1);
    const bool half = h->gemm_split && d.H == 256 && (he_opt == 2 || (he_opt == 1 && d.cutoff2 >= 0.f));
    if (h->gemm_split && d.H == 256) {
        if (e_est / 64.0 >= 4.0 * h->n_cus) h->edge_mt = 128;
        // on the half engine the chunked 128-row message kernel wins from ~48 C-alpha pockets (64: 29.6 vs 32.4 us per launch for the 32-row full-K
        // tiles, 96: 37.7 vs 58.7 for the 64-row plane tiles; profiles/r05_t); the coordinate list stays on 32-row tiles until it is long
        if (half && e_est >= 96.0 * h->n_cus) h->edge_mt = 128;
        // ... and more 16-row node tiles than CUs means two rounds of k_node16w where 64-row plane tiles need one
        if (half && (N + 15) / 16 > h->n_cus) h->node_mt = 32;
        // the 32-row full-K coordinate tiles run on the half engine, 16-row tiles on the fp32 instruction: 32 rows from a quarter of a tile per CU
        // (48 pockets: 32.9 us per launch on 16-row tiles, 64 pockets: 18.3 on 32-row ones)
        if (half && h->coord_mt == 16 && ec_est / 32.0 >= 0.25 * h->n_cus) h->coord_mt = 32;
        if (half && h->edge_mt == 16 && e_est / 32.0 >= 0.25 * h->n_cus) h->edge_mt = 32;        // (16 pockets: 29.0 us on 16-row tiles; 32 pockets: 21.2 on 32-row ones)
        if (ec_est / 32.0 >= 3.0 * h->n_cus) h->coord_mt = 128;      // (from 128 C-alpha pockets: profiles/r04_h)
        // dense samples (full-atom pockets: 36 neighbours per node, ~60 coordinate edges per phar point while the points sit at the pocket centre):
        // a receiver's edges outnumber the rows of a 16- / 32-row tile, its sum would be three or more float-atomic partials whose order the
        // hardware picks - the 128-row kernels (variable tiles, >= 128-row chunks) keep it at two, so full-atom chains are reproducible run to run
        if (h->lay.max_n > 128) { h->edge_mt = 128; h->coord_mt = 128; }
        // joint chains noise the pocket nodes too: over the first steps a C-alpha sample is nearly fully connected (~48 k edges per evaluation
        // on average at 64 pockets where the layout estimate says 34 k), and over those lists the 32-row full-K message tiles win - same-box
        // chains at 64 / 128 / 256 pockets: +8.6 / +5.0 / +4.7 % (profiles/r06_m); the coordinate list (the same edges) stays on 128 rows
        else if (half && d.joint && h->edge_mt == 128) h->edge_mt = 32;
    }
    // (the fp32 instruction / other widths have no 128-row kernels: their largest tile keeps most dense receivers at two partials)
    if (!(h->gemm_split && d.H == 256) && h->lay.max_n > 128) { h->edge_mt = 64; h->coord_mt = 64; }
    h->node_mt = (int)opt_of(h, "node_mt", h->node_mt);
    h->edge_mt = (int)opt_of(h, "edge_mt", h->edge_mt);
    h->coord_mt = (int)opt_of(h, "coord_mt", h->coord_mt);
    // grids of the persistent-style edge kernels: enough workgroups for the estimated tile count, capped at
    // what is co-resident per CU (2 at 64-row tiles, 4 below); surplus tiles are picked up by the loop
    auto grid_for = [&](double rows, int mt) {
        const double tiles = rows / mt + 1.0;
        const int cap = (mt >= 64 ? 2 : 4) * h->n_cus;
        int g = (int)(tiles * 1.25) + 8;
        return g < h->n_cus / 4 ? h->n_cus / 4 : (g > cap ? cap : g);
    };
    h->edge_grid = grid_for(e_est, h->edge_mt);
    h->coord_grid = grid_for(ec_est, h->coord_mt);
    // the 128-row kernels' fused main loop pays once a workgroup (two per CU) walks more than one tile: same-box chains, profiles/r06_f
    // (64 C-alpha pockets, one 96-row tile per workgroup: -1.6 %; 96 pockets, one 128-row tile: +1.2 %; 128 pockets: +2 %; 256: +3 %; full-atom: +8 %)
    h->e128_fused = (e_est > 160.0 * h->n_cus ? 1 : 0) | (ec_est > 160.0 * h->n_cus ? 2 : 0);
    if (opt_set(h, "e128_fused")) h->e128_fused = (int)opt_of(h, "e128_fused", 3) & 3;
    if (opt_set(h, "edge_wgs_per_cu")) h->edge_grid = (int)opt_of(h, "edge_wgs_per_cu", 2) * h->n_cus;
    if (opt_set(h, "coord_wgs_per_cu")) h->coord_grid = (int)opt_of(h, "coord_wgs_per_cu", 2) * h->n_cus;
    if (h->node_mt != 64 && h->node_mt != 32 && h->node_mt != 16) h->node_mt = 64;
    for (int* m : {&h->edge_mt, &h->coord_mt}) if (*m != 128 && *m != 64 && *m != 32 && *m != 16) *m = 64;     // 128: kernels_edge128.hip
}

static int set_layout_impl(cmdgen_handle* h, int64_t batch, const int64_t* nph, const int64_t* npk, bool on_stream, hipStream_t stream);
extern "C" int cmdgen_set_layout(cmdgen_handle* h, int64_t batch, const int64_t* nph, const int64_t* npk) {
    return set_layout_impl(h, batch, nph, npk, false, nullptr);
}
extern "C" int cmdgen_set_layout_on_stream(cmdgen_handle* h, int64_t batch, const int64_t* nph, const int64_t* npk, cmdgen_stream stream) {
    return set_layout_impl(h, batch, nph, npk, true, (hipStream_t)stream);
}
static int set_layout_impl(cmdgen_handle* h, int64_t batch, const int64_t* nph, const int64_t* npk, bool on_stream, hipStream_t stream) {
    if (!h || batch < 1 || !nph || !npk) return fail(h, CMDGEN_EINVAL, "bad layout arguments");
    if (h->have_layout && (int64_t)h->cur_nphar.size() == batch &&
        memcmp(h->cur_nphar.data(), nph, batch * sizeof(int64_t)) == 0 &&
        memcmp(h->cur_npocket.data(), npk, batch * sizeof(int64_t)) == 0)
        return CMDGEN_OK;
    hipSetDevice(h->device);
    const Dims& d = h->dims;
    const int B = (int)batch;
    std::vector<int> vph(B), vpk(B), bph(B), bpk(B);
    int64_t Nl = 0, Np = 0, ecap = 0, eccap = 0; int max_n = 0;
    for (int b = 0; b < B; ++b) {
        if (nph[b] < 0 || npk[b] < 0) return fail(h, CMDGEN_EINVAL, "negative node count");
        vph[b] = (int)nph[b]; vpk[b] = (int)npk[b]; bph[b] = (int)Nl; bpk[b] = (int)Np;
        Nl += nph[b]; Np += npk[b];
        const int64_t n = nph[b] + npk[b];
        ecap += n * n; eccap += (d.joint ? n : nph[b]) * n;   // dense bound per sample: never overflows
        if (n > max_n) max_n = (int)n;
    }
    const int64_t N = Nl + Np;
    if (N < 1 || ecap > (int64_t)2000000000) return fail(h, CMDGEN_EINVAL, "batch too large for int32 edge indexing (dense bound %lld)", (long long)ecap);
    if (edge_lds_bytes(max_n) > kEdgeLdsMax)
        return fail(h, CMDGEN_EINVAL, "a sample has %d nodes; the per-sample neighbour search keeps positions and offsets in LDS (%d B per node, at most %d nodes)",
                    max_n, 28, (int)(kEdgeLdsMax / 28));
    std::vector<int> ns(N);
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < vph[b]; ++i) ns[bph[b] + i] = b;
        for (int i = 0; i < vpk[b]; ++i) ns[Nl + bpk[b] + i] = b;
    }
    // graphs bake the layout into their kernel arguments; chain buffers are sized by it.  Kernels of the previous
    // layout may still be reading the index arrays rewritten below: wait for the streams this handle has been
    // given (ordering contract in include/cmdgen_hip.h; torch's side streams are non-blocking, so the null-stream
    // copies below are not ordered against them by themselves).
    // (cmdgen_set_layout_on_stream: when the new layout fits the workspaces nothing is waited for - the index arrays go to
    // the OTHER of two device blocks, copied from pinned staging in stream order, so kernels of the previous layout that
    // are still running on `stream` keep reading theirs)
    const bool fits_now = h->cap_B >= B && h->cap_Nl >= Nl && h->cap_Np >= Np && h->cap_N >= N && h->cap_e >= ecap && h->cap_ec >= eccap;
    const bool no_wait = on_stream && fits_now && h->have_layout && h->chain_allocs.empty() && h->joint_allocs.empty() &&
                         (h->last_stream == stream) && !h->step_graph && !h->joint_graph;
    if (!no_wait) {
        if (h->own_stream) hipStreamSynchronize(h->own_stream);
        if (h->have_layout) hipStreamSynchronize(h->last_stream);
        if (on_stream) hipStreamSynchronize(stream);
    }
    if (h->step_graph) { hipGraphExecDestroy(h->step_graph); h->step_graph = nullptr; }
    if (h->joint_graph) { hipGraphExecDestroy(h->joint_graph); h->joint_graph = nullptr; }
    int rc; void* p;
    Layout& L = h->lay; Work& w = h->work;
    // Workspaces are capacity-based: a new batch that fits the current capacities (every training step and every
    // dataset batch has its own ragged layout) only re-uploads the small index arrays - no hipMalloc/hipFree.
    const bool fits = h->cap_B >= B && h->cap_Nl >= Nl && h->cap_Np >= Np && h->cap_N >= N && h->cap_e >= ecap && h->cap_ec >= eccap;
    if (!fits) {
        hipDeviceSynchronize();
        free_pool(h->layout_allocs); free_pool(h->chain_allocs); h->chain_K = -1;
        free_pool(h->joint_allocs); h->joint_steps = -1; h->joint_key.clear();
        cmdgen_train_free(h->train); h->train = nullptr;
        h->have_layout = false;
        auto grow = [](int64_t v) { return v + v / 4 + 64; };
        const int64_t cB = h->cap_B ? grow(B) : B, cNl = h->cap_B ? grow(Nl) : Nl, cNp = h->cap_B ? grow(Np) : Np;
        const int64_t cN = cNl + cNp, ce = h->cap_B ? grow(ecap) : ecap, cec = h->cap_B ? grow(eccap) : eccap;
        const int64_t cNm = d.joint ? cN : cNl;
#define ALLOC(dst, type, count, zero) do { rc = dev_alloc(h, h->layout_allocs, &p, (size_t)(count) * sizeof(type), zero); if (rc) return rc; dst = (type*)p; } while (0)
        // the index arrays live in ONE block, [gid (int64) | num_phar | num_pocket | phar_base | pocket_base | node_sample]: a new
        // layout (every training step has its own) is one host-to-device copy instead of six
        {
            h->idx_ints = 6 * cB + cN;
            for (int i = 0; i < 2; ++i) {
                ALLOC(h->idx_blk[i], int, h->idx_ints, true);
                if (h->idx_stage[i]) hipHostFree(h->idx_stage[i]);
                HIPCHK(h, hipHostMalloc((void**)&h->idx_stage[i], (size_t)h->idx_ints * sizeof(int), hipHostMallocDefault));
                if (!h->idx_ev[i]) HIPCHK(h, hipEventCreateWithFlags(&h->idx_ev[i], hipEventDisableTiming));
            }
            h->idx_cur = 0;
        }
        const size_t H = d.H;
        ALLOC(w.X0, float4, cNm, true); ALLOC(w.XP, float4, cNp, true);
        ALLOC(w.XL, float4, (size_t)d.L * cNm, true); ALLOC(w.ACC, float4, (size_t)d.L * cNm, true);
        ALLOC(w.h, float, cN * H, true); ALLOC(w.P, float, cN * H, true); ALLOC(w.Q, float, cN * H, true);
        ALLOC(w.Pc, float, cN * H, true); ALLOC(w.Qc, float, cN * H, true); ALLOC(w.agg, float, cN * H, true);
        ALLOC(w.adiv, float, cN, true); ALLOC(w.degL, int, cN, true); ALLOC(w.need_qc, int, cN, true); ALLOC(w.pocketE, int, cB, true); ALLOC(w.pocketEph, int, cB, true);
        ALLOC(w.pocketEns, int, cB, true); ALLOC(w.pocketEnsQ, int, cB, true);
        ALLOC(w.erow, int, ce, false); ALLOC(w.ecol, int, ce, false); ALLOC(w.ed0, float, ce, false); ALLOC(w.ehop, int, ce, false);
        ALLOC(w.crow, int, cec, false); ALLOC(w.ccol, int, cec, false); ALLOC(w.cd0, float, cec, false);
        ALLOC(w.totals, int, 4, true); ALLOC(w.counters, unsigned long long, 8, true); ALLOC(w.nan_flag, int, 4, true);
        ALLOC(w.eps_tmp, float, (size_t)cNl * (3 + d.P), true);
        ALLOC(w.dbg, unsigned long long, 64, true);
#undef ALLOC
        h->cap_B = cB; h->cap_Nl = cNl; h->cap_Np = cNp; h->cap_N = cN; h->cap_e = ce; h->cap_ec = cec;
    } else {
        // chain / joint buffers are sized by the exact layout and cheap: rebuilt on the next chain
        free_pool(h->chain_allocs); h->chain_K = -1;
        free_pool(h->joint_allocs); h->joint_steps = -1; h->joint_key.clear();
    }
    if (edge_lds_bytes(max_n) > 64 * 1024) cmdgen_edge_kernels_allow_lds(edge_lds_bytes(max_n));
    L.B = B; L.Nl = (int)Nl; L.Np = (int)Np; L.N = (int)N; L.max_n = max_n;
    L.Nm = d.joint ? (int)N : (int)Nl;
    {
        const int64_t cB = h->cap_B;
        const int cur = (h->idx_cur ^= 1);
        int* blk = h->idx_blk[cur];
        h->d_gid = reinterpret_cast<int64_t*>(blk); L.pocket_gid = h->d_gid;
        L.num_phar = blk + 2 * cB; L.num_pocket = blk + 3 * cB; L.phar_base = blk + 4 * cB; L.pocket_base = blk + 5 * cB;
        L.node_sample = blk + 6 * cB;
        int* stage = h->idx_stage[cur];
        hipEventSynchronize(h->idx_ev[cur]);              // the copy that last used this staging buffer (two layouts ago) is long done
        memset(stage, 0, (size_t)(6 * cB) * sizeof(int));
        int64_t* gid = reinterpret_cast<int64_t*>(stage);
        for (int b = 0; b < B; ++b) {
            gid[b] = b;
            stage[2 * cB + b] = vph[b]; stage[3 * cB + b] = vpk[b]; stage[4 * cB + b] = bph[b]; stage[5 * cB + b] = bpk[b];
        }
        memcpy(stage + 6 * cB, ns.data(), (size_t)N * sizeof(int));
        const size_t bytes = (size_t)(6 * cB + N) * sizeof(int);
        if (no_wait) {
            HIPCHK(h, hipMemcpyAsync(blk, stage, bytes, hipMemcpyHostToDevice, stream));
            HIPCHK(h, hipEventRecord(h->idx_ev[cur], stream));
            HIPCHK(h, hipMemsetAsync(w.agg, 0, (size_t)N * d.H * sizeof(float), stream));
            HIPCHK(h, hipMemsetAsync(w.totals, 0, 4 * sizeof(int), stream));
        } else {
            // (plain hipMemcpy: ordered after all earlier work of the blocking streams that may still read the old arrays)
            HIPCHK(h, hipMemcpy(blk, stage, bytes, hipMemcpyHostToDevice));
            if (fits) {   // reused buffers: restore the invariants a fresh (zeroed) workspace has
                HIPCHK(h, hipMemset(w.agg, 0, (size_t)N * d.H * sizeof(float)));
                HIPCHK(h, hipMemset(w.totals, 0, 4 * sizeof(int)));
            }
        }
    }
    h->ecap = ecap; h->eccap = eccap;
    h->cur_nphar.assign(nph, nph + B); h->cur_npocket.assign(npk, npk + B);
    pick_tiles(h);
    h->have_layout = true;
    return CMDGEN_OK;
}

int check_ready(cmdgen_handle* h) {
    if (!h) return CMDGEN_EINVAL;
    if (!h->finalized) return fail(h, CMDGEN_ESTATE, "weights not finalised (cmdgen_finalize_weights)");
    if (!h->have_layout) return fail(h, CMDGEN_ESTATE, "no batch layout (cmdgen_set_layout)");
    return 0;
}

// Entry of every call that queues evaluation work: readiness, device, the stream for cmdgen_set_layout's ordering
// contract, and the workspace invariant "agg is zero between blocks" after a debug prefix run.
int begin_work(cmdgen_handle* h, hipStream_t s) {
    if (!h) return CMDGEN_EINVAL;
    if (!h->have_layout) return fail(h, CMDGEN_ESTATE, "no batch layout (cmdgen_set_layout)");
    hipSetDevice(h->device);
    h->last_stream = s;
    if (h->agg_dirty) {
        HIPCHK(h, hipMemsetAsync(h->work.agg, 0, (size_t)h->lay.N * h->dims.H * sizeof(float), s));
        h->agg_dirty = false;
    }
    return 0;
}

EvalLaunch make_launch(cmdgen_handle* h) {
    EvalLaunch a; a.lay = h->lay; a.w = h->work; a.d = h->dims; a.sw = h->small; a.layers = h->layers.data();
    a.edge_grid = h->edge_grid; a.coord_grid = h->coord_grid;
    a.prof_events = nullptr; a.ablate = 0;
    a.node_mt = h->node_mt; a.edge_mt = h->edge_mt; a.coord_mt = h->coord_mt;
    a.split = h->gemm_split ? 1 : 0;
    a.n_cus = h->n_cus;
    const bool sp256 = h->dims.H == 256 && h->gemm_split;
    if (!sp256) { if (a.edge_mt == 128) a.edge_mt = 64; if (a.coord_mt == 128) a.coord_mt = 64; }      // the 128-row kernels are split-engine, H = 256
    if (h->dims.sin && h->dims.H == 512) { if (a.edge_mt > 32) a.edge_mt = 32; if (a.coord_mt > 32) a.coord_mt = 32; }   // (64-row tiles + the 55 KB of feature columns exceed the LDS)
    a.edge_fullk = (sp256 && opt_of(h, "edge_fullk", 1) != 0) ? 1 : 0;
    a.e128_wgs = (int)opt_of(h, "e128_wgs_per_cu", 2);
    a.e128_fused = h->e128_fused;
    {   // the half engine's operands end at 65504: by default only where the radial features are bounded by a cutoff (every shipped config);
        // 2 forces it, 0 keeps the three-piece bf16 split everywhere
        const int he = (int)opt_of(h, "half_engine", 1);
        a.half_engine = he == 2 || (he == 1 && h->dims.cutoff2 >= 0.f) ? 1 : 0;
    }
    a.write_embed = opt_of(h, "write_embed", 1) != 0 ? 1 : 0;
    {   // k_node64 (kernels_node64.hip: 64-row node tiles, the A operand as producer-side bf16 planes, one workgroup per CU) against
        // k_node<H, 32> (register split, two workgroups per CU).  Per launch the 64-row kernel takes ~0.89 of a co-resident pair of
        // 32-row tiles, a 32-row tile alone on its CU ~0.62 of that 64-row tile (profiles/r03_m_node64.txt), so the choice is a matter
        // of how the tiles fill the CUs: compare the rounds each needs.  Option "node64" = 0 / 1 (/ 32: the 32-row planes tile) overrides.
        int on = 0;
        if (sp256 && h->n_cus > 0 && h->have_layout) {
            const int ncu = h->n_cus, t64 = (h->lay.N + 63) / 64, t32 = (h->lay.N + 31) / 32;
            const float cost64 = (float)((t64 + ncu - 1) / ncu);
            const int full = t32 / (2 * ncu), rem = t32 - full * 2 * ncu;
            float cost32 = 1.12f * full + (rem == 0 ? 0.f : rem <= ncu ? 0.62f : 1.12f);
            // half engine: k_node64 has a half form, the register-split 32-row tile has not - measured per launch (profiles/r05_t) 54.0 vs 44.6 us at
            // 96 pockets (177 32-row tiles, one per CU), 87.5 vs 46.1 at 160 (two per CU)
            if (a.half_engine) cost32 = 1.9f * full + (rem == 0 ? 0.f : rem <= ncu ? 1.22f : 1.9f);
            on = cost64 < cost32 && !(a.half_engine && h->node_mt == 16);
            // Round 6: on the half engine the 32-ROW plane tile (k_node32p: 193 registers, 67 KB of LDS - TWO workgroups per CU, so one's HBM phases run
            // beside the other's GEMMs) beats both the 64-row tile and the register-split 32-row tile wherever the eight-wave 16-row tile does not apply:
            // per evaluation -13 % at 80 C-alpha pockets, -10 % at 128, -1.4 % at 256, -2 % / -1.4 % at 64 / 256 full-atom pockets
            // (profiles/r06_h_node_tile_sweep.txt; the 64-row tile stays behind option node64 = 1)
            if (a.half_engine && h->node_mt != 16) on = 32;
            // ... except where the 32-row tiles need both slots of a CU and the 64-row tiles still fit one per CU (8 k < N <= 16 k rows on 256 CUs: 144 - 272
            // C-alpha pockets, the north star's 256): there the 64-row tile on EIGHT waves (k_node64e) streams the weights once per CU instead of twice and its
            // GEMM phases run at the matrix pipe's rate (k_node32p's are bound by the 64 B/clk of L1 fill: 85 B/clk asked) - per evaluation -1.4 .. -1.9 %
            // (profiles/r06_n_node64e.txt)
            if (on == 32 && t32 > ncu && t64 <= ncu) on = 8;
            // ... and with more 64-row tiles than CUs the lean 64-row tile, two workgroups per CU (k_node64d: 43 B/clk of weight fragments asked, and a
            // partner workgroup beside every phase): per launch -5 % at 288 C-alpha pockets, -8 % at 384, -15 % at 512, -7 % / -11 % at 64 / 128 full-atom pockets
            else if (on == 32 && t64 > ncu) on = 2;
            if (opt_set(h, "node64")) { const int64_t v = opt_of(h, "node64", 0); on = v == 32 ? 32 : v == 8 ? 8 : v == 2 ? 2 : v != 0; }
        }
        a.node64 = on;
        a.dead_skip = (h->dims.joint || h->dims.S != 1) ? 0 : (int)opt_of(h, "dead_skip", 2);   // (hop levels count blocks of ONE GCL)      // 2 (default): every block by hop level; 1: the last block only; 0: off
        if (!on && !a.dead_skip) a.w.need_qc = nullptr;
        a.w.hop_levels = a.dead_skip >= 2 ? h->dims.L : 1;
        if (!a.dead_skip) a.w.ehop = nullptr;           // the graph pass fills the flags only for the kernels that read them
    }
    // 16-row node tiles on the split engine too (v_mfma_f32_16x16x32_bf16; H >= 128): k_node<256,16> 35.0 -> 31.4 us at B=64 -
    // bound by the 6 B/weight stream of one workgroup per 16 rows, not by the matrix pipe (profiles/r03_b_*); option "node16_split" = 0 opts out
    a.split16 = (a.split && h->dims.H >= 128 && opt_of(h, "node16_split", 1) != 0) ? 1 : 0;
    a.node16w = opt_of(h, "node16w", 1) != 0 ? 1 : 0;
    {   // k_embed: inside a conditional chain only the phar tiles take the full path (the pocket rows come from the per-chain
        // cache), and they are few: 16-row tiles spread them over twice the CUs and halve the two projection passes of each
        // (B=256: 120 tiles of 32 rows 38.6 us -> 240 tiles of 16 rows)
        a.embed_mt = (int)opt_of(h, "embed_mt", (((double)h->lay.Nl / 16.0 <= 2.0 * h->n_cus && !h->dims.joint) ? 16 : a.node_mt));
        if (a.embed_mt != 16 && a.embed_mt != 32 && a.embed_mt != 64) a.embed_mt = a.node_mt;
    }
    // (the node kernel avoids its 64-row register-split tiles on the split engine: 87 vs 132 us at B=256, profiles/r02_o_tile_sweep_split.txt)
    if (a.split && a.node_mt == 64 && !opt_set(h, "node_mt")) a.node_mt = 32;
    return a;
}

// ---------------------------------------------------------------------------------
// options
// ---------------------------------------------------------------------------------
static const char* const kOptionKeys[] = {
    "node_mt", "edge_mt", "coord_mt", "embed_mt", "edge_wgs_per_cu", "coord_wgs_per_cu", "e128_wgs_per_cu", "e128_fused", "half_engine", "edge_fullk", "node64", "node16_split", "node16w",
    "dead_skip", "write_embed", "fused_step", "pocket_cache", "graph_steps",
    "wgrad_split", "wgrad_tile", "wgrad_split_wgs128", "wgrad_split_wgs64", "wgrad_wgs", "dgrad_mt", "dgrad_tail", "wgrad_stream", "train_half", "wgrad_silu", "train_node16", "wgrad_k128"};

static void drop_graphs(cmdgen_handle* h) {
    // captured graphs bake the kernel choice in; a replay of the graph destroyed here may still be running
    hipSetDevice(h->device);
    if (h->own_stream) hipStreamSynchronize(h->own_stream);
    if (h->have_layout) hipStreamSynchronize(h->last_stream);
    if (h->step_graph) { hipGraphExecDestroy(h->step_graph); h->step_graph = nullptr; }
    if (h->joint_graph) { hipGraphExecDestroy(h->joint_graph); h->joint_graph = nullptr; }
}
static void refresh_tune(cmdgen_handle* h) {
    TrainTune t;
    t.wgrad_split = (int)opt_of(h, "wgrad_split", t.wgrad_split); t.wgrad_tile = (int)opt_of(h, "wgrad_tile", t.wgrad_tile);
    t.wgrad_split_wgs128 = (int)opt_of(h, "wgrad_split_wgs128", t.wgrad_split_wgs128); t.wgrad_split_wgs64 = (int)opt_of(h, "wgrad_split_wgs64", t.wgrad_split_wgs64);
    t.wgrad_wgs = (int)opt_of(h, "wgrad_wgs", t.wgrad_wgs); t.dgrad_mt = (int)opt_of(h, "dgrad_mt", t.dgrad_mt); t.dgrad_tail = (int)opt_of(h, "dgrad_tail", t.dgrad_tail);
    t.wgrad_stream = (int)opt_of(h, "wgrad_stream", t.wgrad_stream); t.wgrad_k128 = (int)opt_of(h, "wgrad_k128", t.wgrad_k128);
    h->tune = t;
}

extern "C" int cmdgen_set_option(cmdgen_handle* h, const char* key, int64_t value, int32_t unset) {
    if (!h || !key) return fail(h, CMDGEN_EINVAL, "null argument");
    bool known = false;
    for (const char* k : kOptionKeys) known = known || strcmp(k, key) == 0;
    if (!known) return fail(h, CMDGEN_EINVAL, "unknown option '%s'", key);
    drop_graphs(h);
    if (unset) h->opts.erase(key); else h->opts[key] = value;
    refresh_tune(h);
    if (h->have_layout) pick_tiles(h);
    return CMDGEN_OK;
}

extern "C" int cmdgen_get_option(cmdgen_handle* h, const char* key, int64_t* value, int32_t* is_set) {
    if (!h || !key) return fail(h, CMDGEN_EINVAL, "null argument");
    bool known = false;
    for (const char* k : kOptionKeys) known = known || strcmp(k, key) == 0;
    if (!known) return fail(h, CMDGEN_EINVAL, "unknown option '%s'", key);
    if (value) *value = opt_of(h, key, 0);
    if (is_set) *is_set = opt_set(h, key) ? 1 : 0;
    return CMDGEN_OK;
}

// ---------------------------------------------------------------------------------
// one evaluation
// ---------------------------------------------------------------------------------
extern "C" int cmdgen_dynamics_forward(cmdgen_handle* h, const float* xh_phar, const float* xh_pocket,
                                       const float* t, float* eps_phar, float* eps_pocket, cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (!xh_phar || !xh_pocket || !t || !eps_phar) return fail(h, CMDGEN_EINVAL, "null device pointer");
    if (h->dims.joint && !eps_pocket) return fail(h, CMDGEN_EINVAL, "joint mode (update_pocket_coords) needs eps_pocket: the pocket velocity is part of the output");
    hipStream_t s = (hipStream_t)stream;
    rc = begin_work(h, s); if (rc) return rc;
    EvalLaunch a = make_launch(h);
    cmdgen_launch_eval(a, xh_phar, xh_pocket, t, nullptr, nullptr, eps_phar, eps_pocket, s, nullptr);
    if (!h->dims.joint) cmdgen_launch_nan_fix(a, eps_phar, s);     // joint: k_vel_com applied the reset already
    HIPCHK(h, hipGetLastError());
    return CMDGEN_OK;
}

extern "C" int cmdgen_debug_eval_prefix(cmdgen_handle* h, const float* xh_phar, const float* xh_pocket, const float* t,
                                        int32_t block, int32_t stage, cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (!xh_phar || !xh_pocket || !t) return fail(h, CMDGEN_EINVAL, "null device pointer");
    if (block < 0 || block >= h->dims.L || stage < 1 || stage > 3) return fail(h, CMDGEN_EINVAL, "block in [0, n_layers), stage in 1..3");
    hipStream_t s = (hipStream_t)stream;
    rc = begin_work(h, s); if (rc) return rc;
    EvalLaunch a = make_launch(h);
    a.stop_block = block; a.stop_stage = stage;
    cmdgen_launch_eval(a, xh_phar, xh_pocket, t, nullptr, nullptr, h->work.eps_tmp, nullptr, s, nullptr);
    h->agg_dirty = true;                               // stage 1 leaves the segment sums in agg
    HIPCHK(h, hipGetLastError());
    return CMDGEN_OK;
}

extern "C" int cmdgen_radius_graph(cmdgen_handle* h, const float* x, const int64_t* counts_host, int64_t batch,
                                   int32_t* row_dev, int32_t* col_dev, int64_t cap, int64_t* n_edges, cmdgen_stream stream) {
    if (!h || !x || !counts_host || batch < 1 || !row_dev || !col_dev || !n_edges) return fail(h, CMDGEN_EINVAL, "bad radius-graph arguments");
    hipSetDevice(h->device);
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)batch;
    std::vector<int> cnt(B), base(B), zeros(B, 0);
    int64_t N = 0, need = 0; int max_n = 0;
    for (int b = 0; b < B; ++b) {
        if (counts_host[b] < 0) return fail(h, CMDGEN_EINVAL, "negative node count");
        cnt[b] = (int)counts_host[b]; base[b] = (int)N; N += counts_host[b]; need += counts_host[b] * counts_host[b];
        if (cnt[b] > max_n) max_n = cnt[b];
    }
    if (N < 1 || need > (int64_t)2000000000) return fail(h, CMDGEN_EINVAL, "bad total node count / dense bound");
    if (cap < need) return fail(h, CMDGEN_EINVAL, "edge capacity %lld below the dense bound %lld", (long long)cap, (long long)need);
    if (edge_lds_bytes(max_n) > kEdgeLdsMax) return fail(h, CMDGEN_EINVAL, "a sample has %d nodes (at most %d)", max_n, (int)(kEdgeLdsMax / 28));
    if (edge_lds_bytes(max_n) > 64 * 1024) cmdgen_edge_kernels_allow_lds(edge_lds_bytes(max_n));
    // every node is presented to the radius-graph kernels as a (non-moving) pocket node of a scratch layout
    std::vector<void*> pool; void* p; int rc = 0;
    Layout L{}; Work w{};
    Dims d = h->dims; d.R = 0; d.joint = 0; d.L = 0;                         // rows of x are [3] wide
#define TMP(dst, type, count) do { if (!rc) { rc = dev_alloc(h, pool, &p, (size_t)(count) * sizeof(type), true); dst = (type*)p; } } while (0)
    int *d_np, *d_zero, *d_base;
    TMP(d_np, int, B); TMP(d_zero, int, B); TMP(d_base, int, B);
    TMP(w.X0, float4, 1); TMP(w.ACC, float4, 1); TMP(w.XP, float4, N); TMP(w.degL, int, N);
    TMP(w.pocketE, int, B); TMP(w.pocketEph, int, B); TMP(w.pocketEns, int, B); TMP(w.pocketEnsQ, int, B);
    TMP(w.ed0, float, need); TMP(w.totals, int, 4); TMP(w.counters, unsigned long long, 8); TMP(w.nan_flag, int, 4);
#undef TMP
    if (rc) { free_pool(pool); return rc; }
    w.erow = row_dev; w.ecol = col_dev;
    hipMemcpy(d_np, cnt.data(), B * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(d_base, base.data(), B * sizeof(int), hipMemcpyHostToDevice);
    L.B = B; L.Nl = 0; L.Np = (int)N; L.N = (int)N; L.Nm = 0; L.max_n = max_n;
    L.num_phar = d_zero; L.num_pocket = d_np; L.phar_base = d_zero; L.pocket_base = d_base;
    EvalLaunch a{}; a.lay = L; a.w = w; a.d = d;
    cmdgen_launch_edges(a, x /* no phar rows are read */, x, s);
    hipError_t e = hipStreamSynchronize(s);
    int tot[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(tot, w.totals, sizeof tot, hipMemcpyDeviceToHost);
    free_pool(pool);
    if (e != hipSuccess) return fail(h, CMDGEN_EHIP, "radius graph failed: %s", hipGetErrorString(e));
    *n_edges = tot[0];
    return CMDGEN_OK;
}

extern "C" int cmdgen_get_edges(cmdgen_handle* h, int32_t* row, int32_t* col, int64_t cap, int64_t* n_edges, cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    hipSetDevice(h->device);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(h, hipStreamSynchronize(s));
    int tot[2];
    HIPCHK(h, hipMemcpy(tot, h->work.totals, sizeof tot, hipMemcpyDeviceToHost));
    if (n_edges) *n_edges = tot[0];
    const int64_t n = tot[0] < cap ? tot[0] : cap;
    if (n > 0 && row) HIPCHK(h, hipMemcpy(row, h->work.erow, n * sizeof(int), hipMemcpyDeviceToHost));
    if (n > 0 && col) HIPCHK(h, hipMemcpy(col, h->work.ecol, n * sizeof(int), hipMemcpyDeviceToHost));
    return CMDGEN_OK;
}

extern "C" int cmdgen_debug_read(cmdgen_handle* h, const char* what, float* host, size_t n, cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    const void* src = nullptr; size_t have = 0;
    const std::string k = what ? what : "";
    const Dims& d = h->dims;
    if (k == "h") { src = h->work.h; have = (size_t)h->lay.N * d.H; }
    else if (k == "agg") { src = h->work.agg; have = (size_t)h->lay.N * d.H; }
    else if (k == "P") { src = h->work.P; have = (size_t)h->lay.N * d.H; }
    else if (k == "Q") { src = h->work.Q; have = (size_t)h->lay.N * d.H; }
    else if (k == "x0") { src = h->work.X0; have = (size_t)h->lay.Nm * 4; }
    else if (k == "xl") { src = h->work.XL; have = (size_t)d.L * h->lay.Nm * 4; }
    else if (k == "acc") { src = h->work.ACC; have = (size_t)d.L * h->lay.Nm * 4; }
    else return fail(h, CMDGEN_EINVAL, "unknown debug buffer '%s'", k.c_str());
    if (n > have) n = have;
    HIPCHK(h, hipMemcpy(host, src, n * sizeof(float), hipMemcpyDeviceToHost));
    return CMDGEN_OK;
}

// ---------------------------------------------------------------------------------
// schedule: per-step scalars of sample_p_zs_given_zt, fp32 in the reference's op order
// (en_diffusion.py:79-103, :859-867; conditional_model.py:345-366, :429-433)
// ---------------------------------------------------------------------------------
static inline float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }          // F.softplus, threshold 20
static inline float logsigmoid_f(float x) { return -softplus_f(-x); }
static inline float sigmoid_h(float x) { return 1.0f / (1.0f + expf(-x)); }

static void build_step_table(const std::vector<float>& gamma, int T, int K, std::vector<float>& coef) {
    coef.assign((size_t)(K + 1) * 4, 0.f);
    for (int i = 0; i < K; ++i) {
        const int s = K - 1 - i;
        const float s_arr = (float)s / (float)K, t_arr = (float)(s + 1) / (float)K;
        const float g_s = gamma[(size_t)lrintf(s_arr * (float)T)], g_t = gamma[(size_t)lrintf(t_arr * (float)T)];
        const float sigma2_ts = -expm1f(softplus_f(g_s) - softplus_f(g_t));
        const float alpha_ts = expf(0.5f * (logsigmoid_f(-g_t) - logsigmoid_f(-g_s)));
        const float sigma_ts = sqrtf(sigma2_ts);
        const float sigma_s = sqrtf(sigmoid_h(g_s)), sigma_t = sqrtf(sigmoid_h(g_t));
        coef[i * 4 + 0] = alpha_ts;
        coef[i * 4 + 1] = sigma2_ts / alpha_ts / sigma_t;
        coef[i * 4 + 2] = sigma_ts * sigma_s / sigma_t;
        coef[i * 4 + 3] = t_arr;
    }
    const float g0 = gamma[0];
    coef[K * 4 + 0] = sqrtf(sigmoid_h(g0));          // sigma_0
    coef[K * 4 + 1] = sqrtf(sigmoid_h(-g0));         // alpha_0
    coef[K * 4 + 2] = expf(0.5f * g0);               // SNR(-gamma_0/2)
    coef[K * 4 + 3] = 0.f;                           // t of the final evaluation
}

// Optional: the host supplies the table ([K+1][4], same meaning) computed with its own fp32
// math so that it matches the reference's torch ops bit for bit.
extern "C" int cmdgen_set_step_table(cmdgen_handle* h, int32_t K, const float* coef_host) {
    if (!h || K < 1 || !coef_host) return fail(h, CMDGEN_EINVAL, "bad step table");
    const size_t n = (size_t)(K + 1) * 4;
    // the Python API hands the table over on every sampling call: an unchanged table keeps the chain buffers and
    // the captured step graph (re-capture + re-instantiate cost more than a short chain)
    if (h->user_coef_K == K && h->user_coef.size() == n && memcmp(h->user_coef.data(), coef_host, n * sizeof(float)) == 0)
        return CMDGEN_OK;
    h->user_coef.assign(coef_host, coef_host + n);
    h->user_coef_K = K;
    h->chain_K = -1;
    return CMDGEN_OK;
}

static int prepare_chain(cmdgen_handle* h, int K, bool want_steps) {
    if (h->chain_K == K) return 0;
    hipDeviceSynchronize();
    if (h->step_graph) { hipGraphExecDestroy(h->step_graph); h->step_graph = nullptr; }
    free_pool(h->chain_allocs);
    const Dims& d = h->dims;
    void* p; int rc;
    std::vector<float> coef;
    if (h->user_coef_K == K) coef = h->user_coef; else build_step_table(h->gamma, h->cfg.timesteps, K, coef);
    rc = dev_alloc(h, h->chain_allocs, &p, coef.size() * sizeof(float), false); if (rc) return rc;
    HIPCHK(h, hipMemcpy(p, coef.data(), coef.size() * sizeof(float), hipMemcpyHostToDevice));
    h->chain.coef = (const float4*)p;
    rc = dev_alloc(h, h->chain_allocs, &p, (size_t)h->lay.Nl * (3 + d.P) * sizeof(float), true); if (rc) return rc; h->chain.z_phar = (float*)p;
    rc = dev_alloc(h, h->chain_allocs, &p, (size_t)h->lay.Np * (3 + d.R) * sizeof(float), true); if (rc) return rc; h->chain.xh_pocket = (float*)p;
    rc = dev_alloc(h, h->chain_allocs, &p, (size_t)(K + 3) * 2 * sizeof(unsigned int), true); if (rc) return rc; h->chain.check = (unsigned int*)p;
    rc = dev_alloc(h, h->chain_allocs, &p, sizeof(ChainState), true); if (rc) return rc; h->chain.state = (ChainState*)p;
    rc = dev_alloc(h, h->chain_allocs, &p, 4 * sizeof(unsigned int), true); if (rc) return rc; h->d_cog = (unsigned int*)p;
    {   // storage of the chain-invariant pocket rows of k_embed (PocketCache) and the two pinned time arrays used to build it
        const size_t nq = (size_t)h->lay.Np * d.H * sizeof(float), nh = (size_t)d.H * sizeof(float);
        rc = dev_alloc(h, h->chain_allocs, &p, nq, true); if (rc) return rc; h->pk_c = (float*)p;
        rc = dev_alloc(h, h->chain_allocs, &p, nq, true); if (rc) return rc; h->pk_P0 = (float*)p;
        rc = dev_alloc(h, h->chain_allocs, &p, nq, true); if (rc) return rc; h->pk_Q0 = (float*)p;
        rc = dev_alloc(h, h->chain_allocs, &p, nh, true); if (rc) return rc; h->pk_dh = (float*)p;
        rc = dev_alloc(h, h->chain_allocs, &p, nh, true); if (rc) return rc; h->pk_dP = (float*)p;
        rc = dev_alloc(h, h->chain_allocs, &p, nh, true); if (rc) return rc; h->pk_dQ = (float*)p;
        std::vector<float> t01((size_t)2 * h->lay.B, 0.f);
        for (int b = 0; b < h->lay.B; ++b) t01[h->lay.B + b] = 1.f;
        rc = dev_alloc(h, h->chain_allocs, &p, t01.size() * sizeof(float), false); if (rc) return rc; h->pk_t01 = (float*)p;
        HIPCHK(h, hipMemcpy(p, t01.data(), t01.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    h->chain_K = K;
    (void)want_steps;
    return 0;
}

extern "C" int cmdgen_sample_chain(cmdgen_handle* h, const float* pocket_x, const float* pocket_onehot,
                                   int32_t timesteps, const float* noise, uint64_t seed,
                                   const int64_t* pocket_ids_host, float* xh_phar_out, float* xh_pocket_out,
                                   float* z_steps_out, float* pocket_steps_out, int32_t use_graph, cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (!pocket_x || !pocket_onehot || !xh_phar_out || !xh_pocket_out) return fail(h, CMDGEN_EINVAL, "null device pointer");
    const int K = timesteps;
    if (K < 1 || K > h->cfg.timesteps) return fail(h, CMDGEN_EINVAL, "timesteps=%d must be in [1, %d]", K, h->cfg.timesteps);
    if (h->dims.joint) return fail(h, CMDGEN_ESTATE, "this handle is the joint model (update_pocket_coords=1): use cmdgen_joint_chain");
    h->last_chain_joint = false;
    hipSetDevice(h->device);
    hipStream_t caller = (hipStream_t)stream;
    hipStream_t s = caller;
    if (use_graph && caller == nullptr) {
        // the legacy default stream cannot be captured: run the chain on a stream of our own,
        // ordered after the caller's pending work and before its later work by events
        if (!h->own_stream) {
            HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
        }
        HIPCHK(h, hipEventRecord(h->ev_in, caller));
        HIPCHK(h, hipStreamWaitEvent(h->own_stream, h->ev_in, 0));
        s = h->own_stream;
    }
    rc = prepare_chain(h, K, z_steps_out != nullptr); if (rc) return rc;
    rc = begin_work(h, s); if (rc) return rc;
    h->last_stream = caller;
    const Dims& d = h->dims;
    // global pocket ids for the Philox key
    {
        std::vector<int64_t> gid(h->lay.B);
        for (int b = 0; b < h->lay.B; ++b) gid[b] = pocket_ids_host ? pocket_ids_host[b] : b;
        HIPCHK(h, hipMemcpyAsync(h->d_gid, gid.data(), gid.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
        HIPCHK(h, hipStreamSynchronize(s));          // gid is a stack vector
    }
    ChainBuf c = h->chain;
    c.noise = noise; c.seed = seed; c.z_steps = z_steps_out; c.pocket_steps = pocket_steps_out;
    const ChainState st0{0, K, 0, 0};
    HIPCHK(h, hipMemcpyAsync(c.state, &st0, sizeof st0, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemsetAsync(c.check, 0, (size_t)(K + 3) * 2 * sizeof(unsigned int), s));
    HIPCHK(h, hipMemsetAsync(h->d_cog, 0, 4 * sizeof(unsigned int), s));
    HIPCHK(h, hipStreamSynchronize(s));              // st0 is on the stack
    EvalLaunch a = make_launch(h);
    if (h->kernel_profiling && !use_graph) a.prof_events = h->prof_events;
    cmdgen_launch_chain_init(h->lay, d, c, pocket_x, pocket_onehot, s);
    // One denoising step = the posterior update fused with pass 1 of the next radius graph (k_step_count), then the
    // evaluation at the new state: pass 2 of the graph (k_edge_write), k_embed, the L blocks,
    // k_readout.  The chain is: evaluation 0, K x (step + evaluation), decode.  Option "fused_step" = 0 restores the
    // separate k_ddpm_step / k_edge_count launches on one stream (A/B measurements).
    const bool fused = opt_of(h, "fused_step", 1) != 0;
    if (opt_of(h, "pocket_cache", 1) != 0 && h->lay.Np > 0) {
        // chain-invariant work once per chain: the pocket's features are fixed, so k_embed's output for pocket rows is
        // affine in the time feature - two embed-only passes (t = 0, t = 1) give the cache every later evaluation reads
        cmdgen_build_pocket_cache(a, c.z_phar, c.xh_pocket, h->pk_t01, h->pk_c, h->pk_P0, h->pk_Q0, h->pk_dh, h->pk_dP, h->pk_dQ, s);
        a.pcache = PocketCache{h->pk_c, h->pk_P0, h->pk_Q0, h->pk_dh, h->pk_dP, h->pk_dQ};
    }
    EvalLaunch a2 = a;
    if (fused) { a2 = a; a2.skip_count = 1; }         // (pass 2 of the graph on a side stream was measured and dropped: the fork / join costs ~23 us per
                                                      // step inside the replayed graph, far more than the 10 us it hides; profiles/r02_b_step_fusion.txt)
    cmdgen_launch_eval(a, c.z_phar, c.xh_pocket, nullptr, c.coef, c.state, h->work.eps_tmp, nullptr, s, nullptr);   // evaluation 0 (t = 1)
    auto one_step = [&](hipStream_t ss) {
        if (fused) cmdgen_launch_step_count(h->lay, d, c, h->work, h->work.eps_tmp, ss);
        else cmdgen_launch_ddpm_step(h->lay, d, c, h->work, h->work.eps_tmp, ss);
        cmdgen_launch_eval(a2, c.z_phar, c.xh_pocket, nullptr, c.coef, c.state, h->work.eps_tmp, nullptr, ss, nullptr);
    };
    if (use_graph) {
        // The step is identical every iteration (the step index lives on the device), so it is
        // captured once per (layout, K, noise/z_steps pointers, stream) and replayed K times.
        if (h->step_graph && (h->graph_noise != noise || h->graph_zsteps != z_steps_out || h->graph_psteps != pocket_steps_out || h->graph_stream != s || h->graph_seed != seed)) {
            hipGraphExecDestroy(h->step_graph); h->step_graph = nullptr;
        }
        // G identical steps per graph launch amortise the per-replay floor (~10-16 us host side, a few us of
        // device idle): the step index lives on the device, so a G-step graph is just G copies of the step.
        int G = (int)opt_of(h, "graph_steps", 8);
        if (G < 1) G = 1;
        if (G > K) G = K;
        if (h->step_graph && h->graph_steps != G) { hipGraphExecDestroy(h->step_graph); h->step_graph = nullptr; }
        if (!h->step_graph) {
            hipGraph_t g = nullptr;
            HIPCHK(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < G; ++i) one_step(s);
            HIPCHK(h, hipStreamEndCapture(s, &g));
            HIPCHK(h, hipGraphInstantiate(&h->step_graph, g, nullptr, nullptr, 0));
            hipGraphDestroy(g);
            h->graph_noise = noise; h->graph_zsteps = z_steps_out; h->graph_psteps = pocket_steps_out; h->graph_stream = s; h->graph_seed = seed;
            h->graph_steps = G;
        }
        for (int i = 0; i < K / G; ++i) HIPCHK(h, hipGraphLaunch(h->step_graph, s));
        for (int i = 0; i < K % G; ++i) one_step(s);
    } else {
        for (int i = 0; i < K; ++i) one_step(s);
    }
    // final p(x, h | z0): the last evaluation above ran at t = 0 (coef[K].w); decode
    cmdgen_launch_chain_final(h->lay, d, c, h->work, h->work.eps_tmp, xh_phar_out, xh_pocket_out, h->d_cog, s);
    HIPCHK(h, hipGetLastError());
    if (s != caller) {
        HIPCHK(h, hipEventRecord(h->ev_out, s));
        HIPCHK(h, hipStreamWaitEvent(caller, h->ev_out, 0));
    }
    return CMDGEN_OK;
}


// ---------------------------------------------------------------------------------
// joint model: EnVariationalDiffusion.sample / .inpaint (en_diffusion.py:576-831)
// ---------------------------------------------------------------------------------
// get_repaint_schedule (en_diffusion.py:649-670): denoising steps to run before each jump back
static std::vector<int> repaint_schedule(int resamplings, int jump_length, int timesteps) {
    std::vector<int> sched;
    int curr_t = 0;
    while (curr_t < timesteps) {
        if (curr_t + jump_length < timesteps) {
            if (!sched.empty()) {
                sched.back() += jump_length;
                for (int i = 0; i < resamplings - 1; ++i) sched.push_back(jump_length);
            } else {
                for (int i = 0; i < resamplings; ++i) sched.push_back(jump_length);
            }
            curr_t += jump_length;
        } else {
            const int residual = timesteps - curr_t;
            if (!sched.empty()) sched.back() += residual; else sched.push_back(residual);
            curr_t += residual;
        }
    }
    return std::vector<int>(sched.rbegin(), sched.rend());
}

struct JointPlan {
    std::vector<float> coef, coef2;     // [n_steps+1][4], [n_steps+1][4]
    std::vector<int> iop;               // [n_steps+1][4]
    int n_steps = 0, n_draws = 0;
};

// The op table of one chain: one row per network evaluation, in execution order (the walk of :723-813).
static JointPlan build_joint_plan(const std::vector<float>& gamma, int T, int K, int resamplings, int jump, bool inpaint) {
    JointPlan p;
    const std::vector<int> sched = inpaint ? repaint_schedule(resamplings, jump, K) : std::vector<int>{K};
    auto g_at = [&](int step) { return gamma[(size_t)lrintf(((float)step / (float)K) * (float)T)]; };
    int draw = 1;                       // draw 0 = z_T
    int s = K - 1;
    for (size_t i = 0; i < sched.size(); ++i) {
        for (int j = 0; j < sched[i]; ++j) {
            const float g_s = g_at(s), g_t = g_at(s + 1);
            const float sigma2_ts = -expm1f(softplus_f(g_s) - softplus_f(g_t));
            const float alpha_ts = expf(0.5f * (logsigmoid_f(-g_t) - logsigmoid_f(-g_s)));
            const float sigma_ts = sqrtf(sigma2_ts);
            const float sigma_s = sqrtf(sigmoid_h(g_s)), sigma_t = sqrtf(sigmoid_h(g_t));
            p.coef.insert(p.coef.end(), {alpha_ts, sigma2_ts / alpha_ts / sigma_t, sigma_ts * sigma_s / sigma_t,
                                         (float)(s + 1) / (float)K});
            float re_a = 0.f, re_s = 0.f; int flags = 0;
            const int draw0 = draw;
            draw += inpaint ? 2 : 1;
            if (j == sched[i] - 1 && i + 1 < sched.size()) {      // jump back s -> s + jump_length
                const float g_t2 = g_at(s + jump);
                re_s = sqrtf(-expm1f(softplus_f(g_s) - softplus_f(g_t2)));
                re_a = expf(0.5f * (logsigmoid_f(-g_t2) - logsigmoid_f(-g_s)));
                flags = 1; draw += 1;
                s = s + jump;
            }
            p.coef2.insert(p.coef2.end(), {sqrtf(sigmoid_h(-g_s)), sigma_s, re_a, re_s});
            p.iop.insert(p.iop.end(), {flags, draw0, 0, 0});
            s -= 1;
            p.n_steps += 1;
        }
    }
    const float g0 = gamma[0];
    p.coef.insert(p.coef.end(), {sqrtf(sigmoid_h(g0)), sqrtf(sigmoid_h(-g0)), expf(0.5f * g0), 0.f});
    p.coef2.insert(p.coef2.end(), {0.f, 0.f, 0.f, 0.f});
    p.iop.insert(p.iop.end(), {0, draw, 0, 0});
    p.n_draws = draw + 1;
    return p;
}

static int check_joint_args(cmdgen_handle* h, int K, int resamplings, int jump) {
    if (!h->dims.joint) return fail(h, CMDGEN_ESTATE, "joint chains need a handle created with update_pocket_coords=1");
    if (K < 1 || K > h->cfg.timesteps) return fail(h, CMDGEN_EINVAL, "timesteps=%d must be in [1, %d]", K, h->cfg.timesteps);
    if (resamplings < 1 || jump < 1) return fail(h, CMDGEN_EINVAL, "resamplings and jump_length must be >= 1");
    return 0;
}

extern "C" int cmdgen_joint_plan(cmdgen_handle* h, int32_t timesteps, int32_t resamplings, int32_t jump_length,
                                 int32_t inpaint, int64_t* n_steps, int64_t* n_draws) {
    if (!h) return CMDGEN_EINVAL;
    if (!h->finalized) return fail(h, CMDGEN_ESTATE, "weights not finalised (cmdgen_finalize_weights)");
    int rc = check_joint_args(h, timesteps, resamplings, jump_length); if (rc) return rc;
    const JointPlan p = build_joint_plan(h->gamma, h->cfg.timesteps, timesteps, resamplings, jump_length, inpaint != 0);
    if (n_steps) *n_steps = p.n_steps;
    if (n_draws) *n_draws = p.n_draws;
    return CMDGEN_OK;
}

static int prepare_joint(cmdgen_handle* h, int K, int resamplings, int jump, bool inpaint) {
    const std::vector<int> key{K, resamplings, jump, inpaint ? 1 : 0};
    if (h->joint_steps >= 0 && h->joint_key == key) return 0;
    hipDeviceSynchronize();
    if (h->joint_graph) { hipGraphExecDestroy(h->joint_graph); h->joint_graph = nullptr; }
    free_pool(h->joint_allocs);
    h->joint_steps = -1;
    const Dims& d = h->dims;
    const JointPlan p = build_joint_plan(h->gamma, h->cfg.timesteps, K, resamplings, jump, inpaint);
    void* q; int rc;
    auto up = [&](const void* src, size_t bytes, const void** dst) -> int {
        int r = dev_alloc(h, h->joint_allocs, &q, bytes, false); if (r) return r;
        if (hipMemcpy(q, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(h, CMDGEN_EHIP, "hipMemcpy H2D failed");
        *dst = q; return 0;
    };
    const void* dp;
    rc = up(p.coef.data(), p.coef.size() * sizeof(float), &dp); if (rc) return rc; h->joint.coef = (const float4*)dp;
    rc = up(p.coef2.data(), p.coef2.size() * sizeof(float), &dp); if (rc) return rc; h->joint.coef2 = (const float4*)dp;
    rc = up(p.iop.data(), p.iop.size() * sizeof(int), &dp); if (rc) return rc; h->joint.iop = (const int4*)dp;
    const size_t np_ = (size_t)h->lay.Nl * (3 + d.P) * sizeof(float), nq_ = (size_t)h->lay.Np * (3 + d.R) * sizeof(float);
#define JALLOC(dst, bytes) do { rc = dev_alloc(h, h->joint_allocs, &q, bytes, true); if (rc) return rc; dst = (float*)q; } while (0)
    JALLOC(h->joint.z_phar, np_); JALLOC(h->joint.z_pocket, nq_);
    JALLOC(h->joint.e_phar, np_); JALLOC(h->joint.e_pocket, nq_);
    JALLOC(h->joint.zk_phar, np_); JALLOC(h->joint.zk_pocket, nq_);
    JALLOC(h->joint.x0_phar, np_); JALLOC(h->joint.x0_pocket, nq_);
    JALLOC(h->eps_pocket_tmp, nq_);
#undef JALLOC
    rc = dev_alloc(h, h->joint_allocs, &q, (size_t)(p.n_steps + 3) * 2 * sizeof(unsigned int), true); if (rc) return rc; h->joint.check = (unsigned int*)q;
    rc = dev_alloc(h, h->joint_allocs, &q, sizeof(ChainState), true); if (rc) return rc; h->joint.state = (ChainState*)q;
    rc = dev_alloc(h, h->joint_allocs, &q, 4 * sizeof(unsigned int), true); if (rc) return rc; h->joint_cog = (unsigned int*)q;
    h->joint_steps = p.n_steps;
    h->joint_key = key;
    return 0;
}

extern "C" int cmdgen_joint_chain(cmdgen_handle* h, const float* phar_x, const float* phar_onehot,
                                  const float* pocket_x, const float* pocket_onehot,
                                  const float* phar_fixed, const float* pocket_fixed,
                                  int32_t timesteps, int32_t resamplings, int32_t jump_length,
                                  const float* noise, int64_t n_draws, uint64_t seed, const int64_t* pocket_ids_host,
                                  float* xh_phar_out, float* xh_pocket_out, float* z_steps_out,
                                  int32_t use_graph, cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    rc = check_joint_args(h, timesteps, resamplings, jump_length); if (rc) return rc;
    if (!xh_phar_out || !xh_pocket_out) return fail(h, CMDGEN_EINVAL, "null output pointer");
    const bool inpaint = phar_fixed != nullptr || pocket_fixed != nullptr;
    if (inpaint && (!phar_fixed || !pocket_fixed || !phar_x || !phar_onehot || !pocket_x || !pocket_onehot))
        return fail(h, CMDGEN_EINVAL, "inpainting needs phar_x, phar_onehot, pocket_x, pocket_onehot and both fixed masks");
    hipSetDevice(h->device);
    hipStream_t caller = (hipStream_t)stream;
    hipStream_t s = caller;
    if (use_graph && caller == nullptr) {
        if (!h->own_stream) {
            HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
        }
        HIPCHK(h, hipEventRecord(h->ev_in, caller));
        HIPCHK(h, hipStreamWaitEvent(h->own_stream, h->ev_in, 0));
        s = h->own_stream;
    }
    rc = prepare_joint(h, timesteps, resamplings, jump_length, inpaint); if (rc) return rc;
    rc = begin_work(h, s); if (rc) return rc;
    h->last_stream = caller;
    const int n_steps = h->joint_steps;
    if (noise) {
        int64_t need = 0;
        cmdgen_joint_plan(h, timesteps, resamplings, jump_length, inpaint ? 1 : 0, nullptr, &need);
        if (n_draws < need) return fail(h, CMDGEN_EINVAL, "noise holds %lld combined draws, the schedule needs %lld", (long long)n_draws, (long long)need);
    }
    h->last_chain_joint = true;
    const Dims& d = h->dims;
    {
        std::vector<int64_t> gid(h->lay.B);
        for (int b = 0; b < h->lay.B; ++b) gid[b] = pocket_ids_host ? pocket_ids_host[b] : b;
        HIPCHK(h, hipMemcpyAsync(h->d_gid, gid.data(), gid.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
        HIPCHK(h, hipStreamSynchronize(s));
    }
    JointBuf c = h->joint;
    c.fix_phar = inpaint ? phar_fixed : nullptr; c.fix_pocket = inpaint ? pocket_fixed : nullptr;
    c.noise = noise; c.seed = seed; c.z_steps = z_steps_out;
    const ChainState st0{0, n_steps, 0, 0};
    HIPCHK(h, hipMemcpyAsync(c.state, &st0, sizeof st0, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemsetAsync(c.check, 0, (size_t)(n_steps + 3) * 2 * sizeof(unsigned int), s));
    HIPCHK(h, hipMemsetAsync(h->joint_cog, 0, 4 * sizeof(unsigned int), s));
    HIPCHK(h, hipStreamSynchronize(s));
    EvalLaunch a = make_launch(h);
    if (h->kernel_profiling && !use_graph) a.prof_events = h->prof_events;
    cmdgen_launch_joint_init(h->lay, d, c, phar_x, phar_onehot, pocket_x, pocket_onehot, s);
    auto one_step = [&](hipStream_t ss) {
        cmdgen_launch_eval(a, c.z_phar, c.z_pocket, nullptr, c.coef, c.state, h->work.eps_tmp, h->eps_pocket_tmp, ss, nullptr);
        cmdgen_launch_joint_step(h->lay, d, c, h->work.eps_tmp, h->eps_pocket_tmp, ss);
    };
    if (use_graph) {
        // as in cmdgen_sample_chain: the op index lives on the device, so G captured steps replay for any position
        const void* key[6] = {noise, z_steps_out, (const void*)s, phar_fixed, pocket_fixed, nullptr};
        int G = (int)opt_of(h, "graph_steps", 8);
        if (G < 1) G = 1;
        if (G > n_steps) G = n_steps;
        if (h->joint_graph && (memcmp(key, h->jg_key, sizeof key) != 0 || h->jg_seed != seed || h->jg_steps != G)) {
            hipGraphExecDestroy(h->joint_graph); h->joint_graph = nullptr;
        }
        if (!h->joint_graph) {
            hipGraph_t g = nullptr;
            HIPCHK(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < G; ++i) one_step(s);
            HIPCHK(h, hipStreamEndCapture(s, &g));
            HIPCHK(h, hipGraphInstantiate(&h->joint_graph, g, nullptr, nullptr, 0));
            hipGraphDestroy(g);
            memcpy(h->jg_key, key, sizeof key); h->jg_seed = seed; h->jg_steps = G;
        }
        for (int i = 0; i < n_steps / G; ++i) HIPCHK(h, hipGraphLaunch(h->joint_graph, s));
        for (int i = 0; i < n_steps % G; ++i) one_step(s);
    } else {
        for (int i = 0; i < n_steps; ++i) one_step(s);
    }
    cmdgen_launch_eval(a, c.z_phar, c.z_pocket, nullptr, c.coef, c.state, h->work.eps_tmp, h->eps_pocket_tmp, s, nullptr);
    cmdgen_launch_joint_final(h->lay, d, c, h->work.eps_tmp, h->eps_pocket_tmp, xh_phar_out, xh_pocket_out, h->joint_cog, s);
    HIPCHK(h, hipGetLastError());
    if (s != caller) {
        HIPCHK(h, hipEventRecord(h->ev_out, s));
        HIPCHK(h, hipStreamWaitEvent(caller, h->ev_out, 0));
    }
    return CMDGEN_OK;
}

extern "C" int cmdgen_chain_status(cmdgen_handle* h, float* max_rel, float* max_cog, int64_t* nan_resets, cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (h->last_chain_joint ? h->joint_steps < 0 : h->chain_K < 0) return fail(h, CMDGEN_ESTATE, "no chain has run");
    hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    const int K = h->last_chain_joint ? h->joint_steps : h->chain_K;
    std::vector<unsigned int> chk((size_t)(K + 3) * 2);
    HIPCHK(h, hipMemcpy(chk.data(), h->last_chain_joint ? h->joint.check : h->chain.check, chk.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
    float worst = 0.f;
    for (int i = 0; i < K + 2 && worst == worst; ++i) {
        float largest, err;
        memcpy(&largest, &chk[2 * i], 4); memcpy(&err, &chk[2 * i + 1], 4);
        const float rel = err / (largest + 1e-10f);
        if (rel > worst || rel != rel) worst = rel;      // a NaN sticks: the reference's `assert rel_error < 1e-2` fails on it
    }
    if (max_rel) *max_rel = worst;
    unsigned int cog; HIPCHK(h, hipMemcpy(&cog, h->last_chain_joint ? h->joint_cog : h->d_cog, 4, hipMemcpyDeviceToHost));
    if (max_cog) memcpy(max_cog, &cog, 4);
    unsigned long long cnt[8];
    HIPCHK(h, hipMemcpy(cnt, h->work.counters, sizeof cnt, hipMemcpyDeviceToHost));
    if (nan_resets) *nan_resets = (int64_t)cnt[4];
    return CMDGEN_OK;
}

// ---------------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------------
extern "C" int cmdgen_get_counters(cmdgen_handle* h, cmdgen_counters* out, cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    unsigned long long cnt[8];
    HIPCHK(h, hipMemcpy(cnt, h->work.counters, sizeof cnt, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->evaluations = cnt[0]; out->edges = cnt[1]; out->edges_phar = cnt[2]; out->nodes = cnt[3]; out->nan_resets = cnt[4];
    out->edges_skipped = cnt[6]; out->node_rows_skipped = cnt[7];
    return CMDGEN_OK;
}

extern "C" int cmdgen_reset_counters(cmdgen_handle* h, cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    hipSetDevice(h->device);
    HIPCHK(h, hipMemsetAsync(h->work.counters, 0, 8 * sizeof(unsigned long long), (hipStream_t)stream));
    return CMDGEN_OK;
}

extern "C" int cmdgen_profile_evaluation(cmdgen_handle* h, const float* xh_phar, const float* xh_pocket, const float* t,
                                         float* eps_phar, cmdgen_kernel_times* out, cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (!out) return fail(h, CMDGEN_EINVAL, "null output");
    if (h->dims.joint) return fail(h, CMDGEN_ESTATE, "cmdgen_profile_evaluation supports the conditional model only");
    if (h->dims.S != 1) return fail(h, CMDGEN_ESTATE, "cmdgen_profile_evaluation supports inv_sublayers = 1 only (cmdgen_set_kernel_profiling works for any)");
    hipStream_t s = (hipStream_t)stream;
    rc = begin_work(h, s); if (rc) return rc;
    const int L = h->dims.L;
    const int nev = 2 * (3 + 3 * L);
    std::vector<hipEvent_t> ev(nev);
    for (auto& e : ev) HIPCHK(h, hipEventCreate(&e));
    EvalLaunch a = make_launch(h);
    cmdgen_launch_eval(a, xh_phar, xh_pocket, t, nullptr, nullptr, eps_phar, nullptr, s, ev.data());
    cmdgen_launch_nan_fix(a, eps_phar, s);
    HIPCHK(h, hipStreamSynchronize(s));
    memset(out, 0, sizeof *out);
    auto ms = [&](int i) { float m = 0.f; hipEventElapsedTime(&m, ev[2 * i], ev[2 * i + 1]); return m; };
    int i = 0;
    out->edge_build_ms = ms(i++); out->embed_ms = ms(i++);
    for (int l = 0; l < L; ++l) {
        out->edge_msg_ms += ms(i++); out->node_ms += ms(i++); out->edge_coord_ms += ms(i++);
    }
    out->readout_ms = ms(i++);
    out->edge_msg_launches = L; out->node_launches = L; out->edge_coord_launches = L;
    for (auto& e : ev) hipEventDestroy(e);
    return CMDGEN_OK;
}

extern "C" int cmdgen_set_gemm_mode(cmdgen_handle* h, int32_t split_bf16) {
    if (!h) return CMDGEN_EINVAL;
    if (split_bf16 && h->dims.sin) return fail(h, CMDGEN_ESTATE, "sin_embedding runs on the fp32 matrix instruction only (the split engine's tile builders carry two scalar edge features)");
    if (h->gemm_split != (split_bf16 != 0)) {
        drop_graphs(h);                 // captured graphs bake the kernel choice in; the pocket cache is rebuilt per chain anyway
        h->gemm_split = split_bf16 != 0;
        if (h->have_layout) pick_tiles(h);
    }
    return CMDGEN_OK;
}

// MFMAs per fp32 product of the three tile kernels of an evaluation, as the launchers pick them: 1 = the fp32 matrix instruction, 6 = three bf16
// pieces per operand, 3 = the half engine (two fp16 pieces).  which: 0 messages, 1 node, 2 coordinates.
static int mfmas_per_product(const EvalLaunch& a, int which) {
    const LayerW& lw = a.layers[0];
    const bool sampler = !a.save, h256 = a.d.H == 256;
    if (which == 1) {
        if (h256 && a.split && sampler && a.node64 && lw.W3.ws) return a.half_engine && lw.W3.wh ? 3 : 6;
        if (h256 && a.node_mt == 16 && a.split16 && a.node16w && sampler && lw.W3.ws16) return a.half_engine && lw.W3.wh16 ? 3 : 6;
        if ((a.split && a.node_mt >= 32) || (a.split16 && a.node_mt == 16)) return 6;
        return 1;
    }
    const int mt = which == 0 ? a.edge_mt : a.coord_mt;
    const WPack& W = which == 0 ? lw.W2 : lw.W7;
    if (mt == 128 && h256 && a.split && sampler && W.ws) return a.half_engine && W.wh ? 3 : 6;
    if (a.edge_fullk && sampler && a.split && h256 && mt == 32) return a.half_engine && W.wh ? 3 : 6;
    if (a.split && mt >= 32) return 6;
    return 1;
}

extern "C" int cmdgen_query(cmdgen_handle* h, const char* key, int64_t* value) {
    if (!h || !key || !value) return CMDGEN_EINVAL;
    if (!h->have_layout) return fail(h, CMDGEN_ESTATE, "no batch layout (cmdgen_set_layout)");
    const std::string k = key;
    const EvalLaunch a = make_launch(h);
    if (k == "node_mt") *value = a.node_mt;
    else if (k == "edge_mt") *value = a.edge_mt;
    else if (k == "e128_fused") *value = a.e128_fused;
    else if (k == "coord_mt") *value = a.coord_mt;
    else if (k == "edge_grid") *value = a.edge_grid;
    else if (k == "coord_grid") *value = a.coord_grid;
    else if (k == "gemm_split") *value = a.split;
    else if (k == "half_engine") *value = a.split ? a.half_engine : 0;
    else if (k == "msg_mfmas_per_product") *value = mfmas_per_product(a, 0);
    else if (k == "node_mfmas_per_product") *value = mfmas_per_product(a, 1);
    else if (k == "coord_mfmas_per_product") *value = mfmas_per_product(a, 2);
    else if (k == "node16_split") *value = a.split16;
    else if (k == "node64") *value = a.node64;
    else if (k == "node16w") *value = a.node16w;
    else if (k == "edge_fullk") *value = a.edge_fullk;
    else if (k == "dead_skip") *value = a.dead_skip;
    else if (k == "train_edges") *value = h->train_E;
    else if (k == "train_coord_edges") *value = h->train_Ec;
    else return fail(h, CMDGEN_EINVAL, "unknown query '%s'", key);
    return CMDGEN_OK;
}

extern "C" int cmdgen_time_evaluation(cmdgen_handle* h, const float* xh_phar, const float* xh_pocket, const float* t,
                                      float* eps_phar, int32_t graph_len, int32_t replays, float* mean_ms, cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (!xh_phar || !xh_pocket || !t || !eps_phar || !mean_ms || graph_len < 1 || replays < 1) return fail(h, CMDGEN_EINVAL, "bad arguments");
    if (h->dims.joint) return fail(h, CMDGEN_ESTATE, "cmdgen_time_evaluation supports the conditional model only");
    hipStream_t caller = (hipStream_t)stream, s = caller;
    if (caller == nullptr) {                             // the legacy default stream cannot be captured
        if (!h->own_stream) {
            HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
        }
        HIPCHK(h, hipDeviceSynchronize());
        s = h->own_stream;
    }
    rc = begin_work(h, s); if (rc) return rc;
    EvalLaunch a = make_launch(h);
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // every exit below releases what it holds: a failure between Begin- and EndCapture must still end the capture (the
    // caller's stream would otherwise stay in capture mode and poison every later launch on it)
    hipError_t err = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (err != hipSuccess) return fail(h, CMDGEN_EHIP, "hipStreamBeginCapture: %s", hipGetErrorString(err));
    for (int i = 0; i < graph_len; ++i) {
        cmdgen_launch_eval(a, xh_phar, xh_pocket, t, nullptr, nullptr, eps_phar, nullptr, s, nullptr);
        cmdgen_launch_nan_fix(a, eps_phar, s);
    }
    const hipError_t launch_err = hipGetLastError();
    err = hipStreamEndCapture(s, &g);                    // always: also after a failed launch
    if (err == hipSuccess && launch_err != hipSuccess) err = launch_err;
    if (err == hipSuccess) err = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (g) hipGraphDestroy(g);
    if (err == hipSuccess) err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    if (err == hipSuccess) err = hipGraphLaunch(ge, s);  // warm
    if (err == hipSuccess) err = hipEventRecord(e0, s);
    for (int i = 0; i < replays && err == hipSuccess; ++i) err = hipGraphLaunch(ge, s);
    if (err == hipSuccess) err = hipEventRecord(e1, s);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (ge) hipGraphExecDestroy(ge);
    if (err != hipSuccess) return fail(h, CMDGEN_EHIP, "cmdgen_time_evaluation: %s", hipGetErrorString(err));
    *mean_ms = ms / ((float)replays * (float)graph_len);
    return CMDGEN_OK;
}

extern "C" int cmdgen_time_edge_kernel(cmdgen_handle* h, int32_t layer, int32_t reps, float* mean_ms, cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (layer < 0 || (layer & 0xff) >= h->dims.L || reps < 1 || !mean_ms) return fail(h, CMDGEN_EINVAL, "bad arguments");
    hipSetDevice(h->device);
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    EvalLaunch a = make_launch(h);
    a.ablate = (layer >> 8) & 0xff;                                 // bits 8.. of `layer`: phase-ablation mask (timing only)
    layer &= 0xff;
    cmdgen_launch_edge_msg_only(a, layer, s);                       // warm
    HIPCHK(h, hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) cmdgen_launch_edge_msg_only(a, layer, s);
    HIPCHK(h, hipEventRecord(e1, s));
    HIPCHK(h, hipEventSynchronize(e1));
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    *mean_ms = ms / (float)reps;
    // the replays accumulated into agg; restore the invariant "agg is zero between blocks"
    HIPCHK(h, hipMemsetAsync(h->work.agg, 0, (size_t)h->lay.N * h->dims.H * sizeof(float), s));
    hipEventDestroy(e0); hipEventDestroy(e1);
    return CMDGEN_OK;
}

extern "C" int cmdgen_debug_stamps(cmdgen_handle* h, uint64_t* out64, int32_t reset) {
    if (!h || !h->have_layout || !out64) return CMDGEN_EINVAL;
    hipSetDevice(h->device);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out64, h->work.dbg, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(h, hipMemset(h->work.dbg, 0, 64 * sizeof(uint64_t)));
    return CMDGEN_OK;
}

extern "C" int cmdgen_set_kernel_profiling(cmdgen_handle* h, int32_t on) {
    if (!h) return CMDGEN_EINVAL;
    h->kernel_profiling = on != 0;
    return CMDGEN_OK;
}

extern "C" int cmdgen_get_kernel_profile(cmdgen_handle* h, float total_ms[3], int64_t launches[3], cmdgen_stream stream) {
    int rc = check_ready(h); if (rc) return rc;
    hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    if (h->own_stream) HIPCHK(h, hipStreamSynchronize(h->own_stream));
    for (int k = 0; k < 3; ++k) {
        std::vector<hipEvent_t>& ev = h->prof_events[k];
        double tot = 0.0;
        const size_t n = ev.size() / 2;
        for (size_t i = 0; i < n; ++i) {
            float ms = 0.f;
            HIPCHK(h, hipEventSynchronize(ev[2 * i + 1]));
            hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
            tot += ms;
        }
        for (hipEvent_t e : ev) hipEventDestroy(e);
        ev.clear();
        if (total_ms) total_ms[k] = (float)tot;
        if (launches) launches[k] = (int64_t)n;
    }
    return CMDGEN_OK;
}

extern "C" int cmdgen_debug_noise(cmdgen_handle* h, uint64_t seed, int64_t pocket_id, int32_t draw, int32_t n_nodes,
                                  int32_t width, float* out_dev, cmdgen_stream stream) {
    if (!h || !out_dev || n_nodes < 1 || width < 1 || width > 16) return fail(h, CMDGEN_EINVAL, "bad arguments");
    hipSetDevice(h->device);
    cmdgen_launch_debug_noise(seed, pocket_id, draw, n_nodes, width, out_dev, (hipStream_t)stream);
    HIPCHK(h, hipGetLastError());
    return CMDGEN_OK;
}
