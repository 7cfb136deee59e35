// c_api_demo.cpp - drives libcmdgen_hip.so through its C ABI only (no Python, no torch):
// random weights with the reference checkpoint's names, one batch of synthetic pockets, one
// sampling chain with on-device noise, then the deferred checks and work counters.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/c_api_demo.cpp -Lcmdgen_amd -lcmdgen_hip \
//         -Wl,-rpath,$PWD/cmdgen_amd -o /tmp/c_api_demo && /tmp/c_api_demo
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "cmdgen_hip.h"

#define CK(call) do { int _rc = (call); if (_rc != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, _rc, cmdgen_last_error(h)); return 1; } } while (0)
#define HK(call) do { hipError_t _e = (call); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(_e)); return 1; } } while (0)

int main() {
    cmdgen_config cfg{};
    cfg.phar_nf = 8; cfg.residue_nf = 20; cfg.joint_nf = 32; cfg.hidden_nf = 256; cfg.n_layers = 5; cfg.inv_sublayers = 1;
    cfg.attention = 1; cfg.tanh = 1; cfg.condition_time = 1; cfg.timesteps = 500; cfg.no_com_projection = 0; cfg.edge_cutoff = 6.0f;
    cfg.norm_constant = 1.0f; cfg.normalization_factor = 100.0f; cfg.coords_range = 15.0f;
    cfg.norm_x = 1.0f; cfg.norm_h = 4.0f; cfg.bias_h = 0.0f;
    cmdgen_handle* h = nullptr;
    if (cmdgen_create(&cfg, 0, &h) != 0) { fprintf(stderr, "cmdgen_create: %s\n", cmdgen_last_error(nullptr)); return 1; }
    printf("%s\n", cmdgen_version());

    std::mt19937 rng(0);
    auto load = [&](const std::string& name, int out, int in, bool bias, float gain = 1.0f) -> int {
        std::uniform_real_distribution<float> u(-gain / std::sqrt((float)in), gain / std::sqrt((float)in));
        std::vector<float> w((size_t)out * in);
        for (auto& v : w) v = u(rng);
        int rc = cmdgen_load_weights(h, (name + ".weight").c_str(), w.data(), w.size());
        if (rc == 0 && bias) { std::vector<float> b(out); for (auto& v : b) v = u(rng); rc = cmdgen_load_weights(h, (name + ".bias").c_str(), b.data(), b.size()); }
        return rc;
    };
    const int P = 8, R = 20, J = 32, H = 256;
    CK(load("dynamics.phar_encoder.0", 2 * P, P, true)); CK(load("dynamics.phar_encoder.2", J, 2 * P, true));
    CK(load("dynamics.phar_decoder.0", 2 * P, J, true)); CK(load("dynamics.phar_decoder.2", P, 2 * P, true));
    CK(load("dynamics.residue_encoder.0", 2 * R, R, true)); CK(load("dynamics.residue_encoder.2", J, 2 * R, true));
    CK(load("dynamics.residue_decoder.0", 2 * R, J, true)); CK(load("dynamics.residue_decoder.2", R, 2 * R, true));
    CK(load("dynamics.egnn.embedding", H, J + 1, true)); CK(load("dynamics.egnn.embedding_out", J + 1, H, true));
    for (int b = 0; b < cfg.n_layers; ++b) {
        const std::string g = "dynamics.egnn.e_block_" + std::to_string(b) + ".gcl_0.", c = "dynamics.egnn.e_block_" + std::to_string(b) + ".gcl_equiv.";
        CK(load(g + "edge_mlp.0", H, 2 * H + 2, true)); CK(load(g + "edge_mlp.2", H, H, true));
        CK(load(g + "node_mlp.0", H, 2 * H, true)); CK(load(g + "node_mlp.2", H, H, true)); CK(load(g + "att_mlp.0", 1, H, true));
        CK(load(c + "coord_mlp.0", H, 2 * H + 2, true)); CK(load(c + "coord_mlp.2", H, H, true)); CK(load(c + "coord_mlp.4", 1, H, false, 1e-3f));
    }
    {   // gamma table of the 'polynomial_2' schedule (en_diffusion.py:1135-1184), float64 then fp32
        const int T = cfg.timesteps; std::vector<double> a2(T + 2); a2[0] = 1.0;
        for (int i = 0; i <= T; ++i) { const double x = (double)i * (T + 1) / T / (T + 1); const double v = 1.0 - x * x; a2[i + 1] = v * v; }
        std::vector<float> gamma(T + 1); double cum = 1.0;
        for (int i = 0; i <= T; ++i) { double r = a2[i + 1] / a2[i]; r = std::min(1.0, std::max(0.001, r)); cum *= r; const double a = (1 - 2e-5) * cum + 1e-5; gamma[i] = (float)(-(std::log(a) - std::log(1.0 - a))); }
        CK(cmdgen_load_weights(h, "gamma.gamma", gamma.data(), gamma.size()));
    }
    CK(cmdgen_finalize_weights(h));

    const int B = 16, Np = 44, Nl = 15, K = 50;
    std::vector<int64_t> nph(B, Nl), npk(B, Np);
    CK(cmdgen_set_layout(h, B, nph.data(), npk.data()));
    std::vector<float> px((size_t)B * Np * 3), oh((size_t)B * Np * R, 0.f);
    std::normal_distribution<float> nd(0.f, 6.f);
    for (auto& v : px) v = nd(rng);
    for (int i = 0; i < B * Np; ++i) oh[(size_t)i * R + (rng() % R)] = 1.f;
    float *d_px, *d_oh, *d_xp, *d_xq;
    HK(hipMalloc(&d_px, px.size() * 4)); HK(hipMalloc(&d_oh, oh.size() * 4));
    HK(hipMalloc(&d_xp, (size_t)B * Nl * (3 + P) * 4)); HK(hipMalloc(&d_xq, (size_t)B * Np * (3 + R) * 4));
    HK(hipMemcpy(d_px, px.data(), px.size() * 4, hipMemcpyHostToDevice)); HK(hipMemcpy(d_oh, oh.data(), oh.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s; HK(hipStreamCreate(&s));
    CK(cmdgen_sample_chain(h, d_px, d_oh, K, nullptr, 42, nullptr, d_xp, d_xq, nullptr, nullptr, 1, s));
    float rel = 0, cog = 0; int64_t nans = 0;
    CK(cmdgen_chain_status(h, &rel, &cog, &nans, s));
    cmdgen_counters c; CK(cmdgen_get_counters(h, &c, s));
    std::vector<float> xp((size_t)B * Nl * (3 + P));
    HK(hipMemcpy(xp.data(), d_xp, xp.size() * 4, hipMemcpyDeviceToHost));
    int onehot_ok = 1;
    for (int i = 0; i < B * Nl; ++i) { float su = 0; for (int k = 0; k < P; ++k) su += xp[(size_t)i * (3 + P) + 3 + k]; if (su != 1.0f) onehot_ok = 0; }
    printf("chain of %d steps on %d pockets: evaluations %llu, edges/eval %.0f, max_rel_com_error %.2e, max_cog %.2e, nan_resets %lld, one-hot rows valid %d, x[0]=(%.3f %.3f %.3f)\n",
           K, B, (unsigned long long)c.evaluations, (double)c.edges / c.evaluations, rel, cog, (long long)nans, onehot_ok, xp[0], xp[1], xp[2]);
    cmdgen_destroy(h);
    return (onehot_ok && nans == 0 && rel < 1e-2f && c.evaluations == (unsigned long long)K + 1) ? 0 : 2;
}
