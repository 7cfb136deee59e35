// cmdgen_host.h - host-side state shared by cmdgen_api.hip and cmdgen_train.hip
#pragma once
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

extern std::string g_create_error;
struct TrainState;               // cmdgen_train.hip
void cmdgen_train_free(TrainState*);


struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
};

struct cmdgen_handle {
    cmdgen_config cfg{};
    int device = 0;
    std::string err;
    Dims dims{};
    // weights
    std::map<std::string, std::vector<float>> staged;
    bool finalized = false;
    std::vector<void*> weight_allocs;
    std::vector<LayerW> layers;
    SmallW small{};
    std::vector<float> gamma;              // host copy of the table [T+1]
    // layout + workspace
    bool have_layout = false;
    std::vector<int64_t> cur_nphar, cur_npocket;
    std::vector<void*> layout_allocs;
    Layout lay{};
    Work work{};
    int64_t ecap = 0, eccap = 0;
    int64_t cap_B = 0, cap_Nl = 0, cap_Np = 0, cap_N = 0, cap_e = 0, cap_ec = 0;   // allocated capacities of the workspaces
    int edge_grid = 512, coord_grid = 256;
    int e128_fused = 3;                    // bit 0 / 1: fused main loop of the 128-row message / coordinate kernel (pick_tiles)
    int n_cus = 256;
    int node_mt = 64, edge_mt = 64, coord_mt = 64;   // rows per tile, chosen in cmdgen_set_layout
    bool gemm_split = true;                // tiles of >= 32 rows multiply on the bf16 matrix pipe (cmdgen_set_gemm_mode)
    int64_t* d_gid = nullptr;
    int* idx_blk[2] = {nullptr, nullptr};  // two copies of the index block: a new layout is written to the one the previous layout's kernels do not read
    int* idx_stage[2] = {nullptr, nullptr}; // pinned staging of the same size (cmdgen_set_layout_on_stream)
    hipEvent_t idx_ev[2] = {nullptr, nullptr};
    int idx_cur = 0;
    int64_t idx_ints = 0;
    // chain
    std::vector<void*> chain_allocs;
    ChainBuf chain{};
    int chain_K = -1;
    bool chain_steps_out = false;
    unsigned int* d_cog = nullptr;
    float *pk_c = nullptr, *pk_P0 = nullptr, *pk_Q0 = nullptr, *pk_dh = nullptr, *pk_dP = nullptr, *pk_dQ = nullptr, *pk_t01 = nullptr;   // PocketCache storage
    std::vector<float> user_coef;          // optional host-supplied step table
    int user_coef_K = -1;
    hipGraphExec_t step_graph = nullptr;
    hipStream_t own_stream = nullptr;      // used when the caller's stream is the legacy default stream (not capturable)
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    const float* graph_noise = nullptr; float* graph_zsteps = nullptr; float* graph_psteps = nullptr; hipStream_t graph_stream = nullptr;
    unsigned long long graph_seed = 0;
    int graph_steps = 0;
    // joint-model chain
    std::vector<void*> joint_allocs;
    JointBuf joint{};
    float* eps_pocket_tmp = nullptr;       // [Np][3+R] evaluation output of the joint chain
    unsigned int* joint_cog = nullptr;
    int joint_steps = -1;                  // denoising steps of the prepared plan
    std::vector<int> joint_key;            // (K, resamplings, jump, inpaint) of the prepared plan
    bool last_chain_joint = false;
    hipGraphExec_t joint_graph = nullptr;
    const void* jg_key[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned long long jg_seed = 0; int jg_steps = 0;
    TrainState* train = nullptr;           // training workspace (cmdgen_train.hip)
    float* h_norm = nullptr; hipEvent_t norm_ev = nullptr; bool norm_pending = false;   // deferred gradient-norm readback (pinned host float)
    int train_E = 0, train_Ec = 0;         // message / coordinate edges of the last cmdgen_train_forward (cmdgen_query)
    bool train_bf16 = false;               // GEMM operand precision of the training step (cmdgen_train_set_precision)
    bool agg_dirty = false;                // cmdgen_debug_eval_prefix left segment sums in work.agg
    hipStream_t last_stream = nullptr;     // stream most recently handed to this handle (ordering contract of cmdgen_set_layout)
    std::map<std::string, int64_t> opts;   // cmdgen_set_option: explicit launch choices of this handle (absent key = the library's own choice)
    TrainTune tune{};                      // the training step's part of them
    bool kernel_profiling = false;
    std::vector<hipEvent_t> prof_events[3];
};

inline int fail(cmdgen_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}
#define HIPCHK(h, call) do { hipError_t _e = (call); if (_e != hipSuccess) \
    return fail(h, CMDGEN_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)

inline int dev_alloc(cmdgen_handle* h, std::vector<void*>& pool, void** out, size_t bytes, bool zero) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) return fail(h, CMDGEN_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    pool.push_back(*out);
    if (zero) { e = hipMemset(*out, 0, bytes); if (e != hipSuccess) return fail(h, CMDGEN_EHIP, "hipMemset failed"); }
    return 0;
}
inline void free_pool(std::vector<void*>& pool) { for (void* p : pool) hipFree(p); pool.clear(); }


int check_ready(cmdgen_handle* h);
int begin_work(cmdgen_handle* h, hipStream_t s);   // check_ready + device + workspace invariants; remembers the stream
EvalLaunch make_launch(cmdgen_handle* h);
inline int64_t opt_of(const cmdgen_handle* h, const char* key, int64_t dflt) { auto it = h->opts.find(key); return it == h->opts.end() ? dflt : it->second; }
inline bool opt_set(const cmdgen_handle* h, const char* key) { return h->opts.find(key) != h->opts.end(); }
