// cmdgen_train.hip - the training step behind the C ABI: forward with saved activations,
// backward to parameter gradients, AdamW(amsgrad) and the gradient norm, all on flat fp32 parameter / gradient
// buffers owned by the caller (one contiguous bucket: a single RCCL all-reduce per step for data parallelism).
// Conditional and joint (update_pocket_coords = 1) models.  Kernels: kernels_train.hip.
#include "cmdgen_host.h"
#include <functional>

void cmdgen_launch_edges(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket, hipStream_t s);
void cmdgen_launch_eval(const EvalLaunch& a, const float* xh_phar, const float* xh_pocket, const float* t_arr, const float4* coef,
                        ChainState* chain, float* eps_phar, float* eps_pocket, hipStream_t s, hipEvent_t* ev);
void cmdgen_launch_nan_fix(const EvalLaunch& a, float* eps_phar, hipStream_t s);
void cmdgen_launch_save_positions(const EvalLaunch& a, float4* X, hipStream_t s);
struct RepackFrag { int src_off, ld, out, in, row_split, col_shift; float* dst32; float* dst16; };      // kernels_train.hip
struct RepackMisc { int src_off, ld, rows, cols; float* dst; };
void tr_repack(const float* theta, const void* frag_tab, int n_frag, int max_frag4, const void* misc_tab, int n_misc, int max_misc,
               hipStream_t s);
void cmdgen_sgemm(bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                  int ldc, const float* bias, float alpha, bool accumulate, int split_k, hipStream_t s,
                  int epi = 0, float* aux = nullptr, int ldaux = 0, bool bf16 = false);
struct WgradBatch {               // kernels_train.hip: up to 8 weight gradients dW (+)= dY^T X (+ bias gradients) in one launch
    const float* dy[8]; const float* x[8]; float* dw[8]; float* db[8];
    int M[8], N[8], lddy[8], ldx[8], ldw[8];
    int n;
    int xs[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // 1: X_p holds pre-activations, SiLU applied while staging
};
void cmdgen_wgrad_group(const WgradBatch& g, int K, bool bf16, hipStream_t s, bool split3 = false, bool force3 = false);
struct RepackSplitT { int src_off, ld; void* dst; int transpose; };                                       // kernels_train.hip
void tr_repack_split_t(const float* theta, const void* tab, int n, hipStream_t s);
struct RepackHalf { int src_off, ld; void* dst; float* sc; int transpose; };                                 // kernels_train.hip
void tr_repack_half(const float* theta, const void* tab, int n_plain, int n, hipStream_t s);
struct RepackHalf16 { int src_off, ld, out, in, row_split, col_shift; void* dst; float* sc; };            // kernels_train.hip
void tr_repack_half16(const float* theta, const void* tab, int n, int max8, hipStream_t s);
void cmdgen_dgrad_split(int M, const float* A0, const void* W0, const float* A1, const void* W1, float* Y, bool accumulate, float div,
                        const float* pre, hipStream_t s, int pieces = 3, const void* W0b = nullptr, float* Yb = nullptr,
                        bool accumulate_b = false, float div_b = 1.0f, int force_mt = 0, const float* Yin = nullptr, const float* rowdiv_b = nullptr);
void tr_reduce_pair(int E, int H, const float* scratch_a, float* out_w, float* out_b, const float* scratch_t, float* dWcol, int ldw, hipStream_t s);
void tr_silu_bwd(float* g, const float* pre, size_t n, hipStream_t s);
void tr_scale(float* x, float a, size_t n, hipStream_t s);
void tr_scale_rows(float* x, const float* div, int H, size_t n, hipStream_t s);
void tr_coord_out_bwd(int E, const int* row, const int* col, const float4* X, const float* phi, int use_tanh, float range,
                      float nc, const float* dacc, float dacc_div, int n_moving, float* dphi, float4* dcd, hipStream_t s, const float* adiv = nullptr);
void tr_edge_tail_bwd(int E, int H, const int* row, const int* col, const float* g, const float* d0, const float* Wcol, int ldw,
                      const float4* X, float nc, const float4* dcd, int n_moving, float* dP, float* dQ, float* dWcol, float* dX,
                      float* scratch, hipStream_t s);
size_t tr_edge_tail_scratch_floats(size_t E, size_t H);
void cmdgen_dgrad_tail(int E, const float* dY, const void* Wt, const float* pre1, const int* row, const int* col, const float* d0,
                       const float* Wcol, int ldw, const float4* X, float nc, const float4* dcd, int n_moving, float* dP, float* dQ,
                       float* dWcol, float* dX, float* scratch, int pieces, hipStream_t s, bool defer_reduce = false);
size_t tr_partial_scratch_floats(size_t E, size_t H);
void tr_gate_bwd(int E, int H, const int* row, const float* pre2, const float* wa, const float* z, int attention, const float* dagg,
                 float* dpre2, float* scratch, float* d_wa, float* d_ba, float* zero, size_t zero_floats, hipStream_t s,
                 bool defer_reduce = false);
struct CoordOutArgs { const int* row; const int* col; const float4* X; const float* phi; int use_tanh; float range, norm_constant;
                      const float* dacc; float dacc_div; const float* adiv; float4* dcd_out; };       // kernels_train.hip
void tr_head_bwd(int E, int H, const float* dphi, const float* w5, const float* pre7, float* dpre7, float* scratch, float* d_w5,
                 float* zero, size_t zero_floats, hipStream_t s, bool defer_reduce = false, const CoordOutArgs* co = nullptr);
void tr_colsum(int E, int H, const float* X, int ldx, const float* sv, float* out, int ldo, hipStream_t s);
void tr_center_per_sample(const Layout& lay, float* v, hipStream_t s);
void tr_eps_bwd(int n_rows, int F, int row0, const float* deps, float* dvel, float* ddec, hipStream_t s);
void tr_bwd_init(int Nl, int N, int P, int dyn, const float* deps, float* dX, float* ddec, float* dhfin, hipStream_t s);
void tr_adamw(size_t n, float* theta, const float* grad, float* m, float* v, float* vmax, float lr, float b1, float b2,
              float eps, float wd, float bias1, float bias2_sqrt, float clip, hipStream_t s, const float* sqnorm = nullptr,
              float max_norm = 0.f, int skip_nonfinite = 0);
void tr_sqsum(size_t n, const float* x, float* out, hipStream_t s);
void tr_norm_guard(float* sq, const int* nan_flag, hipStream_t s);
void tr_noise(const Layout& lay, const Dims& d, const float* px, const float* poh, const float* qx, const float* qoh, const float* tab,
              const float* eps, float* z_t, float* xh_pocket, float* klsum, hipStream_t s);
void tr_noise_joint(const Layout& lay, const Dims& d, const float* px, const float* poh, const float* qx, const float* qoh, const float* tab,
                    const float* raw_l, const float* raw_q, float* z_l, float* z_q, float* e_l, float* e_q, float* klsum, hipStream_t s);
void tr_loss_joint(const Layout& lay, const Dims& d, int l2, float T, const float* net_l, const float* net_q, const float* e_l, const float* e_q,
                   const float* z_l, const float* z_q, const float* poh, const float* qoh, const float* tab, const float* klsum, float* terms,
                   float* d_l, float* d_q, float* means, hipStream_t s);
void tr_loss(const Layout& lay, const Dims& d, int l2, float T, const float* net, const float* eps, const float* z_t, const float* poh,
             const float* tab, const float* klsum, float* terms, float* d_eps, float* means, hipStream_t s);

// ---------------------------------------------------------------------------------
// flat parameter layout: the reference's registration order (state_dict order below 'dynamics.'), weight then bias
// ---------------------------------------------------------------------------------
struct PRef { size_t w = 0, b = 0; int out = 0, in = 0; bool has_bias = true; std::string name; };

struct ParamTable {
    PRef pe0, pe2, pd0, pd2, re0, re2, rd0, rd2, emb, embo;
    struct Blk { PRef e0, e2, n0, n2, att, c0, c2, c4; };
    std::vector<Blk> blk;
    std::vector<PRef*> order;
    size_t total = 0;
};

static void build_table(const Dims& d, ParamTable& t) {
    t.blk.resize((size_t)d.L * d.S);
    t.order.clear();
    size_t off = 0;
    auto add = [&](PRef& r, const std::string& name, int out, int in, bool bias) {
        r.name = name; r.out = out; r.in = in; r.has_bias = bias;
        off = (off + 3) & ~(size_t)3;                       // every tensor starts 16-byte aligned (vector loads in the GEMM);
        r.w = off; off += (size_t)out * in;                 // the padding floats are never read and stay zero
        if (bias) { off = (off + 3) & ~(size_t)3; r.b = off; off += out; }
        t.order.push_back(&r);
    };
    const int P = d.P, R = d.R, J = d.J, H = d.H;
    add(t.pe0, "phar_encoder.0", 2 * P, P, true);      add(t.pe2, "phar_encoder.2", J, 2 * P, true);
    add(t.pd0, "phar_decoder.0", 2 * P, J, true);      add(t.pd2, "phar_decoder.2", P, 2 * P, true);
    add(t.re0, "residue_encoder.0", 2 * R, R, true);   add(t.re2, "residue_encoder.2", J, 2 * R, true);
    add(t.rd0, "residue_decoder.0", 2 * R, J, true);   add(t.rd2, "residue_decoder.2", R, 2 * R, true);
    add(t.emb, "egnn.embedding", H, d.dyn, true);      add(t.embo, "egnn.embedding_out", d.dyn, H, true);
    // one Blk per GCL ("unit" l * S + sub, egnn_new.py:127-131) in registration order: gcl_0 .. gcl_{S-1}, then the block's gcl_equiv, whose
    // tensors ride with the block's LAST unit (c0 / c2 / c4 of the other units stay empty)
    for (int l = 0; l < d.L; ++l) {
        for (int sub = 0; sub < d.S; ++sub) {
            const std::string g = "egnn.e_block_" + std::to_string(l) + ".gcl_" + std::to_string(sub) + ".";
            ParamTable::Blk& b = t.blk[(size_t)l * d.S + sub];
            add(b.e0, g + "edge_mlp.0", H, 2 * H + 2, true);   add(b.e2, g + "edge_mlp.2", H, H, true);
            add(b.n0, g + "node_mlp.0", H, 2 * H, true);       add(b.n2, g + "node_mlp.2", H, H, true);
            if (d.attention) add(b.att, g + "att_mlp.0", 1, H, true);
        }
        const std::string c = "egnn.e_block_" + std::to_string(l) + ".gcl_equiv.";
        ParamTable::Blk& b = t.blk[(size_t)l * d.S + d.S - 1];
        add(b.c0, c + "coord_mlp.0", H, 2 * H + 2, true);  add(b.c2, c + "coord_mlp.2", H, H, true);
        add(b.c4, c + "coord_mlp.4", 1, H, false);
    }
    t.total = (off + 3) & ~(size_t)3;
}

struct TrainState {
    ParamTable tab;
    std::vector<void*> node_allocs, edge_allocs;
    int E = 0, Ec = 0;                  // edges of the last forward
    size_t ecap = 0, eccap = 0;
    bool have_forward = false;
    bool split_packs_valid = false;     // the last forward re-packed the transposed split fragments (backward may use them)
    int wsilu = 0;                      // which activations the last forward did NOT store (option wgrad_silu)
    bool fwd_on_half = false;           // the last forward's tile kernels ran on the half engine (fp16 range: a non-finite gradient after it skips the update)
    bool bf16 = false;                  // GEMM operands in bf16 (fp32 accumulation); default exact fp32
    const float* theta = nullptr;       // parameters used by the last forward (backward reads the same)
    const float* xh_phar = nullptr; const float* xh_pocket = nullptr;
    // node level
    float *enc1_l, *enca_l, *enc1_p, *enca_p, *enc_out, *hdyn, *h /* [L+1][N][H] */, *P, *Q, *aggn /* [L] */,
          *pre3 /* [L] */, *nact, *accx, *hfin, *dec1, *deca, *dec_out;
    float4* X;                          // [L+1][N]
    // edge level (saved per block)
    float *pre1, *pre2, *z, *pre6, *pre7, *phi;
    float *act1, *act6;                 // SiLU of pre1 / pre6, written by the producing kernel (the x operands of two weight gradients)
    // edge level scratch
    float *actA, *actB, *dphi, *tail_scratch, *part_scratch;
    float4* dcd;
    // backward node level
    float *dh, *dX, *dagg, *dP, *dQ, *dn, *dhfin, *ddec, *ddeca, *dhdyn, *denca_l, *denca_p;
    float *vel, *qdec1, *qdeca, *qdec_out, *dqdec, *dqdeca;     // velocity [N][4]; residue decoder (joint model's pocket output)
    float* d_scalar;                    // [4] device scalars (sum of squares, ...)
    // Weight gradients on a second stream (option wgrad_stream, default on): they are off the chain of data gradients, and most of
    // them are launches that cannot fill the chip (256 x 256 outputs over 4k-36k rows, split-K) - beside the data-gradient kernels they
    // cost a fraction of what they cost alone.  What they read must outlive the main stream's next writer of the same buffer, so the
    // buffers one block's weight gradients read while the next kernels of the chain write alternate: dpre2 / dpre7 by block parity,
    // dP | dQ separately for the coordinate and the message list, dn by block parity, dh between the two sides of the node model.
    int* h_tot = nullptr; hipEvent_t tot_ev = nullptr;          // pinned landing place of the list lengths and the event behind their copy (cmdgen_train_forward)
    hipStream_t ws = nullptr, ws_low = nullptr;      // (ws_low: the same at the device's lowest stream priority, wgrad_stream = 2)
    hipEvent_t br_in = nullptr, br_out = nullptr;    // a caller on the legacy default stream: the pass runs on ws, bracketed by these (train_backward_stages)
    std::vector<hipEvent_t> evs, evs2;
    float *part_x[4] = {nullptr, nullptr, nullptr, nullptr}, *tail_x[4] = {nullptr, nullptr, nullptr, nullptr};
    float *actA2 = nullptr, *actB2 = nullptr, *dPx[3] = {nullptr, nullptr, nullptr}, *dn2 = nullptr, *dh2 = nullptr, *dh3 = nullptr;
    // the fused forward (the sampler's evaluation kernels with save hooks): per-step packed copies of the parameters
    std::vector<void*> pack_allocs;
    std::vector<LayerW> layers;         // device pointers: packed fragments + vectors inside theta (rebuilt per call: theta is the caller's)
    struct PackBlk { float *pq_e32, *pq_e16, *w2_32, *w2_16, *w3_32, *w3_16, *w4_32, *w4_16, *pq_c32, *pq_c16, *w7_32, *w7_16, *rd_e, *rd_c;
                     // split-bf16 fragment packs of the TRANSPOSED 256 x 256 blocks (data gradients, cmdgen_dgrad_split); H = 256 only
                     void *t_e0a, *t_e0b, *t_e2, *t_n0a, *t_n0b, *t_n2, *t_c0a, *t_c0b, *t_c2;
                     void *s_e2, *s_c2;          // split packs of edge_mlp.2 / coord_mlp.2 themselves: the forward's two edge kernels
                     void *h_e2, *h_c2; float *hs_e2, *hs_c2;
                     void *h16_w3, *h16_w4, *h16_pqc, *h16_pqe; float *hs_w3, *hs_w4, *hs_pqc, *hs_pqe; };   // 16-row half packs of the node kernel (k_repack_half16)   // ... and their half-engine packs with the device-side {scale, 1 / scale} (k_repack_half)
    std::vector<PackBlk> pack;          // rd_e / rd_c: [2][H] radial column then d0 column of edge_mlp.0 / coord_mlp.0
    float *emb_wT = nullptr, *embo_wT = nullptr;
    void *frag_tab = nullptr, *misc_tab = nullptr, *split_tab = nullptr, *half_tab = nullptr, *half16_tab = nullptr;
    int n_frag = 0, n_misc = 0, max_frag4 = 0, max_misc = 0, n_split = 0, n_half = 0, n_half_fwd = 0, n_half16 = 0, max_half16 = 0;
};

void cmdgen_train_free(TrainState* t) {
    if (!t) return;
    free_pool(t->node_allocs); free_pool(t->edge_allocs); free_pool(t->pack_allocs);
    for (hipEvent_t e : t->evs) hipEventDestroy(e);
    for (hipEvent_t e : t->evs2) hipEventDestroy(e);
    if (t->h_tot) hipHostFree(t->h_tot);
    if (t->tot_ev) hipEventDestroy(t->tot_ev);
    if (t->br_in) hipEventDestroy(t->br_in);
    if (t->br_out) hipEventDestroy(t->br_out);
    if (t->ws) hipStreamDestroy(t->ws);
    if (t->ws_low) hipStreamDestroy(t->ws_low);
    delete t;
}

static int ensure_state(cmdgen_handle* h) {
    if (h->train) return 0;
    if (h->dims.H > 256) return fail(h, CMDGEN_ESTATE, "the training step is built for hidden_nf <= 256");
    if (h->dims.sin)
        return fail(h, CMDGEN_ESTATE, "the training step supports sin_embedding False only (every shipped config); this handle samples only");
    TrainState* t = new TrainState();
    build_table(h->dims, t->tab);
    t->bf16 = h->train_bf16;
    const Dims& d = h->dims;
    // sized by the layout CAPACITIES (cmdgen_set_layout): the state survives every new batch that fits them
    const size_t N = h->cap_N, Nl = h->cap_Nl, Np = h->cap_Np, H = d.H, L = d.L, U = (size_t)d.L * d.S;      // U: GCLs ("units")
    int rc; void* p;
#define NA(dst, type, count) do { rc = dev_alloc(h, t->node_allocs, &p, (size_t)(count) * sizeof(type), true); \
        if (rc) { cmdgen_train_free(t); return rc; } dst = (type*)p; } while (0)
    NA(t->enc1_l, float, Nl * 2 * d.P); NA(t->enca_l, float, Nl * 2 * d.P);
    NA(t->enc1_p, float, Np * 2 * d.R); NA(t->enca_p, float, Np * 2 * d.R);
    NA(t->enc_out, float, N * d.J); NA(t->hdyn, float, N * d.dyn);
    NA(t->h, float, (U + 1) * N * H); NA(t->X, float4, (L + 1) * N);
    NA(t->P, float, N * H); NA(t->Q, float, N * H); NA(t->aggn, float, U * N * H); NA(t->pre3, float, U * N * H);
    NA(t->nact, float, U * N * H); NA(t->accx, float, N * 4); NA(t->hfin, float, N * d.dyn);
    NA(t->dec1, float, Nl * 2 * d.P); NA(t->deca, float, Nl * 2 * d.P); NA(t->dec_out, float, Nl * d.P);
    NA(t->dh, float, N * H); NA(t->dX, float, N * 4); NA(t->dagg, float, N * H);
    NA(t->dP, float, 2 * N * H); t->dQ = t->dP + N * H;      // adjacent: zeroed by one memset
    NA(t->dn, float, N * H); NA(t->dhfin, float, N * d.dyn);
    for (int i = 0; i < 3; ++i) NA(t->dPx[i], float, 2 * N * H);       // (dP | dQ pairs: coordinate / message list by block parity, with t->dP)
    NA(t->dn2, float, N * H); NA(t->dh2, float, N * H); NA(t->dh3, float, N * H);
    NA(t->ddec, float, Nl * d.P); NA(t->ddeca, float, Nl * 2 * d.P); NA(t->dhdyn, float, N * d.dyn);
    NA(t->denca_l, float, Nl * 2 * d.P); NA(t->denca_p, float, Np * 2 * d.R);
    NA(t->vel, float, N * 4); NA(t->qdec1, float, Np * 2 * d.R); NA(t->qdeca, float, Np * 2 * d.R);
    NA(t->qdec_out, float, Np * d.R); NA(t->dqdec, float, Np * d.R); NA(t->dqdeca, float, Np * 2 * d.R);
    NA(t->d_scalar, float, 4);
#undef NA
    {   // packed-parameter buffers and the (offset-based, theta-independent) re-pack tables
        const ParamTable& tb = t->tab;
        std::vector<RepackFrag> ft; std::vector<RepackMisc> mt;
        auto alloc = [&](size_t floats, float** out) -> int {
            int r = dev_alloc(h, t->pack_allocs, &p, floats * sizeof(float), true); if (!r) *out = (float*)p; return r; };
        const int ld1 = 2 * (int)H + 2;
        t->pack.resize(U);
        for (size_t l = 0; l < U && !rc; ++l) {                 // (l runs over the units; coordinate tensors exist on a block's last unit only)
            TrainState::PackBlk& k = t->pack[l];
            const ParamTable::Blk& b = tb.blk[l];
            auto frag = [&](const PRef& r, int out, int in, int row_split, int col_shift, float** d32, float** d16) {
                if (rc) return;
                rc = alloc((size_t)out * in, d32); if (rc) return;
                rc = alloc((size_t)out * in, d16); if (rc) return;
                ft.push_back(RepackFrag{(int)r.w, r.in, out, in, row_split, col_shift, *d32, *d16});
            };
            frag(b.e0, 2 * (int)H, (int)H, (int)H, (int)H, &k.pq_e32, &k.pq_e16);      // rows 0..H-1: columns 0..H-1 (-> P); rows H..: columns H..2H-1 (-> Q)
            frag(b.e2, (int)H, (int)H, 0, 0, &k.w2_32, &k.w2_16);
            frag(b.n0, (int)H, 2 * (int)H, 0, 0, &k.w3_32, &k.w3_16);
            frag(b.n2, (int)H, (int)H, 0, 0, &k.w4_32, &k.w4_16);
            if (b.c0.out) {
            frag(b.c0, 2 * (int)H, (int)H, (int)H, (int)H, &k.pq_c32, &k.pq_c16);
            frag(b.c2, (int)H, (int)H, 0, 0, &k.w7_32, &k.w7_16);
            }
            if (!rc) rc = alloc(2 * H, &k.rd_e);
            if (!rc) rc = alloc(2 * H, &k.rd_c);
            if (!rc) {
                mt.push_back(RepackMisc{(int)b.e0.w + 2 * (int)H, ld1, (int)H, 2, k.rd_e});
                if (b.c0.out) mt.push_back(RepackMisc{(int)b.c0.w + 2 * (int)H, ld1, (int)H, 2, k.rd_c});
            }
        }
        std::vector<RepackSplitT> st;
        if (H == 256) {
            for (size_t l = 0; l < U && !rc; ++l) {
                TrainState::PackBlk& k = t->pack[l];
                const ParamTable::Blk& b = tb.blk[l];
                auto tp = [&](const PRef& r, int col0, void** dst, int transpose = 1) {
                    if (rc || !r.out) return;
                    float* q = nullptr;
                    rc = alloc((size_t)H * H * 6 / 4, &q); if (rc) return;          // three bf16 pieces per weight
                    *dst = q;
                    st.push_back(RepackSplitT{(int)r.w + col0, r.in, q, transpose});
                };
                tp(b.e2, 0, &k.s_e2, 0); tp(b.c2, 0, &k.s_c2, 0);
                tp(b.e0, 0, &k.t_e0a); tp(b.e0, (int)H, &k.t_e0b); tp(b.e2, 0, &k.t_e2);
                tp(b.n0, 0, &k.t_n0a); tp(b.n0, (int)H, &k.t_n0b); tp(b.n2, 0, &k.t_n2);
                tp(b.c0, 0, &k.t_c0a); tp(b.c0, (int)H, &k.t_c0b); tp(b.c2, 0, &k.t_c2);
            }
            if (!rc) rc = dev_alloc(h, t->pack_allocs, &p, st.size() * sizeof(RepackSplitT), false);
            if (!rc) { t->split_tab = p; hipMemcpy(p, st.data(), st.size() * sizeof(RepackSplitT), hipMemcpyHostToDevice); t->n_split = (int)st.size(); }
            std::vector<RepackHalf> ht;
            for (size_t l = 0; l < U && !rc; ++l) {
                TrainState::PackBlk& k = t->pack[l];
                const ParamTable::Blk& b = tb.blk[l];
                auto hp = [&](const PRef& r, void** dst, float** sc) {
                    if (rc || !r.out) return;
                    float* q = nullptr;
                    rc = alloc((size_t)H * H + 4, &q); if (rc) return;              // two fp16 pieces per weight, then {scale, 1 / scale}
                    *dst = q; *sc = q + (size_t)H * H;
                    ht.push_back(RepackHalf{(int)r.w, r.in, q, *sc, 0});
                };
                hp(b.e2, &k.h_e2, &k.hs_e2); hp(b.c2, &k.h_c2, &k.hs_c2);
            }
            t->n_half_fwd = (int)ht.size();
            if (!rc) rc = dev_alloc(h, t->pack_allocs, &p, ht.size() * sizeof(RepackHalf), false);
            if (!rc) { t->half_tab = p; hipMemcpy(p, ht.data(), ht.size() * sizeof(RepackHalf), hipMemcpyHostToDevice); t->n_half = (int)ht.size(); }
            std::vector<RepackHalf16> h16;
            for (size_t l = 0; l < U && !rc; ++l) {
                TrainState::PackBlk& k = t->pack[l];
                const ParamTable::Blk& b = tb.blk[l];
                auto hp = [&](const PRef& r, int out, int in, int row_split, int col_shift, void** dst, float** sc) {
                    if (rc || !r.out) return;
                    float* q = nullptr;
                    rc = alloc((size_t)out * in + 4, &q); if (rc) return;           // two fp16 pieces per weight, then {scale, 1 / scale}
                    *dst = q; *sc = q + (size_t)out * in;
                    h16.push_back(RepackHalf16{(int)r.w, r.in, out, in, row_split, col_shift, q, *sc});
                    t->max_half16 = std::max(t->max_half16, out * in / 8);
                };
                hp(b.n0, (int)H, 2 * (int)H, 0, 0, &k.h16_w3, &k.hs_w3);
                hp(b.n2, (int)H, (int)H, 0, 0, &k.h16_w4, &k.hs_w4);
                hp(b.c0, 2 * (int)H, (int)H, (int)H, (int)H, &k.h16_pqc, &k.hs_pqc);
                hp(b.e0, 2 * (int)H, (int)H, (int)H, (int)H, &k.h16_pqe, &k.hs_pqe);
            }
            if (!rc) rc = dev_alloc(h, t->pack_allocs, &p, h16.size() * sizeof(RepackHalf16), false);
            if (!rc) { t->half16_tab = p; hipMemcpy(p, h16.data(), h16.size() * sizeof(RepackHalf16), hipMemcpyHostToDevice); t->n_half16 = (int)h16.size(); }
        }
        if (!rc) rc = alloc((size_t)H * d.dyn, &t->emb_wT);
        if (!rc) rc = alloc((size_t)H * d.dyn, &t->embo_wT);
        if (!rc) {
            mt.push_back(RepackMisc{(int)tb.emb.w, d.dyn, (int)H, d.dyn, t->emb_wT});       // [H][dyn] -> [dyn][H]
            mt.push_back(RepackMisc{(int)tb.embo.w, (int)H, d.dyn, (int)H, t->embo_wT});    // [dyn][H] -> [H][dyn]
            rc = dev_alloc(h, t->pack_allocs, &p, ft.size() * sizeof(RepackFrag), false);
        }
        if (!rc) {
            t->frag_tab = p; hipMemcpy(p, ft.data(), ft.size() * sizeof(RepackFrag), hipMemcpyHostToDevice);
            rc = dev_alloc(h, t->pack_allocs, &p, mt.size() * sizeof(RepackMisc), false);
        }
        if (!rc) { t->misc_tab = p; hipMemcpy(p, mt.data(), mt.size() * sizeof(RepackMisc), hipMemcpyHostToDevice); }
        if (rc) { cmdgen_train_free(t); return rc; }
        t->n_frag = (int)ft.size(); t->n_misc = (int)mt.size();
        for (const RepackFrag& f : ft) t->max_frag4 = std::max(t->max_frag4, f.out * f.in / 4);
        for (const RepackMisc& m : mt) t->max_misc = std::max(t->max_misc, m.rows * m.cols);
    }
    if (hipStreamCreateWithFlags(&t->ws, hipStreamNonBlocking) != hipSuccess) { t->ws = nullptr; (void)hipGetLastError(); }
    {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
            hipStreamCreateWithPriority(&t->ws_low, hipStreamNonBlocking, least) != hipSuccess) { t->ws_low = nullptr; (void)hipGetLastError(); }
    }
    h->train = t;
    return 0;
}

// The second stream of the backward pass (TrainState::ws).  Every event operation on the main stream costs its queue a ~6 us bubble
// (profiles/r05_ad_train_chain.txt), so the protocol is coarse: the weight gradients of a block are queued TOGETHER at the block's end -
// fork(): the side stream waits for everything queued on the main stream so far (one event) - and everything they read sits in buffers
// that the main stream writes again only two blocks later (TrainState: rotating dpre / dP | dQ / dn / dh), so one wait per block for the
// side work of two blocks ago (mark() / wait()) covers every hazard.  join(): the main stream waits for all side work (end of every
// backward call: the caller's next kernel - optimizer, all-reduce - sees complete gradients).  on = false: everything on the caller's stream.
struct SideStream {
    hipStream_t main, ws; bool on; std::vector<hipEvent_t>* pool; size_t next = 0;
    bool forked = false;
    hipEvent_t ev() {
        if (next == pool->size()) { hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming); pool->push_back(e); }
        return (*pool)[next++];
    }
    hipStream_t fork() {
        if (!on) return main;
        hipEvent_t e = ev(); hipEventRecord(e, main); hipStreamWaitEvent(ws, e, 0); forked = true;
        return ws;
    }
    hipEvent_t mark() { if (!on) return nullptr; hipEvent_t e = ev(); hipEventRecord(e, ws); return e; }
    void wait(hipEvent_t e) { if (on && e) hipStreamWaitEvent(main, e, 0); }
    void join() {
        if (!on || !forked) return;
        hipEvent_t e = ev(); hipEventRecord(e, ws); hipStreamWaitEvent(main, e, 0); forked = false;
    }
};

static int ensure_edges(cmdgen_handle* h, TrainState* t, int E, int Ec) {
    if ((size_t)E <= t->ecap && (size_t)Ec <= t->eccap) return 0;
    hipDeviceSynchronize();
    free_pool(t->edge_allocs);
    const size_t ec = (size_t)(E * 1.25) + 64, ecc = (size_t)(Ec * 1.25) + 64, H = h->dims.H, L = h->dims.L, U = L * h->dims.S;
    const size_t em = ec > ecc ? ec : ecc;
    int rc; void* p;
#define EA(dst, type, count) do { rc = dev_alloc(h, t->edge_allocs, &p, (size_t)(count) * sizeof(type), false); \
        if (rc) return rc; dst = (type*)p; } while (0)
    EA(t->pre1, float, U * ec * H); EA(t->pre2, float, U * ec * H); EA(t->z, float, U * ec);
    EA(t->act1, float, U * ec * H); EA(t->act6, float, L * ecc * H);      // (unused where the weight gradient forms SiLU(pre) itself: option wgrad_silu)
    EA(t->pre6, float, L * ecc * H); EA(t->pre7, float, L * ecc * H); EA(t->phi, float, L * ecc);
    EA(t->actA, float, em * H); EA(t->actB, float, em * H);
    EA(t->actA2, float, ec * H); EA(t->actB2, float, ecc * H);
    for (int i = 0; i < 4; ++i) {       // [list: coordinate / message][GCL parity]: the partial sums the side stream reduces (tr_reduce_pair)
        EA(t->part_x[i], float, tr_partial_scratch_floats(i < 2 ? ecc : ec, H));
        EA(t->tail_x[i], float, std::max(tr_edge_tail_scratch_floats(i < 2 ? ecc : ec, H), tr_partial_scratch_floats(i < 2 ? ecc : ec, H)));
    }
    EA(t->dphi, float, ecc);
    EA(t->dcd, float4, ecc);
    EA(t->tail_scratch, float, std::max(tr_edge_tail_scratch_floats(em, H), tr_partial_scratch_floats(em, H)));
    EA(t->part_scratch, float, tr_partial_scratch_floats(em, H));
#undef EA
    t->ecap = ec; t->eccap = ecc;
    return 0;
}

// GEMM operand precision of the call in progress (TrainState::bf16; the tiny encoder / decoder products with K < 64
// always run in fp32)
static thread_local bool g_bf16 = false;
extern thread_local TrainTune g_train_tune;           // kernels_train.hip
static inline bool use_bf16(int K) { return g_bf16 && K >= 64; }

// y[M, out] = x[M, in(ldx)] W^T + b      (W, b inside the flat buffer)
static void linear(const float* theta, const PRef& r, int col0, int in, int M, const float* x, int ldx, float* y, int ldy,
                   bool bias, bool accumulate, hipStream_t s, float* act = nullptr) {
    cmdgen_sgemm(false, true, M, r.out, in, x, ldx, theta + r.w + col0, r.in, y, ldy, (bias && r.has_bias) ? theta + r.b : nullptr,
                 1.0f, accumulate, 1, s, act ? 1 : 0, act, ldy, use_bf16(in));      // act: SiLU(y) written alongside y
}
// dx[M, in] (+)= dy[M, out] W[:, col0:col0+in]
static void linear_dgrad(const float* theta, const PRef& r, int col0, int in, int M, const float* dy, int lddy, float* dx,
                         int lddx, bool accumulate, hipStream_t s, const float* pre = nullptr) {
    cmdgen_sgemm(false, false, M, in, r.out, dy, lddy, theta + r.w + col0, r.in, dx, lddx, nullptr, 1.0f, accumulate, 1, s,
                 pre ? 2 : 0, const_cast<float*>(pre), lddx, use_bf16(r.out));   // pre: dx *= SiLU'(pre) (the activation that fed this Linear)
}
// dW[:, col0:col0+in] += dy^T x ;  split over the M rows (edges / nodes)
static void linear_wgrad(float* grad, const PRef& r, int col0, int in, int M, const float* dy, int lddy, const float* x,
                         int ldx, hipStream_t s) {
    cmdgen_sgemm(true, false, r.out, in, M, dy, lddy, x, ldx, grad + r.w + col0, r.in, nullptr, 1.0f, true, 0, s, 0, nullptr, 0,
                 g_bf16 && r.out >= 64 && in >= 64);
}

extern "C" int cmdgen_param_count(cmdgen_handle* h, int64_t* n) {
    if (!h || !n) return CMDGEN_EINVAL;
    ParamTable t; build_table(h->dims, t);
    *n = (int64_t)t.total;
    return CMDGEN_OK;
}

extern "C" int cmdgen_param_offset(cmdgen_handle* h, const char* name, int64_t* offset, int64_t* count) {
    if (!h || !name) return CMDGEN_EINVAL;
    ParamTable t; build_table(h->dims, t);
    const std::string k = name;
    for (const PRef* r : t.order) {
        if (k == r->name + ".weight") { if (offset) *offset = (int64_t)r->w; if (count) *count = (int64_t)r->out * r->in; return CMDGEN_OK; }
        if (r->has_bias && k == r->name + ".bias") { if (offset) *offset = (int64_t)r->b; if (count) *count = r->out; return CMDGEN_OK; }
    }
    return fail(h, CMDGEN_EINVAL, "no trainable tensor '%s'", name);
}

extern "C" int cmdgen_train_forward(cmdgen_handle* h, const float* theta, const float* xh_phar, const float* xh_pocket,
                                    const float* t_arr, float* eps_phar, float* eps_pocket, cmdgen_stream stream) {
    if (!h) return CMDGEN_EINVAL;
    if (!h->have_layout) return fail(h, CMDGEN_ESTATE, "no batch layout (cmdgen_set_layout)");
    if (!theta || !xh_phar || !xh_pocket || !t_arr || !eps_phar) return fail(h, CMDGEN_EINVAL, "null device pointer");
    if (h->dims.joint && !eps_pocket) return fail(h, CMDGEN_EINVAL, "the joint model's loss needs eps_pocket");
    hipSetDevice(h->device);
    int rc = ensure_state(h); if (rc) return rc;
    TrainState* t = h->train;
    hipStream_t s = (hipStream_t)stream;
    h->last_stream = s;
    const Dims& d = h->dims;
    const int N = h->lay.N, Nl = h->lay.Nl, Np = h->lay.Np, H = d.H, L = d.L, P = d.P, R = d.R, J = d.J;
    const int ldp = 3 + P, ldq = 3 + R, ld1 = 2 * H + 2;
    // radius graph (same compact lists as the sampler), then the edge counts come to the host: grids and the
    // activation store are sized from them
    EvalLaunch a = make_launch(h);
    a.dead_skip = 0; a.w.need_qc = nullptr; a.w.ehop = nullptr; a.w.hop_levels = 1;       // the training forward skips nothing: no hop levels in its graph pass
    cmdgen_launch_edges(a, xh_phar, xh_pocket, s);
    // the list lengths come to the host through pinned memory and an event of their own: the re-packs below depend on the parameters only
    // and are queued BEHIND the copy, so the device works on them while the host wakes up (a stream synchronize would wait for them too)
    if (!t->h_tot) {
        HIPCHK(h, hipHostMalloc((void**)&t->h_tot, 4 * sizeof(int), hipHostMallocDefault));
        HIPCHK(h, hipEventCreateWithFlags(&t->tot_ev, hipEventDisableTiming));
    }
    HIPCHK(h, hipMemcpyAsync(t->h_tot, h->work.totals, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipEventRecord(t->tot_ev, s));
    // The forward pass IS the sampler's fused evaluation (k_embed, then per block k_edge_msg / k_node / k_edge_coord, then
    // k_readout) with save hooks that keep what the backward pass reads (TrainSave): ~20 launches instead of ~190, GEMMs on
    // the fragment-streaming tile kernels.  The parameters the optimizer has just updated are re-packed on the device.
    // the forward's two edge kernels on the half engine (two fp16 pieces, three MFMAs per product: cmdgen_split.h) wherever the sampler would use it
    const bool fwd_half = h->gemm_split && H == 256 && a.half_engine && a.edge_fullk && t->n_half > 0 && opt_of(h, "train_half", 1) != 0;
    // ... and the node kernel as the sampler's eight-wave 16-row tile (k_node16w)
    // (option train_node16: 16-row tiles for the node kernel at EVERY size - the save-hook form of the node kernel exists on the half engine
    // for these tiles only; larger layouts otherwise fall back to the fp32-instruction k_node<H, 32 / 64, SAVE>)
    if (fwd_half && opt_of(h, "train_node16", 1) != 0) a.node_mt = 16;
    const bool node_half = fwd_half && a.node_mt == 16 && t->n_half16 > 0 && opt_of(h, "train_half", 1) != 2;
    // fp32 fragment packs: with every tile kernel on the half engine only k_embed reads one (block 0's P | Q projection: the table's first entry)
    const bool frag_partial = fwd_half && node_half;
    tr_repack(theta, t->frag_tab, frag_partial ? 1 : t->n_frag, t->max_frag4, t->misc_tab, t->n_misc, t->max_misc, s);
    if (h->gemm_split || t->bf16) tr_repack_split_t(theta, t->split_tab, t->n_split, s);        // data gradients (and the forward's two
    t->split_packs_valid = (h->gemm_split || t->bf16) && t->n_split > 0;                         // edge kernels) on the bf16 matrix pipe
    if (fwd_half) tr_repack_half(theta, t->half_tab, t->n_half_fwd, t->n_half_fwd, s);
    if (node_half) tr_repack_half16(theta, t->half16_tab, t->n_half16, t->max_half16, s);
    HIPCHK(h, hipEventSynchronize(t->tot_ev));
    const int E = t->h_tot[0], Ec = t->h_tot[1];
    rc = ensure_edges(h, t, E, Ec); if (rc) return rc;
    g_bf16 = t->bf16; g_train_tune = h->tune;
    h->train_E = E; h->train_Ec = Ec;
    t->E = E; t->Ec = Ec; t->theta = theta; t->xh_phar = xh_phar; t->xh_pocket = xh_pocket;
    const Work& w = h->work;
    const ParamTable& tb = t->tab;
    const size_t NH = (size_t)N * H;
    const int U = L * d.S;
    t->layers.assign(U, LayerW{});
    for (int l = 0; l < U; ++l) {                       // units
        const ParamTable::Blk& b = tb.blk[l];
        const TrainState::PackBlk& k = t->pack[l];
        LayerW& lw = t->layers[l];
        lw.Wpq_e = WPack{(const float4*)k.pq_e32, (const float4*)k.pq_e16}; lw.b1 = theta + b.e0.b; lw.wr_e = k.rd_e; lw.wd_e = k.rd_e + H;
        lw.W2 = WPack{(const float4*)k.w2_32, (const float4*)k.w2_16, H == 256 ? k.s_e2 : nullptr}; lw.b2 = theta + b.e2.b;
        if (fwd_half) { lw.W2.wh = k.h_e2; lw.W2.wh_dev = k.hs_e2; }
        lw.wa = d.attention ? theta + b.att.w : theta + b.e2.b; lw.ba = d.attention ? theta + b.att.b : theta + b.e2.b;
        lw.W3 = WPack{(const float4*)k.w3_32, (const float4*)k.w3_16}; lw.b3 = theta + b.n0.b;
        lw.W4 = WPack{(const float4*)k.w4_32, (const float4*)k.w4_16}; lw.b4 = theta + b.n2.b;
        if (!b.c0.out) {        // not the block's last GCL: never multiplied (the node kernel skips the projection); valid pointers for the bias prefetches
            lw.Wpq_c = lw.Wpq_e; lw.b6 = lw.b1; lw.wr_c = lw.wr_e; lw.wd_c = lw.wd_e; lw.W7 = lw.W2; lw.b7 = lw.b2; lw.w5 = lw.b2;
        } else {
        lw.Wpq_c = WPack{(const float4*)k.pq_c32, (const float4*)k.pq_c16}; lw.b6 = theta + b.c0.b; lw.wr_c = k.rd_c; lw.wd_c = k.rd_c + H;
        lw.W7 = WPack{(const float4*)k.w7_32, (const float4*)k.w7_16, H == 256 ? k.s_c2 : nullptr}; lw.b7 = theta + b.c2.b; lw.w5 = theta + b.c4.w;
        if (fwd_half) { lw.W7.wh = k.h_c2; lw.W7.wh_dev = k.hs_c2; }
        }
        if (node_half) {
            lw.W3.wh16 = k.h16_w3; lw.W3.wh_dev = k.hs_w3; lw.W4.wh16 = k.h16_w4; lw.W4.wh_dev = k.hs_w4;
            if (b.c0.out) { lw.Wpq_c.wh16 = k.h16_pqc; lw.Wpq_c.wh_dev = k.hs_pqc; }
            else { lw.Wpq_c.wh16 = k.h16_pqe; lw.Wpq_c.wh_dev = k.hs_pqe; }
            lw.Wpq_e.wh16 = k.h16_pqe; lw.Wpq_e.wh_dev = k.hs_pqe;
        }
    }
    SmallW sw{};
    sw.pe0_w = theta + tb.pe0.w; sw.pe0_b = theta + tb.pe0.b; sw.pe2_w = theta + tb.pe2.w; sw.pe2_b = theta + tb.pe2.b;
    sw.pd0_w = theta + tb.pd0.w; sw.pd0_b = theta + tb.pd0.b; sw.pd2_w = theta + tb.pd2.w; sw.pd2_b = theta + tb.pd2.b;
    sw.re0_w = theta + tb.re0.w; sw.re0_b = theta + tb.re0.b; sw.re2_w = theta + tb.re2.w; sw.re2_b = theta + tb.re2.b;
    sw.rd0_w = theta + tb.rd0.w; sw.rd0_b = theta + tb.rd0.b; sw.rd2_w = theta + tb.rd2.w; sw.rd2_b = theta + tb.rd2.b;
    sw.emb_wT = t->emb_wT; sw.emb_b = theta + tb.emb.b; sw.embo_wT = t->embo_wT; sw.embo_b = theta + tb.embo.b;
    TrainSave sv{};
    sv.enc1_l = t->enc1_l; sv.enca_l = t->enca_l; sv.enc1_p = t->enc1_p; sv.enca_p = t->enca_p; sv.hdyn = t->hdyn; sv.h = t->h;
    // option wgrad_silu (bit 0: message list, bit 1: coordinate list): SiLU(pre1) / SiLU(pre6) are not stored - the one consumer, the second
    // layer's weight gradient, forms them while it stages its operand
    const int wsilu = (int)opt_of(h, "wgrad_silu", t->bf16 ? 3 : 0);
    t->wsilu = wsilu;
    sv.pre1 = t->pre1; sv.act1 = (wsilu & 1) ? nullptr : t->act1; sv.pre2 = t->pre2; sv.act2 = nullptr; sv.z = t->z;
    sv.aggn = t->aggn; sv.pre3 = t->pre3; sv.nact = t->nact;
    sv.pre6 = t->pre6; sv.act6 = (wsilu & 2) ? nullptr : t->act6; sv.pre7 = t->pre7; sv.act7 = nullptr; sv.phi = t->phi;
    sv.hfin = t->hfin; sv.dec1 = t->dec1; sv.deca = t->deca; sv.dec_out = t->dec_out;
    sv.qdec1 = t->qdec1; sv.qdeca = t->qdeca; sv.qdec_out = t->qdec_out;
    sv.ecap = t->ecap; sv.eccap = t->eccap;
    a.layers = t->layers.data(); a.sw = sw; a.save = &sv; a.skip_count = 2;
    {   // tile rows of the two edge kernels: the training forward knows its lists' lengths (the sampler's pick_tiles estimates them, and its
        // 128-row kernels have no activation-saving form): 32-row tiles until 64-row ones fill every CU four times over
        auto rows = [&](int n) { return n / 64 >= 4 * h->n_cus ? 64 : (n / 32 >= h->n_cus / 4 ? 32 : 16); };
        auto grid = [&](int n, int mt) { const int cap = (mt >= 64 ? 2 : 4) * h->n_cus, g = (int)((n / mt + 1) * 1.25) + 8; return g < h->n_cus / 4 ? h->n_cus / 4 : (g > cap ? cap : g); };
        if (!opt_set(h, "edge_mt") || a.edge_mt == 128) { a.edge_mt = rows(E); a.edge_grid = grid(E, a.edge_mt); }
        if (!opt_set(h, "coord_mt") || a.coord_mt == 128) { a.coord_mt = rows(Ec); a.coord_grid = grid(Ec, a.coord_mt); }
        if (fwd_half) {     // the half form exists for 32-row full-K tiles (three workgroups per CU)
            a.save_half = 1; a.save_half16 = node_half ? 1 : 0;
            a.edge_mt = 32; a.edge_grid = grid(E, 32); a.coord_mt = 32; a.coord_grid = grid(Ec, 32);
        }
    }
    a.save_split = (h->gemm_split && t->split_packs_valid && H == 256) ? 1 : 0;
    if (h->agg_dirty) { HIPCHK(h, hipMemsetAsync(h->work.agg, 0, NH * sizeof(float), s)); h->agg_dirty = false; }
    a.frag_launches = 0;
    cmdgen_launch_eval(a, xh_phar, xh_pocket, t_arr, nullptr, nullptr, eps_phar, eps_pocket, s, nullptr);
    if (!d.joint) cmdgen_launch_nan_fix(a, eps_phar, s);                    // dynamics.py:129-131 (joint: inside k_vel_com)
    cmdgen_launch_save_positions(a, t->X, s);
    // Only the first fp32 fragment pack was refreshed above when every tile kernel was expected on its half form.  The launchers report what they
    // actually ran: a generic (fragment-reading) tile launch here would have multiplied with the weights of an earlier step - refuse the step
    // instead of training on them (the conditions above and in launch_msg_fullk / launch_coord_fullk / cmdgen_launch_node16w must agree).
    t->fwd_on_half = fwd_half;
    if (frag_partial && a.frag_launches > 0) {
        t->have_forward = false;
        return fail(h, CMDGEN_ESTATE, "internal: %d tile launches of the training forward read fp32 weight fragments this step did not re-pack (set option train_half=0)", a.frag_launches);
    }
    (void)Np; (void)ldp; (void)ldq; (void)ld1; (void)P; (void)R; (void)J; (void)w;
    HIPCHK(h, hipGetLastError());
    t->have_forward = true;
    return CMDGEN_OK;
}

// Stages of the backward pass, in execution order: 0 = readout (decoders, embedding_out), 1..L = blocks L-1..0
// (stage k is block L-k), L+1 = embedding and encoders.  The gradient regions of the flat buffer therefore complete
// from the back: after stage k every tensor of blocks >= L-k is final, so the caller can start the all-reduce of that
// (contiguous) tail while the earlier blocks are still being differentiated.
static int train_backward_stages(cmdgen_handle* h, const float* d_eps_phar, const float* d_eps_pocket, float* grad,
                                 int first_stage, int last_stage, cmdgen_stream stream);

extern "C" int cmdgen_train_backward(cmdgen_handle* h, const float* d_eps_phar, const float* d_eps_pocket, float* grad,
                                     cmdgen_stream stream) {
    if (!h) return CMDGEN_EINVAL;
    return train_backward_stages(h, d_eps_phar, d_eps_pocket, grad, 0, h->dims.L + 1, stream);
}

extern "C" int cmdgen_train_backward_stages(cmdgen_handle* h, const float* d_eps_phar, const float* d_eps_pocket, float* grad,
                                            int32_t first_stage, int32_t last_stage, cmdgen_stream stream) {
    if (!h) return CMDGEN_EINVAL;
    if (first_stage < 0 || last_stage > h->dims.L + 1 || first_stage > last_stage)
        return fail(h, CMDGEN_EINVAL, "stages must satisfy 0 <= first <= last <= n_layers + 1");
    return train_backward_stages(h, d_eps_phar, d_eps_pocket, grad, first_stage, last_stage, stream);
}

static int train_backward_stages(cmdgen_handle* h, const float* d_eps_phar, const float* d_eps_pocket, float* grad,
                                 int first_stage, int last_stage, cmdgen_stream stream) {
    if (!h || !h->train || !h->train->have_forward) return fail(h, CMDGEN_ESTATE, "cmdgen_train_backward needs a preceding cmdgen_train_forward");
    if (!d_eps_phar || !grad) return fail(h, CMDGEN_EINVAL, "null device pointer");
    hipSetDevice(h->device);
    TrainState* t = h->train;
    g_bf16 = t->bf16; g_train_tune = h->tune; g_train_tune.dbg = h->work.dbg;
    hipStream_t s = (hipStream_t)stream;
    h->last_stream = s;
    // The pass forks work onto the handle's side streams and joins them with events.  With the LEGACY DEFAULT stream as the caller's stream that
    // protocol is not reliable (round 6: with a second handle alive in the process, gradients of the side streams' tensors came out 1e-4 ... 1e-3
    // off now and then; any other stream as the caller's: never - tools/train_grad_diag4.py, profiles/r06_d).  Such a call runs on streams of the
    // handle's own, ordered behind the caller's pending work and in front of its later work by two events (what the sampler does for its chains).
    // (No stream is added for this - the process's streams share a handful of hardware queues, and a fifth active stream made two of the pass's
    // streams share one: 2.7 -> 4.5 ms per step inside bench.py - the pass's main chain takes the handle's first side stream, the weight
    // gradients its second, and the embedding stage's third stream is folded into the second.)
    const hipStream_t caller = s;
    const bool bridged = caller == nullptr && g_train_tune.wgrad_stream != 0 && g_train_tune.dgrad_tail != 0 && t->ws && t->ws_low;
    if (bridged) {
        if (!t->br_in) {
            HIPCHK(h, hipEventCreateWithFlags(&t->br_in, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&t->br_out, hipEventDisableTiming));
        }
        HIPCHK(h, hipEventRecord(t->br_in, caller));
        HIPCHK(h, hipStreamWaitEvent(t->ws, t->br_in, 0));
        s = t->ws;
    }
    const Dims& d = h->dims;
    const float* theta = t->theta;
    const int N = h->lay.N, Nl = h->lay.Nl, Np = h->lay.Np, H = d.H, L = d.L, P = d.P, R = d.R, J = d.J;
    const int ldp = 3 + P, ldq = 3 + R, ld1 = 2 * H + 2;
    const int E = t->E, Ec = t->Ec, Nm = h->lay.Nm;
    const Work& w = h->work;
    const ParamTable& tb = t->tab;
    const size_t NH = (size_t)N * H;
    const bool w3 = h->gemm_split;                                    // (three-piece weight gradients wherever the handle runs the split engine: cmdgen_wgrad_group)
    const bool sp = t->split_packs_valid && H == 256;                 // [.,256] x [256,256] data gradients on the bf16 matrix pipe:
    const int pcs = g_bf16 ? 1 : 3;                                   // three pieces per operand (fp32-accurate) or the leading one (bf16 operands)
    const bool tail_fused = sp && g_train_tune.dgrad_tail != 0;
    // weight / bias gradients leave the chain of data gradients for the second stream (SideStream above) where the buffers they read
    // rotate (the fused-tail path); ss.on = false: everything on the caller's stream, in the order written
    hipStream_t side = (bridged || (g_train_tune.wgrad_stream == 2 && t->ws_low)) ? t->ws_low : t->ws;
    SideStream ss{s, side, tail_fused && side != nullptr && g_train_tune.wgrad_stream != 0, &t->evs};
    // deferred side work: queued where the serial pass launches it, run on the side stream by flush_side() (serial pass: run at once)
    std::vector<std::function<void(hipStream_t)>> pending;
    auto defer = [&](std::function<void(hipStream_t)> f) { if (ss.on) pending.push_back(std::move(f)); else f(s); };
    auto flush_side = [&]() -> hipEvent_t {
        if (pending.empty()) return nullptr;
        hipStream_t q = ss.fork();
        for (auto& f : pending) f(q);
        pending.clear();
        return ss.mark();
    };
    // node-level weight (and bias) gradients of a block are collected and launched together (cmdgen_wgrad_group)
    WgradBatch wb; wb.n = 0;
    auto defer_wgrad = [&](const PRef& r, int col0, int in, const float* dy, const float* x, bool with_bias) {
        const int q = wb.n++;
        wb.dy[q] = dy; wb.x[q] = x; wb.dw[q] = grad + r.w + col0; wb.db[q] = (with_bias && r.has_bias) ? grad + r.b : nullptr;
        wb.M[q] = r.out; wb.N[q] = in; wb.lddy[q] = r.out; wb.ldx[q] = in; wb.ldw[q] = r.in;
    };
    const bool bf = g_bf16, gs = h->gemm_split;
    auto flush_wgrads = [&]() { const WgradBatch g = wb; defer([=](hipStream_t q) { cmdgen_wgrad_group(g, N, bf, q, gs); }); wb.n = 0; };
    // weight + bias gradient of one edge-level Linear (K = list length) in one launch
    auto edge_wgrad = [&](const PRef& r, const float* dy, const float* x, int K, int x_is_pre) {       // x_is_pre: x is the layer's PRE-activation (its SiLU is formed while the operand is staged)
        WgradBatch one; one.n = 1; one.xs[0] = x_is_pre;
        one.dy[0] = dy; one.x[0] = x; one.dw[0] = grad + r.w; one.db[0] = grad + r.b;
        one.M[0] = H; one.N[0] = H; one.lddy[0] = H; one.ldx[0] = H; one.ldw[0] = r.in;
        defer([=](hipStream_t q) { cmdgen_wgrad_group(one, K, bf, q, w3); });
    };
    // small weight + bias gradients of the readout / embedding stages (encoders, decoders, the two embeddings): a k_sgemm + k_colsum pair per
    // Linear.  Readout stage: with the side stream's first batch.  Embedding stage (the pass's tail): each pair on a third stream as soon as its
    // input exists.  (All of a stage's pairs as ONE kernel was built and is slower: profiles/r05_ar, removed in round 6.)
    SideStream ss3{s, t->ws_low, ss.on && t->ws_low != nullptr && side != t->ws_low, &t->evs2};      // a third stream for the embedding stage's small gradients
    bool tail_stage = false;
    auto small_wgrad = [&](const PRef& r, int in, int M, const float* dy, int lddy, const float* x, int ldx) {
        if (M <= 0) return;
        const PRef* rp = &r;
        if (tail_stage && ss3.on) {     // the pass's last stage: nothing follows to hide a serial tail - every pair starts as soon as its input exists, on a stream of its own
            hipStream_t q = ss3.fork();
            linear_wgrad(grad, *rp, 0, in, M, dy, lddy, x, ldx, q); tr_colsum(M, rp->out, dy, lddy, nullptr, grad + rp->b, 1, q);
            return;
        }
        defer([=](hipStream_t q) { g_bf16 = bf; linear_wgrad(grad, *rp, 0, in, M, dy, lddy, x, ldx, q); tr_colsum(M, rp->out, dy, lddy, nullptr, grad + rp->b, 1, q); });
    };
    // rotating buffers (side stream on): block k of the pass (k = 0 for block L-1) reads dL/dh_{l+1} in dhb[k % 3] and leaves dL/dh_l in
    // dhb[(k + 1) % 3]; dpre2 / dpre7 / dn / the two dP | dQ pairs by the parity of k.  Off: one buffer each, updated in place.
    float* dhb[3] = {t->dh, ss.on ? t->dh2 : t->dh, ss.on ? t->dh3 : t->dh};
    const size_t pq_off = (size_t)(t->dQ - t->dP);                    // dQ = dP + pq_off in every pair
    std::vector<hipEvent_t> blk_done((size_t)L * d.S + 1, nullptr);   // side work of GCL k of THIS call
    if (first_stage == 0) {
    // readout
    float* dh0 = dhb[0];
    const bool one_init = !d_eps_pocket && !d.joint;          // the conditional model: one launch (k_bwd_init) for the three fills and k_eps_bwd
    if (one_init) tr_bwd_init(Nl, N, P, d.dyn, d_eps_phar, t->dX, t->ddec, t->dhfin, s);
    else {
    HIPCHK(h, hipMemsetAsync(t->dX, 0, (size_t)N * 4 * sizeof(float), s));
    tr_eps_bwd(Nl, P, 0, d_eps_phar, t->dX, t->ddec, s);
    HIPCHK(h, hipMemsetAsync(t->dhfin, 0, (size_t)N * d.dyn * sizeof(float), s));
    }
    if (d_eps_pocket) {     // the pocket output exists in the loss (joint model): velocity rows Nl.. and the residue decoder
        tr_eps_bwd(Np, R, Nl, d_eps_pocket, t->dX, t->dqdec, s);
        small_wgrad(tb.rd2, 2 * R, Np, t->dqdec, R, t->qdeca, 2 * R);
        linear_dgrad(theta, tb.rd2, 0, 2 * R, Np, t->dqdec, R, t->dqdeca, 2 * R, false, s);
        tr_silu_bwd(t->dqdeca, t->qdec1, (size_t)Np * 2 * R, s);
        small_wgrad(tb.rd0, J, Np, t->dqdeca, 2 * R, t->hfin + (size_t)Nl * d.dyn, d.dyn);
        linear_dgrad(theta, tb.rd0, 0, J, Np, t->dqdeca, 2 * R, t->dhfin + (size_t)Nl * d.dyn, d.dyn, false, s);
    }
    if (d.joint) tr_center_per_sample(h->lay, t->dX, s);      // adjoint of the velocity's mean removal (a symmetric projection)
    if (!d.joint && Np && !one_init) HIPCHK(h, hipMemsetAsync(t->dX + (size_t)Nl * 4, 0, (size_t)Np * 4 * sizeof(float), s));   // pocket rows do not move
    small_wgrad(tb.pd2, 2 * P, Nl, t->ddec, P, t->deca, 2 * P);
    linear_dgrad(theta, tb.pd2, 0, 2 * P, Nl, t->ddec, P, t->ddeca, 2 * P, false, s);
    tr_silu_bwd(t->ddeca, t->dec1, (size_t)Nl * 2 * P, s);
    small_wgrad(tb.pd0, J, Nl, t->ddeca, 2 * P, t->hfin, d.dyn);
    linear_dgrad(theta, tb.pd0, 0, J, Nl, t->ddeca, 2 * P, t->dhfin, d.dyn, false, s);
    small_wgrad(tb.embo, H, N, t->dhfin, d.dyn, t->h + (size_t)L * d.S * NH, H);
    linear_dgrad(theta, tb.embo, 0, H, N, t->dhfin, d.dyn, dh0, H, false, s);
    flush_side();           // (none of what these read is written again in this pass)
    }
    const int S = d.S, U = L * S;
    const float* rowdiv = d.agg_mean ? w.adiv : nullptr;              // aggregation 'mean': sums were divided by the receiver's edge count (egnn_new.py:288-292)
    for (int l = L - 1; l >= 0; --l) {
        const int stage = L - l;
        if (stage < first_stage || stage > last_stage) continue;
        const int ulast = l * S + S - 1;                                  // the block's last GCL carries its EquivariantUpdate
        const ParamTable::Blk& bc = tb.blk[ulast];
        const TrainState::PackBlk& pkc = t->pack[ulast];
        const float4* Xl = t->X + (size_t)l * N;
        const float* pre6 = t->pre6 + (size_t)l * t->eccap * H; const float* pre7 = t->pre7 + (size_t)l * t->eccap * H;
        const float* phi = t->phi + (size_t)l * t->eccap;
        const size_t pq_floats = pq_off + NH;                             // dP and dQ, adjacent
        const bool pair = tail_fused;                         // the list's two reductions (head / gate partials, tail partials) as one launch
        for (int sub = S - 1; sub >= 0; --sub) {
        const int u = l * S + sub;
        const ParamTable::Blk& b = tb.blk[u];
        const TrainState::PackBlk& pk = t->pack[u];
        const float* hl = t->h + (size_t)u * NH;
        const float* hn = t->h + (size_t)(u + 1) * NH;
        const float* pre1 = t->pre1 + (size_t)u * t->ecap * H; const float* pre2 = t->pre2 + (size_t)u * t->ecap * H;
        const float* aggn = t->aggn + (size_t)u * NH; const float* pre3 = t->pre3 + (size_t)u * NH;
        const float* z = t->z + (size_t)u * t->ecap;
        const float* nact = t->nact + (size_t)u * NH;
        // this GCL's buffers: k counts the GCLs of the pass (k = 0 for the last GCL of block L-1)
        const int k = U - 1 - u, par = k & 1;
        float* dh_in = dhb[k % 3]; float* dh_out = dhb[(k + 1) % 3];      // dL/dh_{u+1} (complete after the coordinate model's part) -> dL/dh_u
        float* actA = ss.on && par ? t->actA2 : t->actA;                  // dpre2 [E]
        float* actB = ss.on && par ? t->actB2 : t->actB;                  // dpre7 [Ec]
        float* dn = ss.on && par ? t->dn2 : t->dn;
        float* dPc = ss.on ? t->dPx[par] : t->dP; float* dQc = dPc + pq_off;                    // coordinate list's dP | dQ
        float* dPe = ss.on && par ? t->dPx[2] : t->dP; float* dQe = dPe + pq_off;               // message list's
        float* part_c = ss.on ? t->part_x[par] : t->part_scratch; float* tail_c = ss.on ? t->tail_x[par] : t->tail_scratch;          // partial sums: coordinate list
        float* part_e = ss.on ? t->part_x[2 + par] : t->part_scratch; float* tail_e = ss.on ? t->tail_x[2 + par] : t->tail_scratch;  // ... message list
        // everything this GCL writes was last read by the side work of GCL k - 2
        if (k >= 2) ss.wait(blk_done[k - 2]);
        if (sub == S - 1) {
        // ---- EquivariantUpdate: x_{l+1} = x_l + acc / nf ; dX holds dL/dx_{l+1} and becomes dL/dx_l
        // (dL/d acc = dX / normalization_factor is formed where it is read; every later kernel of the block only adds to dX)
        // (its per-edge arithmetic runs inside tr_head_bwd: one launch)
        const CoordOutArgs co{w.crow, w.ccol, Xl, phi, d.use_tanh, d.coords_range, d.norm_constant, t->dX, d.norm_factor, rowdiv, t->dcd};
        // actB <- dpre7, d coord_mlp.4; also clears dP | dQ (hidden_nf is 64, 128 or 256: cmdgen_create)
        tr_head_bwd(Ec, H, t->dphi, theta + bc.c4.w, pre7, actB, pair ? part_c : t->tail_scratch, grad + bc.c4.w, dPc, pq_floats, s, pair, &co);
        edge_wgrad(bc.c2, actB, (t->wsilu & 2) ? pre6 : t->act6 + (size_t)l * t->eccap * H, Ec, (t->wsilu & 2) ? 1 : 0);      // weight and bias gradient of coord_mlp.2 (c1 = SiLU(pre6))
        if (tail_fused)     // dpre6 = (dpre7 W7) SiLU'(pre6) and everything done with it, in one kernel: it never reaches HBM
            cmdgen_dgrad_tail(Ec, actB, pkc.t_c2, pre6, w.crow, w.ccol, w.cd0, pair ? pkc.rd_c : theta + bc.c0.w + 2 * H /* radial column: the forward's contiguous copy */, pair ? 1 : ld1, Xl,
                              d.norm_constant, t->dcd, Nm, dPc, dQc, grad + bc.c0.w + 2 * H, t->dX, tail_c, pcs, s, pair);
        if (pair) { float* gw = grad + bc.c4.w; float* gc = grad + bc.c0.w + 2 * H; defer([=](hipStream_t q) { tr_reduce_pair(Ec, H, part_c, gw, nullptr, tail_c, gc, ld1, q); }); }
        else {
            if (sp) cmdgen_dgrad_split(Ec, actB, pkc.t_c2, nullptr, nullptr, t->actA, false, 1.0f, pre6, s, pcs);
            else linear_dgrad(theta, bc.c2, 0, H, Ec, actB, H, t->actA, H, false, s, pre6);  // actA <- dc1 * SiLU'(pre6) = dpre6
            // adjoints of the gathers, the radial / d0 column gradients, d radial and the geometry adjoint: one pass over dpre6
            tr_edge_tail_bwd(Ec, H, w.crow, w.ccol, t->actA, w.cd0, theta + bc.c0.w + 2 * H, ld1, Xl, d.norm_constant, t->dcd, Nm,
                             dPc, dQc, grad + bc.c0.w + 2 * H, t->dX, t->tail_scratch, s);
        }
        if (sp) cmdgen_dgrad_split(N, dPc, pkc.t_c0a, dQc, pkc.t_c0b, dh_in, true, 1.0f, nullptr, s, pcs);
        else {
            linear_dgrad(theta, bc.c0, 0, H, N, dPc, H, dh_in, H, true, s);
            linear_dgrad(theta, bc.c0, H, H, N, dQc, H, dh_in, H, true, s);
        }
        // weight / bias gradients of coord_mlp.0 (both halves): with node_mlp.2 below in one grouped launch, while dP, dQ and dh
        // still hold what they are the gradients of
        defer_wgrad(bc.c0, 0, H, dPc, hn, true);
        defer_wgrad(bc.c0, H, H, dQc, hn, false);
        }
        // ---- node model: h_{u+1} = h_u + W4 SiLU(W3 [h_u | aggn] + b3) + b4 ; dh_in holds dL/dh_{u+1}
        defer_wgrad(b.n2, 0, H, dh_in, nact, true);
        flush_wgrads();
        if (u == 0) flush_side();       // the last GCL of the pass: its side work starts as early as it can (nothing comes after to hide it)
        if (sp) {
            cmdgen_dgrad_split(N, dh_in, pk.t_n2, nullptr, nullptr, dn, false, 1.0f, pre3, s, pcs);
            cmdgen_dgrad_split(N, dn, pk.t_n0a, nullptr, nullptr, dh_out, true, 1.0f, nullptr, s, pcs,         // dh_out = dh_in + dpre3 W3[:, :H] and
                               pk.t_n0b, t->dagg, false, d.norm_factor, 0, dh_in, rowdiv);                      // dagg = dpre3 W3[:, H:] / nf: one launch
        } else {
            linear_dgrad(theta, b.n2, 0, H, N, dh_in, H, dn, H, false, s, pre3);       // dn <- dn1 * SiLU'(pre3) = dpre3
            linear_dgrad(theta, b.n0, 0, H, N, dn, H, dh_in, H, true, s);              // dh is now dL/dh_u (residual kept)
            linear_dgrad(theta, b.n0, H, H, N, dn, H, t->dagg, H, false, s);
            if (rowdiv) tr_scale_rows(t->dagg, rowdiv, H, NH, s); else tr_scale(t->dagg, d.norm_factor, NH, s);
        }
        // ---- edge model
        // actA <- dpre2, d att_mlp; also clears dP | dQ
        tr_gate_bwd(E, H, w.erow, pre2, d.attention ? theta + b.att.w : nullptr, z, d.attention, t->dagg, actA, pair ? part_e : t->tail_scratch,
                    d.attention ? grad + b.att.w : nullptr, d.attention ? grad + b.att.b : nullptr, dPe, pq_floats, s, pair);
        edge_wgrad(b.e2, actA, (t->wsilu & 1) ? pre1 : t->act1 + (size_t)u * t->ecap * H, E, (t->wsilu & 1) ? 1 : 0);            // weight and bias gradient of edge_mlp.2 (m1 = SiLU(pre1))
        if (tail_fused)
            cmdgen_dgrad_tail(E, actA, pk.t_e2, pre1, w.erow, w.ecol, w.ed0, pair ? pk.rd_e : theta + b.e0.w + 2 * H, pair ? 1 : ld1, Xl, d.norm_constant, nullptr, Nm,
                              dPe, dQe, grad + b.e0.w + 2 * H, t->dX, tail_e, pcs, s, pair);
        if (pair) {
            float* gaw = d.attention ? grad + b.att.w : nullptr; float* gab = d.attention ? grad + b.att.b : nullptr; float* ge = grad + b.e0.w + 2 * H;
            defer([=](hipStream_t q) { tr_reduce_pair(E, H, part_e, gaw, gab, tail_e, ge, ld1, q); });
        }
        else {
            if (sp) cmdgen_dgrad_split(E, actA, pk.t_e2, nullptr, nullptr, t->actB, false, 1.0f, pre1, s, pcs);
            else linear_dgrad(theta, b.e2, 0, H, E, actA, H, t->actB, H, false, s, pre1);   // actB <- dm1 * SiLU'(pre1) = dpre1
            tr_edge_tail_bwd(E, H, w.erow, w.ecol, t->actB, w.ed0, theta + b.e0.w + 2 * H, ld1, Xl, d.norm_constant, nullptr, Nm,
                             dPe, dQe, grad + b.e0.w + 2 * H, t->dX, t->tail_scratch, s);
        }
        // node_mlp.0 (both halves) and edge_mlp.0 (both halves): the second grouped launch of the GCL
        defer_wgrad(b.n0, 0, H, dn, hl, true);
        defer_wgrad(b.n0, H, H, dn, aggn, false);
        defer_wgrad(b.e0, 0, H, dPe, hl, true);
        defer_wgrad(b.e0, H, H, dQe, hl, false);
        flush_wgrads();
        blk_done[k] = flush_side();                 // the GCL's weight gradients: one fork
        if (sp) cmdgen_dgrad_split(N, dPe, pk.t_e0a, dQe, pk.t_e0b, dh_out, true, 1.0f, nullptr, s, pcs);       // (dh_out = dh_in when the side stream is off)
        else {
            linear_dgrad(theta, b.e0, 0, H, N, dPe, H, dh_out, H, true, s);
            linear_dgrad(theta, b.e0, H, H, N, dQe, H, dh_out, H, true, s);
        }
        }
    }
    if (last_stage < L + 1) {
        ss.join();
        if (bridged) { HIPCHK(h, hipEventRecord(t->br_out, s)); HIPCHK(h, hipStreamWaitEvent(caller, t->br_out, 0)); }
        HIPCHK(h, hipGetLastError());
        return CMDGEN_OK;
    }
    // embedding and encoders
    tail_stage = true;
    float* dhE = dhb[(L * d.S) % 3];
    small_wgrad(tb.emb, d.dyn, N, dhE, H, t->hdyn, d.dyn);
    linear_dgrad(theta, tb.emb, 0, d.dyn, N, dhE, H, t->dhdyn, d.dyn, false, s);
    small_wgrad(tb.pe2, 2 * P, Nl, t->dhdyn, d.dyn, t->enca_l, 2 * P);
    linear_dgrad(theta, tb.pe2, 0, 2 * P, Nl, t->dhdyn, d.dyn, t->denca_l, 2 * P, false, s);
    tr_silu_bwd(t->denca_l, t->enc1_l, (size_t)Nl * 2 * P, s);
    small_wgrad(tb.pe0, P, Nl, t->denca_l, 2 * P, t->xh_phar + 3, ldp);
    const float* dq = t->dhdyn + (size_t)Nl * d.dyn;
    small_wgrad(tb.re2, 2 * R, Np, dq, d.dyn, t->enca_p, 2 * R);
    linear_dgrad(theta, tb.re2, 0, 2 * R, Np, dq, d.dyn, t->denca_p, 2 * R, false, s);
    tr_silu_bwd(t->denca_p, t->enc1_p, (size_t)Np * 2 * R, s);
    small_wgrad(tb.re0, R, Np, t->denca_p, 2 * R, t->xh_pocket + 3, ldq);
    flush_side();
    ss3.join();
    ss.join();
    if (bridged) {
        HIPCHK(h, hipEventRecord(t->br_out, s));
        HIPCHK(h, hipStreamWaitEvent(caller, t->br_out, 0));
    }
    HIPCHK(h, hipGetLastError());
    return CMDGEN_OK;
}

extern "C" int cmdgen_train_noise(cmdgen_handle* h, const float* phar_x, const float* phar_one_hot, const float* pocket_x,
                                  const float* pocket_one_hot, const float* tab, const float* eps, float* z_t, float* xh_pocket,
                                  float* kl_sums, cmdgen_stream stream) {
    if (!h) return CMDGEN_EINVAL;
    if (!h->have_layout) return fail(h, CMDGEN_ESTATE, "no batch layout (cmdgen_set_layout)");
    if (h->dims.joint) return fail(h, CMDGEN_ESTATE, "cmdgen_train_noise is the conditional model's noising");
    if (!phar_x || !phar_one_hot || !pocket_x || !pocket_one_hot || !tab || !eps || !z_t || !xh_pocket || !kl_sums)
        return fail(h, CMDGEN_EINVAL, "null device pointer");
    hipSetDevice(h->device);
    h->last_stream = (hipStream_t)stream;
    tr_noise(h->lay, h->dims, phar_x, phar_one_hot, pocket_x, pocket_one_hot, tab, eps, z_t, xh_pocket, kl_sums, (hipStream_t)stream);
    HIPCHK(h, hipGetLastError());
    return CMDGEN_OK;
}

extern "C" int cmdgen_train_loss(cmdgen_handle* h, int32_t l2, float T, const float* net_out, const float* eps, const float* z_t,
                                 const float* phar_one_hot, const float* tab, const float* kl_sums, float* terms, float* d_eps,
                                 float* means, cmdgen_stream stream) {
    if (!h) return CMDGEN_EINVAL;
    if (!h->have_layout) return fail(h, CMDGEN_ESTATE, "no batch layout (cmdgen_set_layout)");
    if (h->dims.joint) return fail(h, CMDGEN_ESTATE, "cmdgen_train_loss is the conditional model's loss");
    if (!net_out || !eps || !z_t || !phar_one_hot || !tab || !kl_sums || !terms || !d_eps || !means)
        return fail(h, CMDGEN_EINVAL, "null device pointer");
    hipSetDevice(h->device);
    h->last_stream = (hipStream_t)stream;
    tr_loss(h->lay, h->dims, l2, T, net_out, eps, z_t, phar_one_hot, tab, kl_sums, terms, d_eps, means, (hipStream_t)stream);
    HIPCHK(h, hipGetLastError());
    return CMDGEN_OK;
}

extern "C" int cmdgen_train_noise_joint(cmdgen_handle* h, const float* phar_x, const float* phar_one_hot, const float* pocket_x,
                                        const float* pocket_one_hot, const float* tab, const float* draw_phar, const float* draw_pocket,
                                        float* z_phar, float* z_pocket, float* eps_phar, float* eps_pocket, float* kl_sums,
                                        cmdgen_stream stream) {
    if (!h) return CMDGEN_EINVAL;
    if (!h->have_layout) return fail(h, CMDGEN_ESTATE, "no batch layout (cmdgen_set_layout)");
    if (!h->dims.joint) return fail(h, CMDGEN_ESTATE, "cmdgen_train_noise_joint is the joint model's noising");
    if (!phar_x || !phar_one_hot || !pocket_x || !pocket_one_hot || !tab || !draw_phar || !draw_pocket || !z_phar || !z_pocket ||
        !eps_phar || !eps_pocket || !kl_sums)
        return fail(h, CMDGEN_EINVAL, "null device pointer");
    hipSetDevice(h->device);
    h->last_stream = (hipStream_t)stream;
    tr_noise_joint(h->lay, h->dims, phar_x, phar_one_hot, pocket_x, pocket_one_hot, tab, draw_phar, draw_pocket, z_phar, z_pocket, eps_phar,
                   eps_pocket, kl_sums, (hipStream_t)stream);
    HIPCHK(h, hipGetLastError());
    return CMDGEN_OK;
}

extern "C" int cmdgen_train_loss_joint(cmdgen_handle* h, int32_t l2, float T, const float* net_phar, const float* net_pocket,
                                       const float* eps_phar, const float* eps_pocket, const float* z_phar, const float* z_pocket,
                                       const float* phar_one_hot, const float* pocket_one_hot, const float* tab, const float* kl_sums,
                                       float* terms, float* d_eps_phar, float* d_eps_pocket, float* means, cmdgen_stream stream) {
    if (!h) return CMDGEN_EINVAL;
    if (!h->have_layout) return fail(h, CMDGEN_ESTATE, "no batch layout (cmdgen_set_layout)");
    if (!h->dims.joint) return fail(h, CMDGEN_ESTATE, "cmdgen_train_loss_joint is the joint model's loss");
    if (!net_phar || !net_pocket || !eps_phar || !eps_pocket || !z_phar || !z_pocket || !phar_one_hot || !pocket_one_hot || !tab ||
        !kl_sums || !terms || !d_eps_phar || !d_eps_pocket || !means)
        return fail(h, CMDGEN_EINVAL, "null device pointer");
    hipSetDevice(h->device);
    h->last_stream = (hipStream_t)stream;
    tr_loss_joint(h->lay, h->dims, l2, T, net_phar, net_pocket, eps_phar, eps_pocket, z_phar, z_pocket, phar_one_hot, pocket_one_hot, tab,
                  kl_sums, terms, d_eps_phar, d_eps_pocket, means, (hipStream_t)stream);
    HIPCHK(h, hipGetLastError());
    return CMDGEN_OK;
}

extern "C" int cmdgen_train_set_precision(cmdgen_handle* h, int32_t bf16_gemm) {
    if (!h) return CMDGEN_EINVAL;
    if (!h->have_layout) return fail(h, CMDGEN_ESTATE, "no batch layout (cmdgen_set_layout)");
    hipSetDevice(h->device);
    int rc = ensure_state(h); if (rc) return rc;
    h->train->bf16 = bf16_gemm != 0;
    h->train_bf16 = bf16_gemm != 0;
    return CMDGEN_OK;
}

extern "C" int cmdgen_grad_sqnorm(cmdgen_handle* h, const float* grad, int64_t n, float* out_host, cmdgen_stream stream) {
    if (!h || !grad || !out_host || n < 1) return fail(h, CMDGEN_EINVAL, "bad arguments");
    hipSetDevice(h->device);
    int rc = ensure_state(h); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    h->last_stream = s;
    HIPCHK(h, hipMemsetAsync(h->train->d_scalar, 0, sizeof(float), s));
    tr_sqsum((size_t)n, grad, h->train->d_scalar, s);
    HIPCHK(h, hipMemcpyAsync(out_host, h->train->d_scalar, sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return CMDGEN_OK;
}

extern "C" int cmdgen_adamw_step(cmdgen_handle* h, float* theta, const float* grad, float* exp_avg, float* exp_avg_sq,
                                 float* max_exp_avg_sq, int64_t n, int64_t step, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, float clip_coef, cmdgen_stream stream) {
    if (!h || !theta || !grad || !exp_avg || !exp_avg_sq || !max_exp_avg_sq || n < 1 || step < 1) return fail(h, CMDGEN_EINVAL, "bad arguments");
    hipSetDevice(h->device);
    const float bias1 = 1.0f - powf(beta1, (float)step);
    const float bias2 = 1.0f - powf(beta2, (float)step);
    tr_adamw((size_t)n, theta, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, lr, beta1, beta2, eps, weight_decay, bias1,
             sqrtf(bias2), clip_coef, (hipStream_t)stream);
    HIPCHK(h, hipGetLastError());
    return CMDGEN_OK;
}

extern "C" int cmdgen_last_grad_norm(cmdgen_handle* h, float* grad_norm_host);
extern "C" int cmdgen_adamw_step_clipped(cmdgen_handle* h, float* theta, const float* grad, float* exp_avg, float* exp_avg_sq,
                                         float* max_exp_avg_sq, int64_t n, int64_t step, float lr, float beta1, float beta2,
                                         float eps, float weight_decay, float max_grad_norm, float* grad_norm_host,
                                         cmdgen_stream stream) {
    if (!h || !theta || !grad || !exp_avg || !exp_avg_sq || !max_exp_avg_sq || n < 1 || step < 1)
        return fail(h, CMDGEN_EINVAL, "bad arguments");
    hipSetDevice(h->device);
    int rc = ensure_state(h); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    h->last_stream = s;
    float* sq = h->train->d_scalar;
    HIPCHK(h, hipMemsetAsync(sq, 0, sizeof(float), s));
    tr_sqsum((size_t)n, grad, sq, s);
    const float bias1 = 1.0f - powf(beta1, (float)step), bias2 = 1.0f - powf(beta2, (float)step);
    // the update is queued behind the norm without a host round trip: the clipping coefficient is formed on the device
    // (after a half-engine forward a non-finite norm skips the update on the device: see k_adamw)
    const bool guard = h->train && h->train->fwd_on_half;
    if (guard) tr_norm_guard(sq, (const int*)h->work.nan_flag, s);
    tr_adamw((size_t)n, theta, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, lr, beta1, beta2, eps, weight_decay, bias1, sqrtf(bias2), 1.0f, s,
             (max_grad_norm > 0.f || guard) ? sq : nullptr, max_grad_norm, guard ? 1 : 0);
    // (the readback state lives in the handle: a later cmdgen_set_layout that outgrows the workspaces frees the training state)
    if (!h->h_norm) { HIPCHK(h, hipHostMalloc((void**)&h->h_norm, sizeof(float), hipHostMallocDefault)); HIPCHK(h, hipEventCreateWithFlags(&h->norm_ev, hipEventDisableTiming)); }
    HIPCHK(h, hipMemcpyAsync(h->h_norm, sq, sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipEventRecord(h->norm_ev, s));
    h->norm_pending = true;
    if (!grad_norm_host) return CMDGEN_OK;          // deferred: cmdgen_last_grad_norm collects it (the host keeps queueing the next step)
    return cmdgen_last_grad_norm(h, grad_norm_host);
}

extern "C" int cmdgen_last_grad_norm(cmdgen_handle* h, float* grad_norm_host) {
    if (!h || !grad_norm_host) return CMDGEN_EINVAL;
    if (!h->norm_pending) return fail(h, CMDGEN_ESTATE, "no gradient norm outstanding (cmdgen_adamw_step_clipped)");
    hipSetDevice(h->device);
    HIPCHK(h, hipEventSynchronize(h->norm_ev));
    *grad_norm_host = sqrtf(*h->h_norm);
    h->norm_pending = false;
    return CMDGEN_OK;
}

// C[M,N] = A op B through the training GEMM (test aid)
extern "C" int cmdgen_debug_sgemm(cmdgen_handle* h, int32_t ta, int32_t tb, int32_t M, int32_t N, int32_t K, const float* A,
                                  int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc, const float* bias,
                                  int32_t accumulate, int32_t split_k, cmdgen_stream stream) {
    if (!h || !A || !B || !C) return fail(h, CMDGEN_EINVAL, "null pointer");
    hipSetDevice(h->device);
    cmdgen_sgemm(ta != 0, tb != 0, M, N, K, A, lda, B, ldb, C, ldc, bias, 1.0f, (accumulate & 1) != 0, split_k, (hipStream_t)stream,
                 0, nullptr, 0, (accumulate & 2) != 0);
    HIPCHK(h, hipGetLastError());
    return CMDGEN_OK;
}

// Y[M,256] (+)= (A0 W0[:, :256] + A1 W1[:, :256]) / div * SiLU'(pre) through the training step's data-gradient kernel (test aid):
// W0 / W1 dev [256][256] row-major nn.Linear weights (dX = dY W), packed here as the step packs them.  tile_rows 0 = the
// launcher's choice, 32 / 64 forces the tile; pieces 3 (fp32-accurate) or 1 (bf16 operands).
extern "C" int cmdgen_debug_dgrad(cmdgen_handle* h, int32_t M, const float* A0, const float* W0, const float* A1, const float* W1,
                                  float* Y, int32_t accumulate, float div, const float* pre, int32_t pieces, int32_t tile_rows,
                                  cmdgen_stream stream) {
    if (!h || !A0 || !W0 || !Y || (A1 && !W1) || M < 1 || (pieces != 1 && pieces != 3)) return fail(h, CMDGEN_EINVAL, "bad arguments");
    hipSetDevice(h->device);
    g_train_tune = h->tune;
    hipStream_t s = (hipStream_t)stream;
    void* packs = nullptr; void* tab = nullptr;
    const size_t pack_bytes = (size_t)256 * 256 * 6;
    HIPCHK(h, hipMalloc(&packs, 2 * pack_bytes));
    HIPCHK(h, hipMalloc(&tab, 2 * sizeof(RepackSplitT)));
    // offsets are relative to W0: the second matrix is addressed through the pointer difference (both are device pointers)
    const RepackSplitT t2[2] = {{0, 256, packs, 1}, {(int)(A1 ? W1 - W0 : 0), 256, (char*)packs + pack_bytes, 1}};
    HIPCHK(h, hipMemcpyAsync(tab, t2, sizeof t2, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));
    tr_repack_split_t(W0, tab, A1 ? 2 : 1, s);
    cmdgen_dgrad_split(M, A0, packs, A1, A1 ? (char*)packs + pack_bytes : nullptr, Y, accumulate != 0, div, pre, s, pieces, nullptr, nullptr,
                       false, 1.0f, tile_rows == 64 ? 64 : (tile_rows ? 32 : 0));
    hipError_t e = hipStreamSynchronize(s);
    hipFree(packs); hipFree(tab);
    HIPCHK(h, e);
    HIPCHK(h, hipGetLastError());
    return CMDGEN_OK;
}

// dW[M][N] += dY^T X, db[M] += column sums of dY, through the training step's grouped weight-gradient launch (test aid).
// mode 0: fp32 instruction (k_wgrad_group<false>); 1: bf16 operands (k_wgrad_split<1> where the shape allows, else k_wgrad_group<true>);
// 3: three-piece split (k_wgrad_split128<3> / k_wgrad_split<3> where the shape allows, else as mode 0).
extern "C" int cmdgen_debug_wgrad(cmdgen_handle* h, int32_t K, int32_t M, int32_t N, const float* dY, const float* X, float* dW,
                                  float* db, int32_t mode, cmdgen_stream stream) {
    if (!h || !dY || !X || !dW || K < 1 || M < 1 || N < 1) return fail(h, CMDGEN_EINVAL, "bad arguments");
    hipSetDevice(h->device);
    g_train_tune = h->tune;
    WgradBatch one; one.n = 1;
    one.dy[0] = dY; one.x[0] = X; one.dw[0] = dW; one.db[0] = db;
    one.M[0] = M; one.N[0] = N; one.lddy[0] = M; one.ldx[0] = N; one.ldw[0] = N;
    cmdgen_wgrad_group(one, K, mode == 1, (hipStream_t)stream, mode == 3, mode == 3);
    HIPCHK(h, hipGetLastError());
    return CMDGEN_OK;
}
