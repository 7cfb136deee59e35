/*
 * cmdgen_hip.h - C ABI of libcmdgen_hip.so: the MI355X (gfx950) implementation of
 * DiffPhar's pocket-conditioned denoising path.
 *
 * The reference (zyrlia1018/CMD-GEN, DiffPhar/) is pure Python and has no FFI of its
 * own; the entry points below are what a binding for this path replaces, cited as
 * file:line relative to /root/reference/DiffPhar.  INTEGRATION.md shows the ctypes
 * stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  "dev" pointers are device memory
 *     owned by the CALLER (e.g. torch tensors' data_ptr()); "host" pointers are host
 *     memory.  The handle owns packed weights, the schedule table, workspaces and
 *     hipGraph executables, nothing else.
 *   - Every launch goes to the caller-supplied stream; no hidden device synchronise
 *     except where a function is documented to return host-visible results.
 *   - Return 0 on success, a negative CMDGEN_E* code otherwise; cmdgen_last_error()
 *     gives the message.  Nothing throws across the ABI.
 *   - A handle is bound to one device and is not thread-safe (one handle per GPU /
 *     host thread), like the reference's module objects.
 *   - Flat, un-padded node lists (PyG style): all samples' nodes concatenated along
 *     dim 0, sample membership given by per-sample counts; masks must be ascending
 *     and contiguous, as every mask the reference builds is (utils.py:137-145,
 *     dataset.py:59-60, lightning_modules.py:445-448).
 *   - All floating point is fp32 (constants.py:8).
 */
#ifndef CMDGEN_HIP_H
#define CMDGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMDGEN_OK            0
#define CMDGEN_EINVAL       -1   /* bad argument / unsupported configuration */
#define CMDGEN_ESTATE       -2   /* call order (weights not finalised, no layout, ...) */
#define CMDGEN_EHIP         -3   /* a HIP runtime call failed */
#define CMDGEN_ENOMEM       -4

typedef struct cmdgen_handle cmdgen_handle;
typedef void* cmdgen_stream;     /* hipStream_t */

/* Hyper-parameters that fix every shape on the path
 * (EGNNDynamics.__init__ dynamics.py:10-73, EGNN.__init__ egnn_new.py:160-191,
 *  EnVariationalDiffusion.__init__ en_diffusion.py:18-62; values of
 *  configs/crossdocked_ca_cond.yml:22-41 in comments). */
typedef struct cmdgen_config {
    int32_t phar_nf;               /* 8  */
    int32_t residue_nf;            /* 20 (CA) or 11 (full-atom) */
    int32_t joint_nf;              /* 32 */
    int32_t hidden_nf;             /* 256; one of 64, 128, 256, 512 (512: sampling only, fp32-image tiles; the plane kernels are 256 only) */
    int32_t n_layers;              /* 5  */
    int32_t inv_sublayers;         /* 1; GCLs per EquivariantBlock (egnn_new.py:127-131): >= 1 for sampling (dead-work skipping is off above 1),
                                      the training step supports 1 only */
    int32_t attention;             /* 1  */
    int32_t tanh;                  /* 1  */
    int32_t condition_time;        /* 1  */
    int32_t timesteps;             /* T of the gamma table (500) */
    int32_t no_com_projection;     /* 0; 1 = SimpleConditionalDDPM (conditional_model.py:481-525): pocket centred once,
                                      no centre-of-mass projection of the samples */
    int32_t update_pocket_coords;  /* 0; 1 = mode 'joint' (lightning_modules.py:125, dynamics.py:105-107, :133-136):
                                      pocket nodes move too and the velocity's centre of mass is removed;
                                      required by cmdgen_joint_chain */
    float   edge_cutoff;           /* 6.0; < 0 means no cutoff (complete graph per sample) */
    float   norm_constant;         /* 1  */
    float   normalization_factor;  /* 100 (aggregation_method 'sum': segment sums are divided by it, egnn_new.py:285-286) */
    float   coords_range;          /* 15 (egnn_new.py:161; quirk: never divided by n_layers) */
    float   norm_x;                /* norm_values[0] = 1 */
    float   norm_h;                /* norm_values[1] = 4 */
    float   bias_h;                /* norm_biases[1] = 0 */
    int32_t aggregation_mean;      /* 0 = aggregation_method 'sum'; 1 = 'mean' (egnn_new.py:288-292: every segment sum is divided by its
                                      receiver's edge count instead of normalization_factor); sampling only */
    int32_t sin_embedding;         /* 0; 1 = SinusoidsEmbeddingNew on both distance features of an edge (egnn_new.py:174-176, :249-260: the
                                      first layer of the edge / coordinate MLPs has 2H + 24 inputs); sampling only, on the fp32 matrix
                                      instruction (the plane kernels of the split engine carry the two scalar features only) */
} cmdgen_config;

/* Work counters accumulated on the device since the last reset (for the
 * algorithmic-FLOP roofline). */
typedef struct cmdgen_counters {
    uint64_t evaluations;          /* network evaluations (EGNNDynamics.forward calls) */
    uint64_t edges;                /* sum over evaluations of directed edges incl. self loops */
    uint64_t edges_phar;           /* ... of those the coordinate update runs on: receiver is a pharmacophore node and
                                      the edge is not a self loop (a self loop's coord_diff is exactly 0) */
    uint64_t nodes;                /* sum over evaluations of nodes */
    uint64_t nan_resets;           /* evaluations whose velocity was reset (dynamics.py:129-131) */
    uint64_t edges_skipped;        /* edges of k_edge_msg tiles that were skipped as dead work (last block of a conditional evaluation: no receiver
                                      of the tile is still read), summed over launches: the executed edge work is edges * n_layers - this */
    uint64_t node_rows_skipped;    /* rows of k_node tiles skipped for the same reason, summed over launches */
    uint64_t reserved[1];
} cmdgen_counters;

/* Per-kernel timing of one profiled evaluation (hipEvent pairs on the launch stream). */
typedef struct cmdgen_kernel_times {
    float edge_build_ms, embed_ms, edge_msg_ms, node_ms, edge_coord_ms, readout_ms, ddpm_ms;
    int32_t edge_msg_launches, node_launches, edge_coord_launches;
} cmdgen_kernel_times;

/* ---- lifetime --------------------------------------------------------------------- */
/* Replaces constructing EGNNDynamics + ConditionalDDPM (lightning_modules.py:110-139). */
int  cmdgen_create(const cmdgen_config* cfg, int device, cmdgen_handle** out);
void cmdgen_destroy(cmdgen_handle* h);
const char* cmdgen_last_error(const cmdgen_handle* h);   /* h may be NULL: last create error */
const char* cmdgen_version(void);

/* ---- parameters ------------------------------------------------------------------- */
/* Replaces nn.Module.load_state_dict for the 'ddpm.' sub-tree of a Lightning checkpoint
 * (generate_phars.py:32).  `name` is the reference key below 'ddpm.', e.g.
 * "dynamics.egnn.e_block_0.gcl_0.edge_mlp.0.weight" or "gamma.gamma"; `host` holds `n`
 * fp32 values in nn.Linear layout [out, in] row-major. */
int cmdgen_load_weights(cmdgen_handle* h, const char* name, const float* host, size_t n);
/* Checks that every tensor arrived, packs them into MFMA fragment order on the device. */
int cmdgen_finalize_weights(cmdgen_handle* h);

/* ---- batch layout ----------------------------------------------------------------- */
/* Per-sample node counts of the flat batch (host arrays of length `batch`):
 * num_phar = num_nodes_phar, num_pocket = pocket['size'] (conditional_model.py:388-408).
 * Sizes workspaces; cheap when the layout is unchanged. */
int cmdgen_set_layout(cmdgen_handle* h, int64_t batch,
                      const int64_t* num_phar_host, const int64_t* num_pocket_host);
/* Ordering contract: the call rewrites index arrays that kernels of the PREVIOUS layout read.  Before touching
 * them it waits for the stream most recently passed to this handle (and its internal stream); work the caller
 * queued for this handle on any OTHER stream must be complete before calling.  Limits: a sample's nodes are kept
 * in LDS by the neighbour search (24 B per node; at most ~6500 nodes per sample), the dense edge bound
 * sum(n_b^2) must fit int32. */
/* The same, ordered on `stream` instead of waited for: when the new layout fits the workspaces already allocated (every
 * training step has its own ragged layout of about the same size) and `stream` is the stream this handle was last given,
 * the index arrays are written to the second of two device blocks from pinned staging with a stream-ordered copy, so
 * kernels of the previous layout still running on `stream` are not disturbed and the host does not wait.  Otherwise it
 * behaves as cmdgen_set_layout (and also waits for `stream`). */
int cmdgen_set_layout_on_stream(cmdgen_handle* h, int64_t batch, const int64_t* num_phar_host,
                                const int64_t* num_pocket_host, cmdgen_stream stream);

/* ---- one network evaluation ------------------------------------------------------- */
/* EGNNDynamics.forward (dynamics.py:75-139); conditional mode, or joint mode when
 * config.update_pocket_coords = 1 (then eps_pocket is required: its x columns are the pocket velocity).
 *   xh_phar   dev [Nl, 3+phar_nf]      xh_pocket dev [Np, 3+residue_nf]
 *   t         dev [batch]              (the reference's [B,1]; a single-sample batch uses t[0])
 *   eps_phar  dev [Nl, 3+phar_nf]  out
 *   eps_pocket dev [Np, 3+residue_nf] out, may be NULL (every conditional caller discards it:
 *              conditional_model.py:115, :246, :355) */
int cmdgen_dynamics_forward(cmdgen_handle* h, const float* xh_phar, const float* xh_pocket,
                            const float* t, float* eps_phar, float* eps_pocket,
                            cmdgen_stream stream);

/* Radius graph of the last evaluation, EGNNDynamics.get_edges (dynamics.py:141-147):
 * copies up to `cap` edges as (row, col) int32 pairs in the reference's flat node
 * numbering (phar nodes first) to HOST arrays; returns the edge count in *n_edges.
 * Synchronises the stream.  Debug / parity aid. */
int cmdgen_get_edges(cmdgen_handle* h, int32_t* row_host, int32_t* col_host, int64_t cap,
                     int64_t* n_edges, cmdgen_stream stream);

/* EGNNDynamics.get_edges(batch_mask, x) (dynamics.py:141-147) for ARBITRARY coordinates, independent of the
 * handle's batch layout: x dev [sum counts, 3] holds `batch` samples back to back (ascending mask), counts_host
 * their sizes.  Writes the edges (same sample, ||x_i - x_j|| <= edge_cutoff, self loops kept) sorted by (row, col)
 * to the DEVICE arrays row_dev / col_dev (int32, capacity `cap` >= sum counts^2, else CMDGEN_EINVAL) and the
 * count to *n_edges.  Synchronises the stream. */
int cmdgen_radius_graph(cmdgen_handle* h, const float* x, const int64_t* counts_host, int64_t batch,
                        int32_t* row_dev, int32_t* col_dev, int64_t cap, int64_t* n_edges, cmdgen_stream stream);

/* Parity aid: run one evaluation only up to a point, so that intermediates the fused kernels never keep can be
 * read with cmdgen_debug_read: all of blocks 0..block-1, then of block `block` stage 1 = after the edge-message
 * kernel ("agg" holds the un-normalised segment sums of e_ij, egnn_new.py:50-52), 2 = after the node kernel ("h"
 * is the block's output, "agg" is zero again), 3 = after the coordinate kernel ("acc" row `block` holds the sums
 * of trans, egnn_new.py:91-98).  Leaves the workspace unusable for a chain until the next full evaluation. */
int cmdgen_debug_eval_prefix(cmdgen_handle* h, const float* xh_phar, const float* xh_pocket, const float* t,
                             int32_t block, int32_t stage, cmdgen_stream stream);

/* Copy an internal activation of the last evaluation to host (parity aid): what = "h" | "agg" | "P" | "Q" [N, hidden],
 * "x0" [Nm, 4], "xl" / "acc" [n_layers, Nm, 4].  After a CONDITIONAL evaluation whose pocket output was not requested
 * (eps_pocket == NULL, every chain) the POCKET rows of h / P / Q are undefined unless option "dead_skip" is 0: blocks skip
 * tiles whose result nobody reads, so those rows hold a mix of earlier blocks.  Phar rows, agg (zero) and the positions are
 * always complete; cmdgen_debug_eval_prefix never skips. */
int cmdgen_debug_read(cmdgen_handle* h, const char* what, float* host, size_t n, cmdgen_stream stream);

/* Fills out_dev[n_nodes * width] with the standard normals the sampler would draw for draw index
 * `draw`, global pocket id `pocket_id` (Philox4x32-10 + Box-Muller; the production noise source of
 * cmdgen_sample_chain when `noise` is NULL).  Test aid for the statistical checks. */
int cmdgen_debug_noise(cmdgen_handle* h, uint64_t seed, int64_t pocket_id, int32_t draw, int32_t n_nodes,
                       int32_t width, float* out_dev, cmdgen_stream stream);

/* ---- the denoising loop ----------------------------------------------------------- */
/* ConditionalDDPM.sample_given_pocket (conditional_model.py:388-465) with return_frames=1:
 * init noise around the pocket COM, `timesteps` posterior steps (sample_p_zs_given_zt
 * :342-374), final p(x,h|z0) decode (:108-131), CoG drift fix (:451-457).
 *   pocket_x      dev [Np, 3]           un-normalised coordinates
 *   pocket_onehot dev [Np, residue_nf]  one-hot (un-normalised; scaled by 1/norm_h inside)
 *   timesteps     K <= T, strided schedule t=(s+1)/K as the reference
 *   noise         dev [K+2, Nl, 3+phar_nf] Gaussian draws in the reference's draw order, or
 *                 NULL to draw on the device (Philox4x32-10 keyed by seed, global pocket id,
 *                 draw index, node) - the reference draws with torch.randn on the compute
 *                 device (en_diffusion.py:946-949), which no other device can reproduce.
 *   pocket_ids_host  host [batch] global pocket indices for the Philox key (NULL: 0..batch-1);
 *                 makes results independent of how pockets are sharded over GPUs.
 *   xh_phar_out   dev [Nl, 3+phar_nf]    x in Angstrom, h one-hot (as floats)
 *   xh_pocket_out dev [Np, 3+residue_nf] translated pocket, h = one_hot
 *   z_steps_out   dev [K, Nl, 3+phar_nf] z after each posterior step, or NULL
 *   pocket_steps_out dev [K, Np, 3] translated pocket coordinates after each step (normalised space), or NULL
 *                 (both feed return_frames > 1, conditional_model.py:439-442)
 *   use_graph     1: replay the step as a hipGraph, 0: eager launches
 * Asynchronous on `stream`; call cmdgen_chain_status afterwards for the deferred checks. */
int cmdgen_sample_chain(cmdgen_handle* h, const float* pocket_x, const float* pocket_onehot,
                        int32_t timesteps, const float* noise, uint64_t seed,
                        const int64_t* pocket_ids_host,
                        float* xh_phar_out, float* xh_pocket_out, float* z_steps_out,
                        float* pocket_steps_out, int32_t use_graph, cmdgen_stream stream);

/* The JOINT model's loops (config.update_pocket_coords = 1), return_frames=1:
 *   phar_fixed == NULL && pocket_fixed == NULL: EnVariationalDiffusion.sample (en_diffusion.py:576-647) -
 *       phar AND pocket nodes start from noise; phar_x/phar_onehot/pocket_x/pocket_onehot are ignored (may be NULL).
 *   otherwise: EnVariationalDiffusion.inpaint (en_diffusion.py:672-831), RePaint with the schedule of
 *       get_repaint_schedule(resamplings, jump_length, timesteps) (:649-670).  generate_phars' inpainting
 *       branch (lightning_modules.py:466-486) passes phar_x = 0, phar_onehot = 0, phar_fixed = 0, pocket_fixed = 1.
 *   phar_x dev [Nl,3], phar_onehot dev [Nl,phar_nf], pocket_x dev [Np,3], pocket_onehot dev [Np,residue_nf]:
 *       used RAW - the reference's inpaint never calls normalize (quirk kept);
 *   phar_fixed dev [Nl], pocket_fixed dev [Np]: 1.0 = known node, 0.0 = node to generate;
 *   noise  dev [n_draws][Nl*(3+phar_nf) + Np*(3+residue_nf)] or NULL (Philox on the device).  One row per
 *       sample_combined_position_feature_noise call (:555-574) in the reference's call order: the phar block
 *       [Nl,3+phar_nf] then the pocket block [Np,3+residue_nf]; x columns hold the raw draw (the centre-of-mass
 *       projection is applied inside).  Draws: 1 (z_T) + per denoising step [1 if inpainting: known part] + 1
 *       + [1 if the step is followed by a jump back] + 1 (final decode);
 *   n_draws  rows available in `noise` (checked against the schedule; ignored when noise == NULL);
 *   xh_phar_out dev [Nl,3+phar_nf], xh_pocket_out dev [Np,3+residue_nf]: x, one-hot h (floats);
 *   z_steps_out dev [n_steps][Nl*(3+phar_nf) + Np*(3+residue_nf)] or NULL: z after every denoising step
 *       (after the known/unknown merge, before any jump back); n_steps = sum of the schedule.
 * Asynchronous on `stream`; cmdgen_chain_status reports the deferred mean-zero checks / CoG drift. */
int cmdgen_joint_chain(cmdgen_handle* h, const float* phar_x, const float* phar_onehot,
                       const float* pocket_x, const float* pocket_onehot,
                       const float* phar_fixed, const float* pocket_fixed,
                       int32_t timesteps, int32_t resamplings, int32_t jump_length,
                       const float* noise, int64_t n_draws, uint64_t seed, const int64_t* pocket_ids_host,
                       float* xh_phar_out, float* xh_pocket_out, float* z_steps_out,
                       int32_t use_graph, cmdgen_stream stream);

/* Number of denoising steps (= network evaluations - 1) and of combined noise draws cmdgen_joint_chain will use. */
int cmdgen_joint_plan(cmdgen_handle* h, int32_t timesteps, int32_t resamplings, int32_t jump_length,
                      int32_t inpaint, int64_t* n_steps, int64_t* n_draws);

/* ---- training step (conditional model) -------------------------------------------------- */
/* The trainable tensors of EGNNDynamics live in ONE flat fp32 device buffer owned by the caller, in the
 * reference's registration order (the state_dict order below 'ddpm.dynamics.': weight then bias of every
 * nn.Linear, dynamics.py:21-60, egnn_new.py:15-29, :78-83) - so that the gradient is one contiguous bucket
 * (a single RCCL all-reduce per step replaces DDP's bucketing, train.py:111-121) and the optimizer is one
 * elementwise kernel.  Every tensor starts on a 16-byte boundary (a few zero padding floats in between).
 * cmdgen_param_count (buffer length incl. padding) / cmdgen_param_offset describe the layout
 * (name e.g. "egnn.e_block_0.gcl_0.edge_mlp.0.weight"). */
int cmdgen_param_count(cmdgen_handle* h, int64_t* n_params);
int cmdgen_param_offset(cmdgen_handle* h, const char* name, int64_t* offset, int64_t* count);

/* EGNNDynamics.forward (dynamics.py:75-139) on the parameters `theta`, keeping every activation the backward
 * pass needs (after cmdgen_set_layout; no cmdgen_finalize_weights needed).  t dev [batch].  Writes
 * eps_phar dev [Nl, 3+phar_nf] and, when non-NULL, eps_pocket dev [Np, 3+residue_nf] (required for the joint model,
 * config.update_pocket_coords = 1; the conditional loss never uses it).  Waits once for the host copy of the two list lengths (they
 * size the activation store and the grids; the parameter re-packs are queued behind that copy, so the device keeps working while the
 * host wakes up). */
int cmdgen_train_forward(cmdgen_handle* h, const float* theta, const float* xh_phar, const float* xh_pocket,
                         const float* t, float* eps_phar, float* eps_pocket, cmdgen_stream stream);

/* Backward of the last cmdgen_train_forward: given dL/d eps_phar (dev [Nl, 3+phar_nf]) and optionally
 * dL/d eps_pocket (dev [Np, 3+residue_nf], NULL = zero) ADDS dL/d theta into
 * `grad` (dev, same layout as theta; zero it first for a fresh gradient).  What autograd does for
 * loss.backward() in the reference's training_step (lightning_modules.py:245-260).
 * Streams: the pass queues its weight / bias gradients on streams of the handle beside the chain of data gradients it queues on `stream`
 * (option "wgrad_stream"); before the call returns, `stream` has been made to wait for all of them - work queued on `stream` after the
 * call (the optimizer, an all-reduce ordered behind `stream`) sees the complete gradient, exactly as if everything had run on `stream`.
 * Gradients are accumulated with float atomics: equal run to run up to the order of those sums. */
int cmdgen_train_backward(cmdgen_handle* h, const float* d_eps_phar, const float* d_eps_pocket, float* grad,
                          cmdgen_stream stream);

/* The same pass in stages, so that the gradient all-reduce of the blocks already differentiated overlaps the rest of
 * the backward pass (DDP's bucketed overlap, train.py:111-121): stage 0 = readout, stage k in 1..n_layers = block
 * n_layers-k, stage n_layers+1 = embedding and encoders.  Call with consecutive, ascending stage ranges covering
 * 0..n_layers+1; after stage k the flat gradient is final from cmdgen_param_offset("egnn.e_block_<n_layers-k>.
 * gcl_0.edge_mlp.0.weight") to its end. */
int cmdgen_train_backward_stages(cmdgen_handle* h, const float* d_eps_phar, const float* d_eps_pocket, float* grad,
                                 int32_t first_stage, int32_t last_stage, cmdgen_stream stream);

/* The loss side of ConditionalDDPM.forward in training mode (conditional_model.py:198-320) and of
 * PharPocketDDPM.forward (lightning_modules.py:188-239), fused: three launches per step instead of a few hundred
 * small tensor operations.  Per-sample scalars that depend only on t and on the node counts are made by the host
 * and passed as `tab`, dev, COLUMN-major [CMDGEN_TT_COLS][batch]:
 *   0 alpha_t, 1 sigma_t, 2 t_is_zero, 3 SNR weight 1 - SNR(gamma_s - gamma_t), 4 alpha_T, 5 sigma_T,
 *   6 -log_constants_p_x_given_z0, 7 delta_log_px, 8 log p(N), 9 t_int, 10 t = t_int / T (column 10 is the `t`
 *   argument of cmdgen_train_forward), 11 sigma_t * norm_values[1].
 * cmdgen_train_noise: normalize + remove_mean_batch + noised_representation (conditional_model.py:80-106, :467-475; on a handle
 *   configured with no_com_projection - SimpleConditionalDDPM, :481-525 - the pocket's centre of mass is subtracted instead and nothing
 *   is projected, and the caller's `tab` counts n_phar * 3 degrees of freedom):
 *   from the raw batch (phar_x [Nl,3], phar_one_hot [Nl,phar_nf], pocket_x [Np,3], pocket_one_hot [Np,residue_nf]) and the
 *   Gaussian draw eps [Nl,3+phar_nf] writes z_t [Nl,3+phar_nf], xh_pocket [Np,3+residue_nf] (the network's inputs) and
 *   kl_sums [batch,2] (sum of (alpha_T x)^2 and (alpha_T h)^2 of the clean sample, for kl_prior, :49-59).
 * cmdgen_train_loss: from net_out = eps_phar of cmdgen_train_forward writes
 *   terms [batch, CMDGEN_TS_COLS]: 0 nll, 1 error_t (as logged), 2 loss_0, 3 kl_prior, 4 / 5 mean |eps_hat| of x / h,
 *   6 loss_0_x, 7 loss_0_h, 8 loss_t;   means [CMDGEN_TS_COLS] = their batch means (means[0] is the loss);
 *   d_eps [Nl,3+phar_nf] = d loss / d net_out, the input of cmdgen_train_backward.
 *   l2 != 0: loss_type 'l2' (lightning_modules.py:198-205); 0: the vlb weighting (:206-212) with T diffusion steps. */
#define CMDGEN_TT_COLS 12
#define CMDGEN_TS_COLS 12
int cmdgen_train_noise(cmdgen_handle* h, const float* phar_x, const float* phar_one_hot, const float* pocket_x,
                       const float* pocket_one_hot, const float* tab, const float* eps, float* z_t, float* xh_pocket,
                       float* kl_sums, cmdgen_stream stream);
int cmdgen_train_loss(cmdgen_handle* h, int32_t l2, float T, const float* net_out, const float* eps, const float* z_t,
                      const float* phar_one_hot, const float* tab, const float* kl_sums, float* terms, float* d_eps,
                      float* means, cmdgen_stream stream);

/* The same two launches for the JOINT model (mode 'joint': EnVariationalDiffusion.forward in training mode, en_diffusion.py:332-465,
 * with the joint branch of lightning_modules.py:198-217): the pocket is noised and denoised too.  `tab` as above with the node count
 * n_phar + n_pocket in columns 6 and 7 and the JOINT log p(n_phar, n_pocket) in column 8.
 * cmdgen_train_noise_joint: draw_phar [Nl,3+phar_nf] / draw_pocket [Np,3+residue_nf] are raw Gaussian draws; writes eps_phar / eps_pocket
 *   (the draws with the x-part's centre of mass over all nodes of the sample removed, :555-574), z_phar / z_pocket = alpha_t xh +
 *   sigma_t eps (the network's inputs) and kl_sums [batch,2] over both parts.
 * cmdgen_train_loss_joint: net_phar / net_pocket = both outputs of cmdgen_train_forward; terms columns 0..8 as above (1, 4, 5 for the
 *   phar part; 6 = loss_0_x of both parts), 9 error_t of the pocket part, 10 / 11 mean |eps_hat| of the pocket's x / h;
 *   d_eps_phar / d_eps_pocket = d loss / d net outputs, the inputs of cmdgen_train_backward. */
int cmdgen_train_noise_joint(cmdgen_handle* h, const float* phar_x, const float* phar_one_hot, const float* pocket_x,
                             const float* pocket_one_hot, const float* tab, const float* draw_phar, const float* draw_pocket,
                             float* z_phar, float* z_pocket, float* eps_phar, float* eps_pocket, float* kl_sums, cmdgen_stream stream);
int cmdgen_train_loss_joint(cmdgen_handle* h, int32_t l2, float T, const float* net_phar, const float* net_pocket,
                            const float* eps_phar, const float* eps_pocket, const float* z_phar, const float* z_pocket,
                            const float* phar_one_hot, const float* pocket_one_hot, const float* tab, const float* kl_sums,
                            float* terms, float* d_eps_phar, float* d_eps_pocket, float* means, cmdgen_stream stream);

/* GEMM operand precision of the training step's backward products: 0 (default) = fp32 results (fp32-accurate split-bf16
 * products or, with cmdgen_set_gemm_mode(0), the fp32 matrix instruction), 1 = operands rounded to bf16 (nearest-even),
 * fp32 accumulation on v_mfma_f32_32x32x16_bf16.  Parameters, gradients, optimizer state, stored activations and all
 * elementwise math stay fp32 either way. */
int cmdgen_train_set_precision(cmdgen_handle* h, int32_t bf16_gemm);

/* Sum of squares of a device vector -> host float (the global gradient norm of utils.get_grad_norm,
 * utils.py:39-61, is its square root).  Synchronises the stream. */
int cmdgen_grad_sqnorm(cmdgen_handle* h, const float* grad, int64_t n, float* out_host, cmdgen_stream stream);

/* One torch.optim.AdamW(amsgrad=True) update (lightning_modules.py:141-143) of the flat buffer, with the norm
 * clipping coefficient of clip_grad_norm_ folded in (pass 1.0 for none; lightning_modules.py:543-568 computes
 * it on the host).  `step` counts from 1.  All pointers dev [n]. */
int cmdgen_adamw_step(cmdgen_handle* h, float* theta, const float* grad, float* exp_avg, float* exp_avg_sq,
                      float* max_exp_avg_sq, int64_t n, int64_t step, float lr, float beta1, float beta2,
                      float eps, float weight_decay, float clip_coef, cmdgen_stream stream);

/* Gradient norm, clipping and the AdamW update in one queue-up (lightning_modules.py:543-568 + :141-143): the norm of
 * `grad` is reduced on the device, the update applies clip_grad_norm_'s coefficient min(1, max_grad_norm / (norm + 1e-6))
 * formed on the device (max_grad_norm <= 0: no clipping), and the norm comes back in *grad_norm_host for the caller's
 * queue of recent norms - the host round trip between norm and update of cmdgen_grad_sqnorm + cmdgen_adamw_step is gone.
 * Synchronises the stream - unless grad_norm_host is NULL: then the norm is copied to pinned host memory behind the update
 * and cmdgen_last_grad_norm collects it later (it is only needed before the NEXT step's bound is formed), so the host can
 * queue the next step's noising and graph build while this step's backward pass is still running. */
int cmdgen_adamw_step_clipped(cmdgen_handle* h, float* theta, const float* grad, float* exp_avg, float* exp_avg_sq,
                              float* max_exp_avg_sq, int64_t n, int64_t step, float lr, float beta1, float beta2,
                              float eps, float weight_decay, float max_grad_norm, float* grad_norm_host,
                              cmdgen_stream stream);
/* The norm of the last cmdgen_adamw_step_clipped called with grad_norm_host = NULL (waits for that update only). */
int cmdgen_last_grad_norm(cmdgen_handle* h, float* grad_norm_host);

/* C[M,N] (+)= op(A) op(B) (+ bias) through the training path's exact-fp32 MFMA GEMM (test aid):
 * ta: A stored [K][M]; tb: B stored [N][K] (nn.Linear weight); accumulate bit 0: C += ..., bit 1: bf16 operands. */
int cmdgen_debug_sgemm(cmdgen_handle* h, int32_t ta, int32_t tb, int32_t M, int32_t N, int32_t K, const float* A,
                       int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc, const float* bias,
                       int32_t accumulate, int32_t split_k, cmdgen_stream stream);

/* Optional: supply the per-step scalars of sample_p_zs_given_zt computed by the host
 * (e.g. with the same torch fp32 ops as the reference, bit for bit) instead of the
 * library's own libm evaluation.  coef_host is [K+1][4]:
 *   rows 0..K-1 (s = K-1 .. 0): alpha_ts, sigma2_ts/alpha_ts/sigma_t, sigma_ts*sigma_s/sigma_t, t
 *   row  K     (final decode) : sigma_0, alpha_0, exp(gamma_0/2), 0
 * (conditional_model.py:345-366, :108-131; en_diffusion.py:79-103).  Used by the next
 * cmdgen_sample_chain whose `timesteps` equals K. */
int cmdgen_set_step_table(cmdgen_handle* h, int32_t K, const float* coef_host);

/* Deferred, non-syncing versions of the reference's in-loop checks, read back after the
 * chain (synchronises the stream):
 *   max_rel_com_error : max over steps of assert_mean_zero_with_mask's relative error
 *                       (en_diffusion.py:919-924; the reference asserts < 1e-2)
 *   max_cog           : the final CoG drift (conditional_model.py:451-457)
 *   nan_resets        : evaluations whose output was reset by the NaN guard */
int cmdgen_chain_status(cmdgen_handle* h, float* max_rel_com_error, float* max_cog,
                        int64_t* nan_resets, cmdgen_stream stream);

/* ---- measurement ------------------------------------------------------------------ */
int cmdgen_get_counters(cmdgen_handle* h, cmdgen_counters* out, cmdgen_stream stream); /* syncs */
int cmdgen_reset_counters(cmdgen_handle* h, cmdgen_stream stream);
/* Runs ONE evaluation on the current inputs with a hipEvent pair around every launch and
 * returns the per-kernel sums (synchronises). */
int cmdgen_profile_evaluation(cmdgen_handle* h, const float* xh_phar, const float* xh_pocket,
                              const float* t, float* eps_phar, cmdgen_kernel_times* out,
                              cmdgen_stream stream);
/* Y[M,256] (+)= (A0 W0 + A1 W1) / div * SiLU'(pre) through the training step's data-gradient kernel (test aid; A1 / pre may be
 * NULL): W0 / W1 dev [256][256] row-major nn.Linear weights (dX = dY W; when both are given they must lie in ONE device
 * allocation), packed as the step packs them.  pieces: 3 = fp32-accurate split products, 1 = bf16 operands; tile_rows: 0 =
 * the launcher's choice, 32 / 64 = forced.  Synchronises the stream. */
int cmdgen_debug_dgrad(cmdgen_handle* h, int32_t M, const float* A0, const float* W0, const float* A1, const float* W1,
                       float* Y, int32_t accumulate, float div, const float* pre, int32_t pieces, int32_t tile_rows,
                       cmdgen_stream stream);

/* dW[M,N] += dY^T X (dY dev [K,M], X dev [K,N], both contiguous), db[M] += column sums of dY (db may be NULL), through the
 * training step's weight-gradient launch (test aid).  mode 0 = fp32 instruction, 1 = bf16 operands, 3 = three-piece split
 * where the shape allows, otherwise as 0). */
int cmdgen_debug_wgrad(cmdgen_handle* h, int32_t K, int32_t M, int32_t N, const float* dY, const float* X, float* dW,
                       float* db, int32_t mode, cmdgen_stream stream);

/* Matrix engine of the tile kernels (evaluation, chains, and the fp32 products of the training step: its two forward
 * edge kernels and every [.,256] x [256,256] data gradient; cmdgen_train_set_precision is the separate bf16-OPERAND switch):
 *   1 (default) = split-bf16: every fp32 operand is the exact sum of three bf16 pieces and every fp32 product is six
 *       exact bf16 products accumulated in fp32 on v_mfma_f32_32x32x16_bf16 - fp32-accurate (the dropped terms are
 *       <= 3 * 2^-24 |a||b|, below the fp32 accumulation rounding both engines share) at about twice the delivered
 *       rate of the fp32 matrix instruction; used by tiles of >= 32 rows;
 *   0 = v_mfma_f32_32x32x2_f32 / 16x16x4_f32 everywhere (bitwise an fmaf chain).
 * 16-row node tiles use v_mfma_f32_16x16x32_bf16 the same way (option "node16_split"), 16-row edge / embedding tiles
 * the fp32 instruction.  Changing the mode drops captured step graphs (they are re-captured on the next chain) and
 * re-picks the tile sizes of the current layout. */
int cmdgen_set_gemm_mode(cmdgen_handle* h, int32_t split_bf16);

/* Explicit launch choices of ONE handle (measurement, parity tests, A/B runs).  The library reads no environment variable:
 * everything that used to be a CMDGEN_* switch is a key here.  unset != 0 removes the key (back to the library's own
 * choice; `value` ignored).  Setting an option drops captured step graphs and re-picks the tiles of the current layout.
 *   rows per tile        "node_mt" 16|32|64, "edge_mt" / "coord_mt" 16|32|64|128 (128: kernels_edge128.hip, split engine,
 *                        hidden_nf 256), "embed_mt" 16|32|64
 *   grids                "edge_wgs_per_cu", "coord_wgs_per_cu" (persistent-style edge grids of the <= 64-row kernels),
 *                        "e128_wgs_per_cu" 1|2, "e128_fused" 0..3 (bit 0 / 1: the 128-row message / coordinate kernel issues the next quarter's
 *                        tile build inside its GEMM; default: by the estimated list length - more than one tile per workgroup)
 *   matrix engine        "half_engine" 0|1|2: the split-engine kernels that have a HALF form (two fp16 pieces per operand, three MFMAs per
 *                        fp32 product instead of three bf16 pieces and six; csrc/cmdgen_split.h) use it: 1 (default) when the model has an
 *                        edge cutoff (the radial features are bounded; fp16 ends at 65504), 2 always, 0 never
 *   kernel variants      "edge_fullk" 0|1 (full-K planes for 32-row edge tiles), "node64" 0|1|32 (64-row planes node kernel;
 *                        32: its 32-row form), "node16_split" 0|1, "node16w" 0|1 (16-row node tiles of H = 256 on eight waves,
 *                        kernels_node16w.hip; default 1), "write_embed" 0|1 (graph pass 2 + k_embed in one launch)
 *   dead work            "dead_skip" 0|1|2 (2, default: every block skips tiles beyond L - l hops of a moving node; 1: the
 *                        last block only; 0: off)
 *   chain                "fused_step" 0|1 (posterior step + graph pass 1 in one kernel), "pocket_cache" 0|1, "graph_steps"
 *                        (denoising steps per captured graph, default 8)
 *   training step        "wgrad_split" -1|0|1 (-1: three-piece weight gradients wherever the handle runs the split engine), "wgrad_tile" 0|64, "wgrad_split_wgs128", "wgrad_split_wgs64", "wgrad_wgs",
 *                        "dgrad_mt" 0|32|64, "dgrad_tail" 0|1   (see TrainTune in csrc/cmdgen_dev.h);
 *                        "wgrad_stream" 0|1|2 (weight / bias gradients and partial-sum reductions on the handle's second stream beside the chain
 *                        of data gradients; 2: at the lowest stream priority; default 1), "train_half" 0|1|2 (the training forward's tile kernels on
 *                        the half engine, packs and scales re-made on the device every step; 2: the two edge kernels only; default 1 where
 *                        "half_engine" resolves on),
 *                        "wgrad_silu" 0..3 (bit 0 / 1: SiLU(pre1) / SiLU(pre6) are not stored by the forward but formed by the second layer's
 *                        weight gradient; default 3 with bf16 operands, 0 for fp32 results), "wgrad_k128" (rows x 128-tiles of a launch from which a
 *                        three-piece weight gradient uses 128 x 128 tiles; default 131072), "train_node16" 0|1 (16-row node tiles in the training
 *                        forward at every batch size; default 1)
 *   (round 6: the options whose A/B lost - "e128_pp", "dgrad_half", "small_wgrads" - left the build: profiles/r06_removed_experiments.patch)
 * cmdgen_get_option: *value = the stored value, *is_set = 0 when the key is not set (either pointer may be NULL).
 * Unknown keys: CMDGEN_EINVAL. */
int cmdgen_set_option(cmdgen_handle* h, const char* key, int64_t value, int32_t unset);
int cmdgen_get_option(cmdgen_handle* h, const char* key, int64_t* value, int32_t* is_set);

/* Diagnostic builds only (kernels compiled with -DCMDGEN_STAMPS): 64 summed in-kernel cycle stamps of k_edge_msg
 * ([wave][phase], wave lifetimes, wave count); all zero in production builds.  Synchronises the device. */
int cmdgen_debug_stamps(cmdgen_handle* h, uint64_t* out64, int32_t reset);

/* Launch configuration in force for the current layout (measurement aid): key = "node_mt" | "edge_mt" | "coord_mt"
 * (rows per tile of the three MFMA kernels), "edge_grid" | "coord_grid" (workgroups of the persistent-style edge
 * kernels), "gemm_split" (the mode above), "node16_split", "node16w", "node64", "edge_fullk", "dead_skip" (as resolved from the
 * options and the layout), "half_engine" (the engine option as resolved: 1 = kernels with a half form use it), "msg_mfmas_per_product" |
 * "node_mfmas_per_product" | "coord_mfmas_per_product" (1: the fp32 matrix instruction, 6: three bf16 pieces per operand, 3: two fp16 pieces - the
 * engine the three tile kernels of the current layout run on), "train_edges" | "train_coord_edges" (edges of the last cmdgen_train_forward). */
int cmdgen_query(cmdgen_handle* h, const char* key, int64_t* value);

/* Steady-state timing of one network evaluation (bench.py's trained-geometry micro-benchmark): `graph_len`
 * evaluations of the given inputs are captured into a hipGraph, replayed `replays` times after one warm-up replay
 * and timed with HIP events on the launch stream.  *mean_ms = time per evaluation.  Counters advance as usual. */
int cmdgen_time_evaluation(cmdgen_handle* h, const float* xh_phar, const float* xh_pocket, const float* t,
                           float* eps_phar, int32_t graph_len, int32_t replays, float* mean_ms, cmdgen_stream stream);

/* Replays the edge-message kernel of block `layer` `reps` times on the state left by the
 * last evaluation and returns the mean launch duration in ms (hipEvents on `stream`). */
int cmdgen_time_edge_kernel(cmdgen_handle* h, int32_t layer, int32_t reps, float* mean_ms,
                            cmdgen_stream stream);

/* Per-launch timing of the three MFMA kernels inside a real chain: while enabled, every eager
 * (use_graph = 0) launch of k_edge_msg / k_node / k_edge_coord is bracketed by a hipEvent pair on
 * the launch stream.  cmdgen_get_kernel_profile synchronises and returns, per kernel class
 * (CMDGEN_K_EDGE_MSG, CMDGEN_K_NODE, CMDGEN_K_EDGE_COORD), the summed duration and the number of
 * launches since the last call; it clears the record. */
#define CMDGEN_K_EDGE_MSG   0
#define CMDGEN_K_NODE       1
#define CMDGEN_K_EDGE_COORD 2
int cmdgen_set_kernel_profiling(cmdgen_handle* h, int32_t on);
int cmdgen_get_kernel_profile(cmdgen_handle* h, float total_ms[3], int64_t launches[3], cmdgen_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* CMDGEN_HIP_H */
