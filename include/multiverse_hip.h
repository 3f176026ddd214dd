/*
 * multiverse_hip.h -- C ABI of the MI355X-native Multiverse engine
 * (libmultiverse_hip.so, built from multiverse_amd/csrc/).
 *
 * The reference (JunweiLiang/Multiverse) has no plugin / operator / FFI
 * interface: its device boundary is `sess.run(fetches, feed_dict)` inside
 *   Tester.step   code/pred_models.py:1761-1790   (greedy forward)
 *   Trainer.step  code/pred_models.py:1719-1742   (training step)
 *   multifuture_inference.py:460-472              (beam-search forward)
 * over the TF-1 graph built by Model.build_forward (code/pred_models.py:123-308).
 * The entry points below are what a binding for that boundary binds: one call
 * per `sess.run`, numpy buffers in, numpy buffers out, weights addressed by
 * their TF-1 variable names (HWIO layout, exactly what tf.train.Saver stores).
 * No torch / TF types appear in any signature.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; the message is
 *     available from mv_last_error(handle) (or mv_last_error(NULL) for
 *     create-time failures).
 *   - all tensors are dense, row-major, NHWC; float = IEEE binary32,
 *     integers = int32.
 *   - input buffers are caller-owned and only read during the call; output
 *     buffers are caller-allocated and fully written on success.
 *   - calls on one handle are not re-entrant (the reference is a single
 *     synchronous Python thread).  Each handle owns one
 *     HIP stream; calls return after their results are on the host unless the
 *     *_async / *_resident variants are used.
 */
#ifndef MULTIVERSE_HIP_H_
#define MULTIVERSE_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MV_MAX_SCALES 2
#define MV_ABI_VERSION 5

typedef struct mv_engine* mv_handle;

/* Mirrors the config fields Model reads; authoritative list:
 * reference code/multifuture_inference.py:419-452. */
typedef struct mv_config {
  int32_t abi_version;       /* MV_ABI_VERSION */
  int32_t batch_size;        /* N, fixed per model instance (pred_models.py:47) */
  int32_t obs_len;           /* T_o */
  int32_t max_pred_len;      /* upper bound of the run-time T_pred */
  int32_t scene_h, scene_w, scene_class;        /* 36, 64, 11 */
  int32_t scene_conv_dim, scene_conv_kernel;    /* 64 (a multiple of 32 up to 128), 3 */
  int32_t emb_size;          /* 32 */
  int32_t hidden_size;       /* enc_hidden_size == dec_hidden_size, 256 */
  int32_t convlstm_kernel;   /* 3; other sizes 1 .. 9 run the generic fp32 loops (compute
                              * mode 0 only, csrc/convlstm_generic.h) */
  int32_t num_scales;        /* len(scene_grid_strides), <= MV_MAX_SCALES */
  int32_t grid_h[MV_MAX_SCALES];
  int32_t grid_w[MV_MAX_SCALES];
  int32_t use_grid[MV_MAX_SCALES];
  int32_t use_gnn;           /* --use_gnn */
  int32_t beam_size;         /* 1 = greedy (pred_models.py:260-285) */
  int32_t diverse_beam;      /* --diverse_beam */
  float   diverse_gamma;     /* --diverse_gamma */
  int32_t fix_num_timestep;  /* --fix_num_timestep */
  /* --use_teacher_forcing at TEST time: decoder_loop_fn takes its teacher_forcing branch
   * (code/pred_models.py:388-406), whose not-training arm feeds hidden2grid(h') itself --
   * raw logits, no argmax / one-hot -- to the class decoder's grid_emb.  0 = the
   * published one-hot feedback. */
  int32_t class_feedback_dense;
  /* --use_single_decoder (code/pred_models.py:274,287-296; "# bad" in code/train.py:98): no
   * regression decoder; the offsets are hidden2grid(class-decoder states) through ONE
   * kernel "person_pred/decode_reg/out_dec_grid/W" shared by the scales.  The regression
   * encoder's variables still exist (it is built, :232-234) but are neither run nor
   * trained.  Greedy decode and training; refused with beam search (the reference's
   * inference script reshapes the [N*B, ...] offsets of that combination as [1, T, -1, 2],
   * code/multifuture_inference.py:478). */
  int32_t use_single_decoder;
  /* 1 = the graph of SimAug/code/pred_models.py (the fork N4 trains with): its gnn_edge
   * concatenates the scene features to the hidden state only under tile_to_beam
   * (SimAug/code/pred_models.py:1219-1227), so the GREEDY decoder's graph attention -- test
   * and training -- sees the hidden state alone; the beam-search decoder is unchanged.
   * Found by executing that file on the TF-1 shim next to code/pred_models.py. */
  int32_t simaug_graph;
  /* --activation_func (code/train.py:58-59, code/pred_utils.py:86-94): the activation of
   * the scene convolutions (code/pred_models.py:155-165) and of grid_emb (:444, :664).
   * 0 = tanh (published), 1 = relu, 2 = lrelu (tf.nn.leaky_relu, alpha 0.2).  relu / lrelu
   * outputs are unbounded: in compute mode 1 their x operand planes carry a per-tensor
   * power-of-two scale taken from max |x| (DESIGN.md section 3c "x exponent"), so every
   * compute mode decodes AND trains them.  In compute mode 2 (bf16) the x k-steps of such a
   * model run as an f16x3 split of the x part alone (fp16 planes under that exponent, three
   * fp16 MFMAs per product) while the h k-steps stay one bf16 plane: one bf16 plane of a
   * pixel-offset embedding of hundreds had cost the regression decoder's kernel gradient its
   * direction (cosine 0.96 vs the fp32 oracle; now 0.99995, the tanh models' figure). */
  int32_t activation;
} mv_config;

/* The feed_dict of Model.get_feed_dict (pred_models.py:1042-1194), minus the
 * placeholders the forward graph never consumes. */
typedef struct mv_inputs {
  const int32_t* obs_scene;        /* [N, T_o] row index into scene_feat */
  const float*   scene_feat;       /* [U, SH, SW, SC] 0/1 masks */
  int32_t        num_scene_frames; /* U */
  int32_t        pred_len;         /* run-time T_pred (<= max_pred_len) */
  const int32_t* grid_obs_labels[MV_MAX_SCALES];   /* [N, T_o] */
  const float*   grid_obs_regress[MV_MAX_SCALES];  /* [N, T_o, H, W, 2] */
} mv_inputs;

/* Fetches of Tester.step (pred_models.py:1774-1790).  Pointers for unused
 * scales may be NULL. */
typedef struct mv_outputs {
  float* grid_pred_class[MV_MAX_SCALES]; /* [N, T_p, H, W, 1] logits */
  float* grid_pred_reg[MV_MAX_SCALES];   /* [N, T_p, H, W, 2] */
} mv_outputs;

/* model.beam_outputs (pred_models.py:276, 805-806) + the two greedy fetches
 * of multifuture_inference.py:468-472 for the single active scale. */
typedef struct mv_beam_outputs {
  float*   best_beam;   /* [N, T, H, W, 1]  == logits[:, 0] */
  float*   grid_reg;    /* [N, T, H, W, 2]  (regression decoder, un-beamed); with
                         * use_single_decoder [N*B, T, H, W, 2]: the offsets decoded from the
                         * class decoder's states traced back along every beam
                         * (code/pred_models.py:287-296) */
  float*   logits;      /* [N, B, T, H*W] */
  int32_t* ids;         /* [N, B, T] */
  float*   logprobs;    /* [N, B] */
} mv_beam_outputs;

/* -- training step: Trainer (pred_models.py:1636-1742) ------------------- */
/* The optimizer / loss fields Trainer.__init__ and Model.build_loss read from
 * the config (train.py:25-138 after process_args). */
typedef struct mv_train_config {
  /* --optimizer (code/pred_models.py:1667-1681): 0 adadelta (published), 1 momentum
   * (0.9), 2 adam, 3 rmsprop; TF-1.15 defaults and ApplyXxx kernel arithmetic */
  int32_t optimizer;
  float   init_lr, emb_lr;
  int32_t use_cosine_lr;      /* --use_cosine_lr */
  int32_t has_decay;          /* learning_rate_decay is not None */
  float   learning_rate_decay;
  int32_t decay_steps;        /* int(train_num_examples / batch_size * num_epoch_per_decay) */
  int32_t max_steps;          /* cosine: int(train_num_examples / batch_size * num_epochs) */
  int32_t do_clip;            /* clip_gradient_norm is not None */
  float   clip_gradient_norm; /* ELEMENT-WISE clip_by_value bound (pred_models.py:1700-1705) */
  float   wd;                 /* weight decay on every variable named .../W */
  float   grid_loss_weight, grid_reg_loss_weight;
  /* --- switches of Model.build_forward / build_loss beyond the published run --- */
  /* class-decoder next input while training (decoder_loop_fn, :388-436):
   *   0  one_hot(argmax(hidden2grid(h')))   --train_w_onehot (published; no gradient)
   *   1  hidden2grid(h') itself             neither flag (differentiable feedback)
   *   2  the ground-truth map of the step   --use_teacher_forcing (pred_gt.read(time)) */
  int32_t class_feedback;
  int32_t reg_teacher_forcing;   /* regression decoder: 0 own output, 1 ground truth */
  /* --use_soft_grid_class (:974-990, 1077-1124): labels = the one-hot map stamped with
   * the --soft_grid kernel (soft_kernel_size 3 or 5, row-major soft_kernel); the loss is
   * softmax_cross_entropy_with_logits with TF's registered gradient softmax - labels */
  int32_t use_soft_grid_class;
  int32_t soft_kernel_size;
  float   soft_kernel[25];
  int32_t mask_grid_regression;  /* --mask_grid_regression (:999-1018) */
  float   keep_prob;             /* DropoutWrapper input keep probability (:130-132) */
} mv_train_config;

/* Training-only placeholders of Model.get_feed_dict(is_train=True)
 * (pred_models.py:1149-1152). */
typedef struct mv_targets {
  const int32_t* grid_pred_labels[MV_MAX_SCALES];   /* [N, T_p] */
  const float*   grid_pred_regress[MV_MAX_SCALES];  /* [N, T_p, H, W, 2] */
} mv_targets;
/* (soft labels, the teacher-forcing inputs and the foreground mask of the masked
 * regression loss are all functions of grid_pred_labels: the engine builds them in HBM) */

/* Fetches of Trainer.step: loss, wd_loss, pred_grid_loss = [cls_s, reg_s, ...]
 * over the enabled scales (pred_models.py:1029, 1719-1742). */
typedef struct mv_losses {
  float   loss, wd_loss;
  float   pred_grid_loss[2 * MV_MAX_SCALES];
  int32_t num_pred_grid_loss;
} mv_losses;

/* -- lifetime ------------------------------------------------------------ */
int  mv_create(const mv_config* cfg, int device, mv_handle* out);
int  mv_destroy(mv_handle h);
const char* mv_last_error(mv_handle h);
int  mv_abi_version(void);

/* -- weights: tf.train.Saver role (pred_utils.py:149-205) ---------------- */
/* tf_name e.g. "person_pred/decoder_grid_class_0/decoder_rnn/dec_grid_0/kernel" */
int  mv_set_param(mv_handle h, const char* tf_name, const float* data,
                  const int64_t* shape, int32_t rank);
int  mv_get_param(mv_handle h, const char* tf_name, float* out,
                  int64_t capacity_elems);
int  mv_num_params(mv_handle h);
/* name/shape of the i-th parameter the engine expects; rank is returned */
int  mv_param_info(mv_handle h, int32_t i, char* name_out, int32_t name_cap,
                   int64_t* shape_out /* [4] */);

/* -- forward: one call == one sess.run ------------------------------------ */
int  mv_forward_greedy(mv_handle h, const mv_inputs* in, mv_outputs* out);
int  mv_forward_beam(mv_handle h, const mv_inputs* in, mv_beam_outputs* out);

/* -- resident-input variants (bench: inputs already in HBM) -------------- */
int  mv_upload_inputs(mv_handle h, const mv_inputs* in);   /* H2D + sync */

/* Device-side batch assembly.  The dense regression maps the graph
 * consumes are a pure function of one (x, y) per trajectory step,
 *   map[n, t, h, w, :] = (float)((double)xy[n, t, :] - centre[h, w, :])
 * (code/preprocess.py:463-475: float32 trajectory minus float64 grid centres, stored
 * as float32), and the scene masks are 0/1 bytes in data_*.npz (preprocess.py:831).
 * The compact forms hand over 16 bytes per step instead of 2*H*W*4 per scale and
 * uint8 masks; the engine expands both in HBM, bit-identical to the dense upload.
 * Rows n >= num_rows (the batcher's padding, code/pred_models.py:1070-1191) get zero
 * maps.  mv_set_grid_centers once per scale (data["grid_center_<s>"], float64
 * [H, W, 2]); then mv_upload_inputs_compact (+ mv_upload_targets_compact when
 * training) and the resident calls (mv_run_*_resident, mv_train_step(h, NULL, NULL, ..)). */
typedef struct mv_inputs_compact {
  const int32_t* obs_scene;        /* [N, T_o] row index into scene_feat_u8 */
  const uint8_t* scene_feat_u8;    /* [U, SH, SW, SC] 0/1 masks */
  int32_t        num_scene_frames; /* U */
  int32_t        pred_len;         /* run-time T_pred (<= max_pred_len) */
  const int32_t* grid_obs_labels[MV_MAX_SCALES];   /* [N, T_o] */
  const double*  obs_xy;           /* [N, T_o, 2] pixel coordinates */
  int32_t        num_rows;         /* rows that carry data (<= N) */
} mv_inputs_compact;

typedef struct mv_targets_compact {
  const int32_t* grid_pred_labels[MV_MAX_SCALES];  /* [N, T_p] */
  const double*  pred_xy;          /* [N, T_p, 2] */
  int32_t        num_rows;
} mv_targets_compact;

int  mv_set_grid_centers(mv_handle h, int32_t scale, const double* centers /* [H, W, 2] */);
int  mv_upload_inputs_compact(mv_handle h, const mv_inputs_compact* in);
int  mv_upload_targets_compact(mv_handle h, const mv_targets_compact* tg);
/* Pipelined greedy forward.  One sess.run of the reference is feed + compute + fetch in turn
 * (code/pred_models.py:1761-1790); an evaluation loop (code/pred_utils.py:415) knows its next
 * batch while the current one computes.  mv_submit_greedy copies the caller's buffers (free
 * again on return) into one of `depth` pinned slots and queues H2D -> forward -> D2H on the
 * engine's copy and compute streams; mv_collect_greedy blocks for the OLDEST submission and
 * fills `out` (sized for that submission's pred_len, returned through *pred_len when not
 * NULL).  At most `depth` submissions may be outstanding.  Results are bitwise those of
 * mv_forward_greedy. */
int  mv_pipeline_create(mv_handle h, int32_t depth);
int  mv_submit_greedy(mv_handle h, const mv_inputs* in);
int  mv_collect_greedy(mv_handle h, mv_outputs* out, int32_t* pred_len);
int  mv_run_greedy_resident(mv_handle h);   /* enqueue on the handle's stream */
int  mv_run_beam_resident(mv_handle h);
int  mv_synchronize(mv_handle h);
int  mv_download_outputs(mv_handle h, mv_outputs* out);
int  mv_download_beam_outputs(mv_handle h, mv_beam_outputs* out);

/* -- training: one call == sess.run([loss, train_op, wd_loss, pred_grid_loss]) */
int  mv_train_init(mv_handle h, const mv_train_config* tc);
/* forward (is_train, --train_w_onehot wiring) + loss + backward + clip +
 * Adadelta + global_step++ on one device */
int  mv_train_step(mv_handle h, const mv_inputs* in, const mv_targets* tg,
                   mv_losses* out);
/* data-parallel form: gradients of the LOCAL batch mean are left in one flat
 * device buffer (mv_grad_buffer) for the caller's all-reduce(sum) over the
 * ranks (RCCL via torch.distributed), then mv_train_apply(1/world) clips,
 * applies Adadelta and advances global_step -- clip AFTER the all-reduce, as a
 * single-device step over the global batch would. */
int  mv_train_forward_backward(mv_handle h, const mv_inputs* in,
                               const mv_targets* tg, mv_losses* out);
/* resident form (bench: batch already in HBM): mv_upload_inputs +
 * mv_upload_targets once, then pass in == tg == NULL to the two calls above */
int  mv_upload_targets(mv_handle h, const mv_targets* tg);
int  mv_grad_buffer(mv_handle h, float** device_ptr, int64_t* elems);
int  mv_train_apply(mv_handle h, float grad_scale);
/* -- in-library gradient all-reduce (RCCL over xGMI) ----
 * One process per GPU.  Rank 0 draws an id (mv_comm_unique_id) and hands its 128 bytes to
 * the other ranks out of band (any bootstrap: a file, MPI, torch.distributed); every rank
 * then calls mv_allreduce_init(h, rank, world, id) after hipSetDevice / mv_create on ITS
 * device.  From then on mv_train_forward_backward leaves the SUM over the ranks in the
 * gradient buffer: the buffer is reduced in buckets (one per ConvLSTM kernel + biases, then
 * the small tensors as one group) on a side stream, each bucket issued as soon as its
 * gradients are final, overlapped with the rest of the backward pass; mv_train_step
 * applies 1/world before the clip and the optimizer.  RCCL is dlopen'ed at the first call;
 * a single-device user never needs it. */
#define MV_COMM_ID_BYTES 128
int  mv_comm_unique_id(uint8_t* id_out /* [MV_COMM_ID_BYTES] */);
int  mv_allreduce_init(mv_handle h, int32_t rank, int32_t world,
                       const uint8_t* unique_id /* [MV_COMM_ID_BYTES] */);
/* rank / world of the communicator, collectives and bytes of the last step */
int  mv_allreduce_info(mv_handle h, int32_t* rank, int32_t* world, int32_t* buckets,
                       double* bytes);
/* -- SimAug training extras (SimAug/code/pred_models.py:60-172
 * white_box_attack, :346-543 multiview_augmentation): targeted FGSM / PGD on the scene
 * features and mixup, on the SAME forward / backward kernels.  The attack loss is the
 * class cross entropy against target labels: run the engine with those labels as
 * grid_pred_labels, grid_reg_loss_weight = 0 and wd = 0 (sign() ignores the loss scale).
 *   mv_attack_begin        snapshot the resident scene features as "clean" and make the
 *                          backward pass go down to d loss / d scene_feat
 *   mv_set/get_scene_feat  replace / read the resident features [U, SH, SW, SC] (the
 *                          attack perturbs per (n, t): upload U = N*T_o frames)
 *   mv_get_scene_grad      d loss / d scene_feat of the last mv_train_forward_backward
 *   mv_attack_step         x <- clip(x - step * sign(g), clip(clean - eps, -1, 1),
 *                          clip(clean + eps, -1, 1))   (FGSM: step = eps, once)
 *   mv_scene_mix           x <- other * weight + x * (1 - weight); other NULL = clean
 *   mv_get_sample_losses   per-sample mean cross entropy [N] of the last step (the
 *                          multi-view selection ranks views by it)
 *   mv_attack_end          back to plain training */
int  mv_attack_begin(mv_handle h);
int  mv_attack_end(mv_handle h);
int  mv_set_scene_feat(mv_handle h, const float* scene_feat);
int  mv_get_scene_feat(mv_handle h, float* out);
int  mv_get_scene_grad(mv_handle h, float* out);
int  mv_attack_step(mv_handle h, float epsilon, float step);
int  mv_scene_mix(mv_handle h, const float* other, float weight);
int  mv_get_sample_losses(mv_handle h, int32_t scale, float* out);
/* SimAug multi-view experiment 3 (SimAug/code/pred_models.py:486-517, 616-636, 1371-1398): the
 * training step on MIXED-UP labels.  obs_labels2[s] [N, T_o] / pred_labels2[s] [N, T_p] = the
 * grid labels of the selected extra camera view (NULL for unused scales); `weight` = the beta
 * weight of the original labels: the class-encoder input and the first decoder input use
 * w * one_hot(label) + one_hot(label2) * (1 - w), the class loss is
 * softmax_cross_entropy_with_logits_v2 against the same mix of the future labels;
 * sample_weight [N] (or NULL) multiplies each sample's class-loss rows (double_weighting: the
 * focal weights).  In force for every following mv_train_* call until cleared. */
int  mv_set_label_mixup(mv_handle h, const int32_t* const* obs_labels2,
                        const int32_t* const* pred_labels2, float weight,
                        const float* sample_weight);
int  mv_clear_label_mixup(mv_handle h);
/* tf.gradients(loss, var) of the last forward_backward, by variable name */
int  mv_get_grad(mv_handle h, const char* tf_name, float* out, int64_t capacity_elems);
int  mv_get_global_step(mv_handle h, int64_t* step);
int  mv_set_global_step(mv_handle h, int64_t step);
/* keep_prob < 1: seed of the NEXT step's dropout masks.  TensorFlow's dropout is
 * unseeded; the engine's masks come from a counter-based hash of (seed, draw number,
 * element) -- oracle/multiverse_oracle.py dropout_keep_mask restates it -- with draws
 * numbered in the reference's cell-call order. */
int  mv_set_dropout_seed(mv_handle h, uint32_t seed);
/* Adam's beta1_power / beta2_power (float32 non-slot variables of the checkpoint) */
int  mv_get_opt_scalars(mv_handle h, float* beta1_power, float* beta2_power);
int  mv_set_opt_scalars(mv_handle h, float beta1_power, float beta2_power);
/* optimizer slots for checkpoint save / restore, TF slot names by optimizer:
 *   adadelta  0 "Adadelta" (accum)   1 "Adadelta_1" (accum_update)
 *   momentum  0 "Momentum"
 *   adam      0 "Adam" (m)           1 "Adam_1" (v)
 *   rmsprop   0 "RMSProp" (ms, initialised to ones)   1 "RMSProp_1" (momentum) */
int  mv_get_opt_slot(mv_handle h, const char* tf_name, int32_t slot, float* out,
                     int64_t capacity_elems);
int  mv_set_opt_slot(mv_handle h, const char* tf_name, int32_t slot,
                     const float* data, int64_t elems);

/* Arithmetic of the gate convolutions (forward, and dgrad / wgrad of a training engine):
 *   0  fp32 MFMA (v_mfma_f32_32x32x2_f32), default;
 *   1  "f16x3": every fp32 operand as two pre-scaled fp16 planes, each product as
 *      three fp16 MFMAs accumulating in fp32 -- fp32-roundoff-class error (DESIGN.md
 *      section 3c).  Where the grid fits its tiling the forward step and dgrad run in
 *      Winograd form over image rows (csrc/convlstm_wino.h: fewer MFMA products for the
 *      same pre-activations, kernel transform in fp64 at pack time); otherwise the direct
 *      3x3 form.  Same outputs contract;
 *   2  bf16 operands (one MFMA per product), fp32 accumulate: REDUCED precision
 *      (BASELINE configs[4]). */
int  mv_set_compute_mode(mv_handle h, int32_t mode);

/* Replay the forward as a captured hipGraph (one graph per (mode, T_pred, U)):
 * the reference's whole forward is ONE sess.run (pred_models.py:1779), here it
 * is one hipGraphLaunch instead of ~150 kernel launches.  Off by default. */
int  mv_set_graph_mode(mv_handle h, int32_t enabled);

/* -- measurement --------------------------------------------------------- */
/* When enabled every kernel launch is bracketed by hipEvents on the handle's
 * stream; totals are read back per kernel name. */
int  mv_set_profiling(mv_handle h, int32_t enabled);
int  mv_reset_kernel_stats(mv_handle h);
int  mv_num_kernel_stats(mv_handle h);
int  mv_kernel_stat(mv_handle h, int32_t i, char* name_out, int32_t name_cap,
                    int64_t* launches, double* total_ms, double* flops,
                    double* bytes);
/* `flops` above are the algorithmic FLOPs the launches EXECUTED (the zero-state first
 * encoder step skips the h half of the gate convolution, SURVEY.md 8d: "do not count
 * them as achieved FLOPs"); this returns the same steps counted densely, as the
 * reference computes them: 2 M 9 (Cx + C) 4C per ConvLSTM step. */
int  mv_kernel_stat_dense_flops(mv_handle h, int32_t i, double* flops_dense);
/* FLOPs the launches ISSUED to the matrix pipe: the executed algorithmic count x 3 for the
 * direct f16x3 gate kernel (three fp16 MFMAs per fp32 product), x 2 for its Winograd F(2,3)
 * form, x 1 on the fp32 / bf16 pipes; 0 for kernels that carry no such figure. */
int  mv_kernel_stat_mfma_flops(mv_handle h, int32_t i, double* flops_mfma);
/* elapsed ms between two events recorded around fn on the engine's stream */
int  mv_time_greedy_resident(mv_handle h, int32_t iters, float* ms_out);
int  mv_time_beam_resident(mv_handle h, int32_t iters, float* ms_out);

/* -- single-kernel entry points (parity tests go through these) ----------- */
/* One tf.contrib.rnn.ConvLSTMCell step on host buffers:
 * x [M,H,W,Cx], c,h [M,H,W,C], kernel [3,3,Cx+C,4C] HWIO, biases [4C]. */
int  mv_op_convlstm_step(int device, const float* x, const float* c,
                         const float* h, const float* kernel,
                         const float* biases, int32_t M, int32_t H, int32_t W,
                         int32_t Cx, int32_t C, float* c_out, float* h_out);
/* The same step on the fp16 matrix pipe at fp32 accuracy (f16x3 operand planes):
 * variant 1 = direct 3x3 form, 2 = Winograd F(2,3) over image rows (W must divide 32),
 * 3 = Winograd F(3,3) over image rows (any W -- widths that do not divide 32 take the halo
 * tiling, operands below 2 GiB --, H >= 3; csrc/convlstm_wino3.h), 4 = REDUCED precision: one
 * bf16 plane per operand on the same row-triple tile (compute mode 2's gate kernel; h16_out is
 * then that one plane decoded).
 * h16_out (optional) [M,H,W,C]: the h' OPERAND PLANES the kernel emitted for the next
 * step, decoded back to fp32 ((hi + lo) / 256), so that a test sees the plane layout. */
int  mv_op_convlstm_step16(int device, int32_t variant, const float* x, const float* c,
                           const float* h, const float* kernel, const float* biases,
                           int32_t M, int32_t H, int32_t W, int32_t Cx, int32_t C,
                           float* c_out, float* h_out, float* h16_out);
/* h + GNN(h): gnn_edge/gnn_mask_edge/gnn_node (pred_models.py:808-909);
 * h [M,H,W,C], scene_mean [M,H,W,D]. */
int  mv_op_gnn(int device, const float* h, const float* scene_mean, int32_t M,
               int32_t H, int32_t W, int32_t C, int32_t D, float* out);
/* hidden2grid (pred_models.py:925-959): h [M,H,W,C], w [3,3,C,P] -> [M,H,W,P] */
int  mv_op_hidden2grid(int device, const float* h, const float* w, int32_t M,
                       int32_t H, int32_t W, int32_t C, int32_t P, float* out);
/* One beam expansion (pred_models.py:557-591 + add_div_penalty :1197-1223):
 * logits [N,B,K], prev_logprob [N,B] -> new_logprob, ids, parents [N,B]. */
int  mv_op_beam_step(int device, const float* logits, const float* prev_logprob,
                     int32_t N, int32_t B, int32_t K, int32_t time,
                     int32_t diverse, float gamma, int32_t fix_num_timestep,
                     float* new_logprob, int32_t* ids, int32_t* parents);

/* Backward of one ConvLSTMCell step (tf.gradients through the cell): inputs as
 * mv_op_convlstm_step (c == h == NULL: zero state) plus d h', d c' [M,H,W,C];
 * outputs d x [M,H,W,Cx], d h, d c [M,H,W,C], d kernel [3,3,Cx+C,4C], d biases [4C]. */
int  mv_op_convlstm_bwd(int device, const float* x, const float* c, const float* h,
                        const float* kernel, const float* biases,
                        const float* dh_new, const float* dc_new, int32_t M,
                        int32_t H, int32_t W, int32_t Cx, int32_t C, float* dx,
                        float* dh, float* dc, float* dkernel, float* dbiases);
/* Backward of h + GNN(h) given g = d out: d h [M,H,W,C], d scene_mean [M,H,W,D]. */
int  mv_op_gnn_bwd(int device, const float* h, const float* scene_mean,
                   const float* g, int32_t M, int32_t H, int32_t W, int32_t C,
                   int32_t D, float* dh, float* dscene_mean);

#ifdef __cplusplus
}
#endif
#endif  /* MULTIVERSE_HIP_H_ */
