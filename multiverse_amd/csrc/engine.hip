// libmultiverse_hip.so -- engine behind include/multiverse_hip.h.
//
// One mv_engine == one Model instance of the reference
// (code/pred_models.py:32-121): fixed batch size N, weights addressed by TF-1
// variable name, one HIP stream.  mv_forward_greedy / mv_forward_beam play the
// role of `sess.run` at code/pred_models.py:1779 and
// code/multifuture_inference.py:468-472.
#include "../../include/multiverse_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "convlstm_mfma.h"
#include "convlstm_wgrad.h"
#include "convlstm_f16x3.h"
#include "convlstm_wino.h"
#include "convlstm_wino3.h"
#include "convlstm_generic.h"
#include "convlstm_wgrad_f16x3.h"
#include "kernels_misc.h"
#include "decode_tail.h"
#include "sparse_x.h"
#include "train_kernels.h"
#include "comm.h"

struct mv_train_holder;

namespace {

thread_local std::string g_create_error;



struct HipError { std::string msg; };

#define HIP_CHECK(expr)                                                       \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) {                                                   \
      char _b[512];                                                           \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr,                \
               hipGetErrorString(_e), __FILE__, __LINE__);                    \
      throw HipError{_b};                                                     \
    }                                                                         \
  } while (0)

#define MV_REQUIRE(cond, ...)                                                 \
  do {                                                                        \
    if (!(cond)) {                                                            \
      char _b[512];                                                           \
      snprintf(_b, sizeof(_b), __VA_ARGS__);                                  \
      throw HipError{_b};                                                     \
    }                                                                         \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  T* base = nullptr;     // allocation start (p - pad)
  size_t n = 0;
  // `pad` elements of zeroed slack before and after (operands of the wgrad
  // kernel, whose masked lanes may read one cell outside the tensor)
  void alloc(size_t count, size_t pad = 0) {
    if (count <= n && p) return;
    release();
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&base), (count + 2 * pad) * sizeof(T)));
    if (pad) HIP_CHECK(hipMemset(base, 0, (count + 2 * pad) * sizeof(T)));
    p = base + pad;
    n = count;
  }
  void release() {
    if (base) (void)hipFree(base);
    p = nullptr; base = nullptr; n = 0;
  }
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

struct Param {
  std::string name;
  std::vector<int64_t> shape;
  std::vector<float> host;
  DevBuf<float> dev;
  bool set = false;
  bool no_grad = false;     // exists in the graph / checkpoints but is neither run nor trained
  size_t elems() const {
    size_t e = 1;
    for (auto d : shape) e *= (size_t)d;
    return e;
  }
};

struct ConvCell {           // one ConvLSTMCell: packed kernel + biases
  Param* kernel = nullptr;
  Param* biases = nullptr;
  DevBuf<float> wpack;
  DevBuf<_Float16> wp16;    // f16x3 compute mode: two scaled fp16 planes, fragment order
  DevBuf<float> wx32;       // f16x3, Cx <= 3: the fp32 x chunk scaled by 2^16
  DevBuf<_Float16> wpb;     // bf16 compute mode: one unscaled bf16 plane, fragment order
  DevBuf<float> wx32u;      // bf16, Cx <= 3: the fp32 x chunk, unscaled
  DevBuf<_Float16> wpw;     // f16x3, Winograd F(2,3) form of the kernel (convlstm_wino.h)
  DevBuf<_Float16> wpw3;    // f16x3, Winograd F(3,3) form of the kernel (convlstm_wino3.h)
  bool host_stale = false;  // device copy was updated by the optimizer
  // f16x3: may this kernel take a Winograd form?  A Winograd form spreads EVERY tap over all
  // its components, so an outlier weight (|w| thousands of times the kernel's typical weight)
  // leaves roundoff of ITS size in outputs it does not feed at all -- the direct form keeps it
  // in the outputs that carry it.  Measured (tests/test_gpu_at_size.py, +-230 outliers in
  // kernels of median |w| 3e-3): F(2,3) / F(3,3) 7e-5 / 1.8e-4 of the output range against
  // fp64, direct form 2.6e-5 / 3.7e-5 (fp32 matrix pipe 2.0e-5 / 2.9e-5).  Set by
  // ensure_packed16 from the host copy: max |w| <= kWinoOutlierRatio x median |w|.
  bool wino_numerics_ok = true;
  int Cx = 0;
};
constexpr float kWinoOutlierRatio = 4096.f;

struct KernelStat {
  std::string name;
  int64_t launches = 0;
  // flops: algorithmic FLOPs the launches EXECUTED (a zero-state step skips the h
  // half of the gate convolution); flops_dense: the same steps as the reference
  // computes them (dense 2*M*9*(Cx+C)*4C).  bytes: algorithmic HBM bytes.
  double total_ms = 0, flops = 0, bytes = 0, flops_dense = 0;
  // FLOPs the launches ISSUED to the matrix pipe (0 for non-MFMA kernels): the algorithmic
  // count x 3 for the direct f16x3 form (three fp16 MFMAs per product), x 2 for its Winograd
  // F(2,3) form (two thirds of them), x 1 for the fp32 and bf16 pipes
  double flops_mfma = 0;
};

struct PendingEvent {
  int stat;
  hipEvent_t a, b;
};

struct ScaleState {
  int H = 0, W = 0, K = 0;
  bool use = false;
  ConvCell enc_cls, enc_reg, dec_cls, dec_reg;
  Param *emb_cls_W = nullptr, *emb_cls_b = nullptr, *emb_reg_W = nullptr,
        *emb_reg_b = nullptr, *out_cls_W = nullptr, *out_reg_W = nullptr;
  DevBuf<float> scene_mean;                 // [N, K, D]
  DevBuf<int32_t> labels;                   // [N, T_o]
  DevBuf<float> obs_reg;                    // [N, T_o, K, 2]
  DevBuf<double> centers;                   // [K, 2] cell centres (compact inputs)
  DevBuf<float> cls_c[2], cls_h[2], cls_hg; // class chain state [R, K, C]
  DevBuf<float> reg_c[2], reg_h[2];         // regression chain state [N, K, C]
  DevBuf<float> xbuf_cls, xbuf_reg;         // ConvLSTM x operand
  DevBuf<float> out_cls;                    // [N, T_p, K, 1]
  DevBuf<float> out_reg;                    // [N, T_p, K, 2]
  DevBuf<int32_t> ids;                      // [N] greedy argmax
  // decoder tail (decode_tail.h): per-cell tap products of hidden2grid and its packs
  DevBuf<float> q_cls, q_reg;               // [R, K, 9], [N, K, 18]
  DevBuf<float> wq_cls, wq_reg;             // pack_h2g_kernel of out_cls_W / out_reg_W
  bool wq_valid = false;
  // sparse x operand of the class chains (sparse_x.h)
  DevBuf<uint32_t> sx_cellyx;               // [K] y << 16 | x
  DevBuf<float> sx_dec_bias, sx_dec_corr;   // [9][4C], [9][25][4C]: functions of the weights
  DevBuf<float> sx_enc_corr;                // [T_o][N][9][4C]: every encoder step
  bool sx_valid = false;
};

}  // namespace

struct mv_engine {
  mv_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::vector<std::unique_ptr<Param>> params;
  Param* decode_reg_W = nullptr;   // --use_single_decoder: the offset kernel shared by the scales
  std::map<std::string, Param*> by_name;
  std::vector<Param*> scene_W, scene_b;
  ScaleState sc[MV_MAX_SCALES];
  // inputs
  DevBuf<int32_t> obs_scene;       // [N, T_o]
  DevBuf<float> scene_feat;        // [U, SH, SW, SC]
  DevBuf<uint8_t> scene_u8;        // compact inputs: the masks as uploaded
  DevBuf<double> xy_dev;           // compact inputs: [N, T, 2] coordinates
  DevBuf<float> scene_conv[MV_MAX_SCALES];  // per level [U, h*w, D]
  std::vector<int> conv_h, conv_w;
  int num_frames = 0;
  int pred_len = 0;
  bool inputs_ready = false;
  // beam
  DevBuf<float> bm_logits;         // [T, N, B, K] per-step logits
  DevBuf<int32_t> bm_ids, bm_parents;  // [T, N, B]
  DevBuf<float> bm_lp[2];          // [N, B]
  DevBuf<float> bm_cand;           // [N, B, K] candidate log-probs of one step
  DevBuf<int32_t> bm_src_row;      // [N*B]
  DevBuf<int32_t> bm_ref;          // [N*B] 1 = some surviving beam continues this state row
  DevBuf<int32_t> bm_trace;        // [N, B, T]
  DevBuf<float> bm_out_logits;     // [N, B, T, K]
  // --use_single_decoder with beam search (code/pred_models.py:274, 287-296): the offsets
  // are hidden2grid of the class decoder's states traced back along every beam
  DevBuf<float> bm_reg_steps;      // [T, N*B, K, 2]   per step, in the step's own row order
  DevBuf<float> bm_out_reg;        // [N*B, T, K, 2]   traced back
  DevBuf<int32_t> bm_out_ids;      // [N, B, T]
  // 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = f16x3 split on the fp16 matrix pipe,
  // 2 = bf16 operands / fp32 accumulate (one plane, one MFMA per product)
  int compute_mode = 0;
  DevBuf<_Float16> px16[mv::kMaxGroup], ph16[mv::kMaxGroup];   // fallback plane scratch per slot
  // F(3,3) gate kernel: the pre-transformed operands of a group slot (convlstm_wino3.h
  // wino3_transform_kernel), x and h
  DevBuf<_Float16> pv3x[mv::kMaxGroup], pv3h[mv::kMaxGroup];
  // relu / lrelu models in f16x3 mode: the x operands of the gate convolutions are unbounded,
  // so their planes carry a per-tensor exponent (max |x| as float bits [64] | exponent [1])
  // instead of the fixed 2^8; the producers do not emit planes for these buffers
  DevBuf<int32_t> xexp[mv::kMaxGroup];
  std::set<const float*> xbufs;
  bool dyn_x() const { return compute_mode == 1 && cfg.activation != 0; }
  // planes that travel with an fp32 operand buffer: producers (conv epilogue, graph
  // attention, embeddings) emit them, the next conv launch consumes them
  struct PlaneBuf { _Float16* p; size_t n; bool valid; };
  std::map<const float*, PlaneBuf> planes;
  std::vector<std::unique_ptr<DevBuf<_Float16>>> plane_store;
  // producer side; stride 0 tells the producer kernels to write ONE bf16 plane
  _Float16* plane_out(const float* dst, size_t* stride) {
    if (compute_mode == 0) return nullptr;
    if (dyn_x() && xbufs.count(dst)) return nullptr;
    auto it = planes.find(dst);
    if (it == planes.end()) return nullptr;
    it->second.valid = true;
    *stride = compute_mode == 2 ? 0 : it->second.n;
    return it->second.p;
  }
  void plane_invalidate(const float* dst) {
    auto it = planes.find(dst);
    if (it != planes.end()) it->second.valid = false;
  }
  // hipGraph replay of the forward (one graph per (mode, T_pred, U))
  bool graph_mode = false;
  std::map<std::tuple<int, int, int>, hipGraphExec_t> graphs;
  void drop_graphs() {
    for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
    graphs.clear();
  }
  // pipelined greedy forward (mv_submit_greedy / mv_collect_greedy): feed of batch k+1 and
  // fetch of batch k-1 on a copy stream while batch k computes
  struct PipeSlot {
    void* pin = nullptr;            // pinned host: inputs, then outputs
    char* dev = nullptr;            // device staging, same layout
    size_t in_bytes = 0, out_bytes = 0;
    hipEvent_t h2d = nullptr, done = nullptr, d2h = nullptr;
    int num_frames = 0, pred_len = 0;
    bool busy = false;
  };
  std::vector<PipeSlot> pipe;
  hipStream_t copy_stream = nullptr;     // H2D (feeds)
  hipStream_t fetch_stream = nullptr;    // D2H (fetches): its own queue, else the feed of
                                         // batch k+1 would sit behind the fetch of batch k,
                                         // which waits for batch k's kernels
  size_t pipe_head = 0, pipe_tail = 0;      // next slot to submit into / to collect from
  // in-library gradient all-reduce (mv_allreduce_init, comm.h); null: single device
  mv::Comm* comm = nullptr;
  // training state (mv_train_init)
  mv_train_holder* train = nullptr;
  bool train_packs_valid = false;
  // profiling
  bool profiling = false;
  std::vector<KernelStat> stats;
  std::vector<PendingEvent> pending;

  Param* add_param(const std::string& name, std::vector<int64_t> shape) {
    params.emplace_back(new Param());
    Param* p = params.back().get();
    p->name = name;
    p->shape = std::move(shape);
    by_name[name] = p;
    return p;
  }
  int stat_index(const char* name) {
    for (size_t i = 0; i < stats.size(); ++i)
      if (stats[i].name == name) return (int)i;
    stats.push_back(KernelStat{name});
    return (int)stats.size() - 1;
  }
};

namespace {

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PROCESS-wide property of the kernel
// (per device): it is only ever raised, so a second engine / mv_op_beam_step with a
// smaller beam_size * K cannot lower the limit under a live engine.
void ensure_beam_step_lds(int device, size_t lds) {
  static std::mutex mu;
  static std::map<int, size_t> granted;
  std::lock_guard<std::mutex> lk(mu);
  MV_REQUIRE(lds <= 160 * 1024, "beam_size*K too large for the LDS beam step (%zu B)", lds);
  size_t& cur = granted[device];
  if (lds > cur) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mv::beam_step_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mv::beam_select_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    cur = lds;
  }
}

// One beam step (log-softmax + diversity penalty + top-B).  K <= 1024: the rank count on
// one wave per (n, b) row over the whole chip, then the per-sample selection; larger K
// (or MV_BEAM_STEP=v1): the single-launch kernel.  `cand` = [N*B, K] scratch.
void launch_beam_step(hipStream_t stream, const float* logits, const float* prev_lp,
                      float* cand, int N, int B, int K, int time, int diverse,
                      float log_gamma, int fix_num_timestep, float* new_lp, int32_t* ids,
                      int32_t* parents, int32_t* src_row, int rows_per_sample,
                      int32_t* row_ref = nullptr) {
  static const bool v1 = getenv("MV_BEAM_STEP") && strcmp(getenv("MV_BEAM_STEP"), "v1") == 0;
  if (v1 || K > 64 * mv::kBeamRankJ || !cand) {
    hipLaunchKernelGGL(mv::beam_step_kernel, dim3(N), dim3(512),
                       ((size_t)2 * B * K + 512) * sizeof(float), stream, logits, prev_lp, B,
                       K, time, diverse, log_gamma, fix_num_timestep, new_lp, ids, parents,
                       src_row, rows_per_sample, row_ref);
    HIP_CHECK(hipGetLastError());   // a refused LDS size must not pass silently
    return;
  }
  const int R = N * B;
  const dim3 grid(cdiv((size_t)R, 4)), block(256);
  if (K <= 64 * 3)
    hipLaunchKernelGGL(mv::beam_rank_kernel<3>, grid, block, 0, stream, logits, prev_lp, R, B,
                       K, time, diverse, log_gamma, cand);
  else if (K <= 64 * 9)
    hipLaunchKernelGGL(mv::beam_rank_kernel<9>, grid, block, 0, stream, logits, prev_lp, R, B,
                       K, time, diverse, log_gamma, cand);
  else
    hipLaunchKernelGGL(mv::beam_rank_kernel<mv::kBeamRankJ>, grid, block, 0, stream, logits,
                       prev_lp, R, B, K, time, diverse, log_gamma, cand);
  hipLaunchKernelGGL(mv::beam_select_kernel, dim3(N), dim3(1024),
                     ((size_t)B * K + 64) * sizeof(float), stream, cand, B, K, time,
                     fix_num_timestep, new_lp, ids, parents, src_row, rows_per_sample, row_ref);
  HIP_CHECK(hipGetLastError());
}

// Launch wrapper: optional hipEvent bracket per launch for the roofline figure.
template <typename F>
void launch(mv_engine* e, const char* name, double flops, double bytes, F&& fn,
            double flops_dense = -1.0, double mfma_factor = 0.0) {
  if (!e->profiling) {
    fn();
    return;
  }
  int si = e->stat_index(name);
  PendingEvent pe{si, nullptr, nullptr};
  HIP_CHECK(hipEventCreate(&pe.a));
  HIP_CHECK(hipEventCreate(&pe.b));
  HIP_CHECK(hipEventRecord(pe.a, e->stream));
  fn();
  HIP_CHECK(hipEventRecord(pe.b, e->stream));
  e->stats[si].launches += 1;
  e->stats[si].flops += flops;
  e->stats[si].flops_dense += flops_dense >= 0 ? flops_dense : flops;
  e->stats[si].flops_mfma += mfma_factor * flops;
  e->stats[si].bytes += bytes;
  e->pending.push_back(pe);
}

void drain_events(mv_engine* e) {
  for (auto& pe : e->pending) {
    float ms = 0.f;
    HIP_CHECK(hipEventSynchronize(pe.b));
    HIP_CHECK(hipEventElapsedTime(&ms, pe.a, pe.b));
    e->stats[pe.stat].total_ms += ms;
    (void)hipEventDestroy(pe.a);
    (void)hipEventDestroy(pe.b);
  }
  e->pending.clear();
}

// ------------------------------------------------------------------ setup

void build_param_table(mv_engine* e) {
  const mv_config& c = e->cfg;
  const int64_t C = c.hidden_size, D = c.scene_conv_dim, E = c.emb_size,
                k = c.convlstm_kernel, sk = c.scene_conv_kernel;
  int64_t cin = c.scene_class;
  char nm[256];
  for (int i = 0; i < c.num_scales; ++i) {
    snprintf(nm, sizeof(nm), "person_pred/scene_conv%d/W", i + 1);
    e->scene_W.push_back(e->add_param(nm, {sk, sk, cin, D}));
    snprintf(nm, sizeof(nm), "person_pred/scene_conv%d/b", i + 1);
    e->scene_b.push_back(e->add_param(nm, {D}));
    cin = D;
  }
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    S.H = c.grid_h[s]; S.W = c.grid_w[s]; S.K = S.H * S.W;
    S.use = c.use_grid[s] != 0;
    if (!S.use) continue;
    auto cell = [&](ConvCell& cc, const char* fmt, int64_t Cx) {
      char base[200];
      snprintf(base, sizeof(base), fmt, s, s);
      cc.Cx = (int)Cx;
      cc.kernel = e->add_param(std::string("person_pred/") + base + "/kernel",
                               {k, k, Cx + C, 4 * C});
      cc.biases = e->add_param(std::string("person_pred/") + base + "/biases", {4 * C});
    };
    cell(S.enc_cls, "encoder_grid_class_%d/enc_grid_%d", D);
    cell(S.enc_reg, "encoder_grid_reg_%d/enc_grid_regress_%d", 2);
    cell(S.dec_cls, "decoder_grid_class_%d/decoder_rnn/dec_grid_%d", E);
    snprintf(nm, sizeof(nm), "person_pred/decoder_grid_class_%d/decoder_rnn/grid_emb/W", s);
    S.emb_cls_W = e->add_param(nm, {3, 3, 1, E});
    snprintf(nm, sizeof(nm), "person_pred/decoder_grid_class_%d/decoder_rnn/grid_emb/b", s);
    S.emb_cls_b = e->add_param(nm, {E});
    snprintf(nm, sizeof(nm), "person_pred/hidden2grid_decoder_grid_class_%d/out_dec_grid/W", s);
    S.out_cls_W = e->add_param(nm, {3, 3, C, 1});
    if (c.use_single_decoder) {
      // --use_single_decoder (code/pred_models.py:287-296): no regression decoder; ONE
      // offset kernel for all scales (scope "decode_reg" has no scale index).  The
      // regression encoder is built by the reference but feeds nothing: its variables
      // exist (checkpoints hold them), it is not run and not trained.
      S.enc_reg.kernel->no_grad = S.enc_reg.biases->no_grad = true;
      if (!e->decode_reg_W)
        e->decode_reg_W = e->add_param("person_pred/decode_reg/out_dec_grid/W", {3, 3, C, 2});
      S.out_reg_W = e->decode_reg_W;
      continue;
    }
    cell(S.dec_reg, "decoder_grid_reg_%d/decoder_rnn/dec_grid_reg_%d", E);
    snprintf(nm, sizeof(nm), "person_pred/decoder_grid_reg_%d/decoder_rnn/grid_emb/W", s);
    S.emb_reg_W = e->add_param(nm, {3, 3, 2, E});
    snprintf(nm, sizeof(nm), "person_pred/decoder_grid_reg_%d/decoder_rnn/grid_emb/b", s);
    S.emb_reg_b = e->add_param(nm, {E});
    snprintf(nm, sizeof(nm), "person_pred/hidden2grid_decoder_grid_reg_%d/out_dec_grid/W", s);
    S.out_reg_W = e->add_param(nm, {3, 3, C, 2});
  }
}

// the ConvLSTM cells a forward / training step of this engine runs
std::vector<ConvCell*> active_cells(const mv_engine* e, ScaleState& S) {
  if (e->cfg.use_single_decoder) return {&S.enc_cls, &S.dec_cls};
  return {&S.enc_cls, &S.enc_reg, &S.dec_cls, &S.dec_reg};
}

void validate_config(const mv_config& c) {
  MV_REQUIRE(c.abi_version == MV_ABI_VERSION, "mv_config.abi_version %d != %d",
             c.abi_version, MV_ABI_VERSION);
  MV_REQUIRE(c.activation >= 0 && c.activation <= 2, "mv_config.activation %d (0 tanh, 1 relu, "
             "2 lrelu; reference code/pred_utils.py:86-94)", c.activation);
  MV_REQUIRE(c.batch_size > 0 && c.obs_len > 0 && c.max_pred_len > 0,
             "batch_size/obs_len/max_pred_len must be positive");
  MV_REQUIRE(c.num_scales >= 1 && c.num_scales <= MV_MAX_SCALES,
             "num_scales %d not in [1,%d]", c.num_scales, MV_MAX_SCALES);
  // --enc_hidden_size / --dec_hidden_size (code/train.py:54-57; one value for both, as the
  // reference's own graph requires): whole 128-column blocks of the gate GEMMs and of the
  // f16x3 wgrad tile; the one-wave-per-cell kernels take up to two 256-channel groups
  MV_REQUIRE(c.hidden_size == 128 || c.hidden_size == 256 || c.hidden_size == 512,
             "hidden_size %d unsupported (128, 256 or 512)", c.hidden_size);
  // --convlstm_kernel (code/train.py:70): 3 runs the matrix-pipe kernels; any other size runs
  // the plain fp32 loops of csrc/convlstm_generic.h (compute mode 0 only: slow, but it runs)
  MV_REQUIRE(c.convlstm_kernel >= 1 && c.convlstm_kernel <= 9,
             "convlstm_kernel %d unsupported (1 .. 9)", c.convlstm_kernel);
  // --scene_conv_dim (code/train.py:69): whole 32-channel chunks of the class encoder's x
  // operand; above 64 the graph attention takes its one-wave-per-cell form (two scene
  // channels per lane) and the class encoder its dense x operand
  MV_REQUIRE(c.scene_conv_dim > 0 && c.scene_conv_dim <= 128 &&
             mv::convlstm_cx_supported(c.scene_conv_dim),
             "scene_conv_dim %d unsupported (a multiple of 32 up to 128)", c.scene_conv_dim);
  // the decoders' x operand: whole 32-channel chunks of the gate GEMM, 16-byte plane vectors
  // and the decode tail's LDS (decode_tail.h) -- checked here, not at the first decode step
  MV_REQUIRE(c.emb_size >= 32 && c.emb_size % 32 == 0 && c.emb_size <= 512 &&
             mv::convlstm_cx_supported(c.emb_size),
             "emb_size %d unsupported (a multiple of 32 up to 512)", c.emb_size);
  MV_REQUIRE(c.beam_size >= 1, "beam_size must be >= 1");
  MV_REQUIRE(!(c.class_feedback_dense && c.beam_size > 1), "class_feedback_dense: greedy only "
             "(grid_decoder_beam_search always feeds one-hot ids)");
  int hh = c.scene_h, ww = c.scene_w, used = 0;
  for (int s = 0; s < c.num_scales; ++s) {
    hh = (hh + 1) / 2; ww = (ww + 1) / 2;   // stride-2 SAME conv chain
    // process_args' round() and the conv chain's ceil()
    // must agree (true for strides 2,4 on 36x64).
    MV_REQUIRE(c.grid_h[s] == hh && c.grid_w[s] == ww,
               "scene_grids[%d] = %dx%d does not match the stride-2 conv chain "
               "(%dx%d); only scene_grid_strides 2,4,.. are supported",
               s, c.grid_h[s], c.grid_w[s], hh, ww);
    used += c.use_grid[s] != 0;
  }
  MV_REQUIRE(used >= 1, "no grid scale enabled");
  if (c.beam_size > 1)
    MV_REQUIRE(used == 1, "beam search: only one scale at a time "
               "(reference pred_models.py:262)");
}

void alloc_buffers(mv_engine* e) {
  const mv_config& c = e->cfg;
  const size_t N = c.batch_size, T = c.obs_len, Tp = c.max_pred_len,
               C = c.hidden_size, D = c.scene_conv_dim, B = c.beam_size;
  const size_t maxU = N * T;
  e->obs_scene.alloc(N * T);
  e->scene_feat.alloc(maxU * c.scene_h * c.scene_w * c.scene_class);
  int hh = c.scene_h, ww = c.scene_w;
  for (int i = 0; i < c.num_scales; ++i) {
    hh = (hh + 1) / 2; ww = (ww + 1) / 2;
    e->conv_h.push_back(hh); e->conv_w.push_back(ww);
    e->scene_conv[i].alloc(maxU * hh * ww * D);
  }
  const size_t xc = (size_t)std::max((int)D, c.emb_size);
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    const size_t K = S.K, R = N * B;
    S.scene_mean.alloc(N * K * D);
    S.labels.alloc(N * T);
    S.obs_reg.alloc(N * T * K * 2);
    for (int i = 0; i < 2; ++i) {
      S.cls_c[i].alloc(R * K * C); S.cls_h[i].alloc(R * K * C);
      S.reg_c[i].alloc(N * K * C); S.reg_h[i].alloc(N * K * C);
    }
    if (c.use_gnn) S.cls_hg.alloc(R * K * C);
    S.xbuf_cls.alloc(R * K * xc);
    S.xbuf_reg.alloc(N * K * xc);
    S.out_cls.alloc(N * Tp * K);
    S.out_reg.alloc(N * Tp * K * 2);
    S.ids.alloc(R);
    // single decoder + beam: the offsets are decoded from all N*B state rows
    S.q_cls.alloc(R * K * 9); S.q_reg.alloc((c.use_single_decoder ? R : N) * K * 18);
    S.wq_cls.alloc(C * 32); S.wq_reg.alloc(C * 32);
    S.sx_cellyx.alloc(K);
    S.sx_dec_bias.alloc(9 * 4 * C); S.sx_dec_corr.alloc(9 * 25 * 4 * C);
    S.sx_enc_corr.alloc(T * N * 9 * 4 * C);
    if (B > 1) {
      e->bm_logits.alloc(Tp * R * K);
      e->bm_ids.alloc(Tp * R);
      e->bm_parents.alloc(Tp * R);
      e->bm_lp[0].alloc(R); e->bm_lp[1].alloc(R);
      e->bm_cand.alloc((size_t)R * K);
      e->bm_src_row.alloc(R);
      e->bm_ref.alloc(R);
      e->bm_trace.alloc(R * Tp);
      e->bm_out_logits.alloc(R * Tp * K);
      e->bm_out_ids.alloc(R * Tp);
      if (c.use_single_decoder) {
        e->bm_reg_steps.alloc(Tp * R * K * 2);
        e->bm_out_reg.alloc(R * Tp * K * 2);
      }
    }
  }
}

void ensure_packed(mv_engine* e, ConvCell& cc) {
  MV_REQUIRE(cc.kernel->set, "parameter %s not set", cc.kernel->name.c_str());
  MV_REQUIRE(cc.biases->set, "parameter %s not set", cc.biases->name.c_str());
  if (cc.wpack.p) return;
  if (e->cfg.convlstm_kernel != 3) return;      // generic taps: straight from the HWIO kernel
  const int C = e->cfg.hidden_size;
  std::vector<float> packed(mv::convlstm_wpack_elems(cc.Cx, C));
  mv::pack_convlstm_weights(cc.kernel->host.data(), cc.Cx, C, packed.data());
  cc.wpack.alloc(packed.size());
  HIP_CHECK(hipMemcpy(cc.wpack.p, packed.data(), packed.size() * sizeof(float),
                      hipMemcpyHostToDevice));
}

static bool C_multiple_ok(const mv_engine* e, const ConvCell& cc) {
  return e->cfg.hidden_size % mv::kWnCh == 0 &&
         (cc.Cx % 16 == 0 || (cc.Cx > 0 && 9 * cc.Cx <= mv::kBK));
}
// f16x3 packs (two scaled fp16 planes in fragment order; the fp32 x chunk of the
// 2-channel regression-encoder input scaled by 2^16), from the CURRENT weights.
void ensure_packed16(mv_engine* e, ConvCell& cc) {
  if (cc.wp16.p) return;
  const int C = e->cfg.hidden_size;
  MV_REQUIRE(mv::f16x3_cx_supported(cc.Cx), "f16x3: Cx %d unsupported", cc.Cx);
  if (cc.host_stale) {
    HIP_CHECK(hipMemcpy(cc.kernel->host.data(), cc.kernel->dev.p,
                        cc.kernel->elems() * sizeof(float), hipMemcpyDeviceToHost));
    cc.host_stale = false;
  }
  {   // 256 w must stay inside fp16 (|w| < 255): true of any sane checkpoint, checked anyway
    float mx = 0.f;
    for (float v : cc.kernel->host) mx = std::max(mx, std::fabs(v));
    // the Winograd packs store TRANSFORMED kernel rows -- (g0 +- g1 + g2) / 2 (F(2,3));
    // (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 6, (g0 + 2 g1 + 4 g2) / 6 (F(3,3)); the dgrad pack the
    // same of the flipped rows -- which reach up to 1.5 max |w| when three taps of one column
    // share a sign: the bound is taken on THOSE values, column by column
    float reach = mx;
    if (mv::wino_enabled() && C_multiple_ok(e, cc)) {
      const size_t Cin = (size_t)cc.Cx + C, N4 = 4 * (size_t)C;
      const float* w = cc.kernel->host.data();
      for (size_t dxc = 0; dxc < 3 * Cin; ++dxc) {          // (dx, input channel) pairs
        const size_t dx = dxc / Cin, ci = dxc - dx * Cin;
        const float* g0 = w + ((0 * 3 + dx) * Cin + ci) * N4;
        const float* g1 = w + ((1 * 3 + dx) * Cin + ci) * N4;
        const float* g2 = w + ((2 * 3 + dx) * Cin + ci) * N4;
        for (size_t n = 0; n < N4; ++n) {
          const float a = g0[n], b = g1[n], c2 = g2[n];
          const float t = std::max(std::max(std::fabs(a + b + c2), std::fabs(a - b + c2)) * 0.5f,
                                   std::max(std::fabs(a + 2.f * b + 4.f * c2),
                                            std::fabs(4.f * a + 2.f * b + c2)) * (1.f / 6.f));
          reach = std::max(reach, t);
        }
      }
    }
    MV_REQUIRE(reach * mv::kF16Scale < 60000.f, "f16x3: |%s| reaches %g (%g in the transformed "
               "kernel planes of the Winograd forms), outside the scaled fp16 range; use compute "
               "mode f32", cc.kernel->name.c_str(), mx, reach);
    // outlier test for the Winograd forms (ConvCell::wino_numerics_ok): median of |w|
    std::vector<float> mag(cc.kernel->host.size());
    for (size_t i = 0; i < mag.size(); ++i) mag[i] = std::fabs(cc.kernel->host[i]);
    std::nth_element(mag.begin(), mag.begin() + mag.size() / 2, mag.end());
    const float med = mag[mag.size() / 2];
    const bool ok = mx <= kWinoOutlierRatio * med;
    if (!ok && cc.wino_numerics_ok)
      fprintf(stderr, "[multiverse_hip] f16x3: %s has max |w| %g at a median |w| of %g (> %g x): "
              "its gate convolution takes the direct 3x3 form instead of a Winograd form "
              "(roundoff of outlier weights would reach unrelated outputs)\n",
              cc.kernel->name.c_str(), mx, med, kWinoOutlierRatio);
    if (ok != cc.wino_numerics_ok) { cc.wpw.release(); cc.wpw3.release(); }
    cc.wino_numerics_ok = ok;
  }
  const bool small = cc.Cx > 0 && 9 * cc.Cx <= mv::kBK;
  // the h part (and an x part that is a multiple of 16 channels) as fp16 planes
  const int Cx16 = small ? 0 : cc.Cx;
  std::vector<_Float16> p16(mv::f16x3_wpack_elems(Cx16, C));
  if (small) {   // drop the x channels: pack a view of the kernel without them
    const int Cin = cc.Cx + C, N4 = 4 * C;
    std::vector<float> wh((size_t)9 * C * N4);
    for (int t = 0; t < 9; ++t)
      memcpy(&wh[(size_t)t * C * N4], &cc.kernel->host[((size_t)t * Cin + cc.Cx) * N4],
             (size_t)C * N4 * sizeof(float));
    mv::pack_f16x3_weights(wh.data(), 0, C, p16.data());
    std::vector<float> packed(mv::convlstm_wpack_elems(cc.Cx, C));
    mv::pack_convlstm_weights(cc.kernel->host.data(), cc.Cx, C, packed.data());
    const int nch = mv::convlstm_xchunks(cc.Cx) + 9 * (C / mv::kBK);
    std::vector<float> wx((size_t)(C / mv::kChBlock) * mv::kBN * mv::kBK);
    for (int cb = 0; cb < C / mv::kChBlock; ++cb)
      for (int i = 0; i < mv::kBN * mv::kBK; ++i)
        wx[(size_t)cb * mv::kBN * mv::kBK + i] =
            packed[((size_t)cb * nch + 0) * mv::kBN * mv::kBK + i] * 65536.0f;
    cc.wx32.alloc(wx.size());
    HIP_CHECK(hipMemcpy(cc.wx32.p, wx.data(), wx.size() * sizeof(float),
                        hipMemcpyHostToDevice));
  } else {
    mv::pack_f16x3_weights(cc.kernel->host.data(), cc.Cx, C, p16.data());
  }
  cc.wp16.alloc(p16.size());
  HIP_CHECK(hipMemcpy(cc.wp16.p, p16.data(), p16.size() * sizeof(_Float16),
                      hipMemcpyHostToDevice));
}

// Winograd F(2,3) pack of the f16x3 forward (convlstm_wino.h), from the CURRENT device
// weights; the transform of the kernel rows runs in fp64 on the device.
void pack_wino(mv_engine* e, ConvCell& cc) {
  const int C = e->cfg.hidden_size;
  const bool small = cc.Cx > 0 && 9 * cc.Cx <= mv::kBK;
  const int Cx16 = small ? 0 : cc.Cx;
  const size_t halves = mv::wino_wpack_elems(Cx16, C);
  mv::wino_init_attributes();
  cc.wpw.alloc(halves);
  const size_t threads = halves / 2;
  hipLaunchKernelGGL(mv::pack_wino_kernel, dim3(cdiv(threads, 256)), dim3(256), 0, e->stream,
                     cc.kernel->dev.p, cc.wpw.p, cc.Cx, Cx16, C, threads);
}
// Winograd F(3,3) pack (convlstm_wino3.h), likewise from the CURRENT device weights.
void pack_wino3(mv_engine* e, ConvCell& cc) {
  const int C = e->cfg.hidden_size;
  const bool small = cc.Cx > 0 && 9 * cc.Cx <= mv::kBK;
  const int Cx16 = small ? 0 : cc.Cx;
  const size_t halves = mv::wino3_wpack_elems(Cx16, C, mv::kW3Nrb);
  mv::wino3_init_attributes();
  cc.wpw3.alloc(halves);
  const size_t threads = halves / 2;
  hipLaunchKernelGGL(mv::pack_wino3_kernel, dim3(cdiv(threads, 256)), dim3(256), 0, e->stream,
                     cc.kernel->dev.p, cc.wpw3.p, cc.Cx, Cx16, C, mv::kW3Nrb, threads);
}
// every weight-mutating path comes through here (or releases both packs): a non-null pack is by
// construction a pack of the CURRENT weights -- re-packed in place when its form is enabled (no
// hipFree / hipMalloc per training step: they synchronise the device), released otherwise
void pack_wino_forms(mv_engine* e, ConvCell& cc) {
  const bool on = mv::wino_enabled() && C_multiple_ok(e, cc) && cc.wino_numerics_ok;
  if (on) pack_wino(e, cc); else cc.wpw.release();
  if (on && mv::wino3_enabled()) pack_wino3(e, cc); else cc.wpw3.release();
}
void ensure_packed_wino(mv_engine* e, ConvCell& cc) {
  if (cc.wpw.p) return;
  pack_wino_forms(e, cc);
}

// bf16 packs (one unscaled plane; the 2-channel regression-encoder input keeps its fp32
// chunk), from the CURRENT weights.
void ensure_packed_bf16(mv_engine* e, ConvCell& cc) {
  if (cc.wpb.p) return;
  const int C = e->cfg.hidden_size;
  MV_REQUIRE(mv::f16x3_cx_supported(cc.Cx), "bf16: Cx %d unsupported", cc.Cx);
  if (cc.host_stale) {
    HIP_CHECK(hipMemcpy(cc.kernel->host.data(), cc.kernel->dev.p,
                        cc.kernel->elems() * sizeof(float), hipMemcpyDeviceToHost));
    cc.host_stale = false;
  }
  const bool small = cc.Cx > 0 && 9 * cc.Cx <= mv::kBK;
  const int Cx16 = small ? 0 : cc.Cx;
  std::vector<_Float16> pb(mv::bf16_wpack_elems(Cx16, C));
  if (small) {
    const int Cin = cc.Cx + C, N4 = 4 * C;
    std::vector<float> wh((size_t)9 * C * N4);
    for (int t = 0; t < 9; ++t)
      memcpy(&wh[(size_t)t * C * N4], &cc.kernel->host[((size_t)t * Cin + cc.Cx) * N4],
             (size_t)C * N4 * sizeof(float));
    mv::pack_bf16_weights(wh.data(), 0, C, pb.data());
    std::vector<float> packed(mv::convlstm_wpack_elems(cc.Cx, C));
    mv::pack_convlstm_weights(cc.kernel->host.data(), cc.Cx, C, packed.data());
    const int nch = mv::convlstm_xchunks(cc.Cx) + 9 * (C / mv::kBK);
    std::vector<float> wx((size_t)(C / mv::kChBlock) * mv::kBN * mv::kBK);
    for (int cb = 0; cb < C / mv::kChBlock; ++cb)
      memcpy(&wx[(size_t)cb * mv::kBN * mv::kBK],
             &packed[((size_t)cb * nch + 0) * mv::kBN * mv::kBK],
             (size_t)mv::kBN * mv::kBK * sizeof(float));
    cc.wx32u.alloc(wx.size());
    HIP_CHECK(hipMemcpy(cc.wx32u.p, wx.data(), wx.size() * sizeof(float),
                        hipMemcpyHostToDevice));
  } else {
    mv::pack_bf16_weights(cc.kernel->host.data(), cc.Cx, C, pb.data());
  }
  cc.wpb.alloc(pb.size());
  HIP_CHECK(hipMemcpy(cc.wpb.p, pb.data(), pb.size() * sizeof(_Float16),
                      hipMemcpyHostToDevice));
}

// scene channels the graph attention sees: all of them, except in the greedy decoder of the
// SimAug fork's graph (mv_config.simaug_graph), which attends over the hidden state alone
int gnn_scene_dim(const mv_engine* e) {
  return (e->cfg.simaug_graph && e->cfg.beam_size == 1) ? 0 : e->cfg.scene_conv_dim;
}

bool sparse_x_on(const mv_engine* e, const ScaleState& S) {
  static const bool off = getenv("MV_SPARSE_X") && atoi(getenv("MV_SPARSE_X")) == 0;
  const mv_config& c = e->cfg;
  return !off && e->compute_mode != 0 && !e->train && S.use && S.H >= 3 && S.W >= 3 &&
         S.dec_cls.Cx == c.emb_size && c.emb_size % 16 == 0 && c.scene_conv_dim % 16 == 0 &&
         c.scene_conv_dim <= 64;
}

void ensure_params(mv_engine* e) {
  for (auto& p : e->params)
    MV_REQUIRE(p->set, "parameter %s not set (mv_set_param)", p->name.c_str());
  for (int s = 0; s < e->cfg.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    for (ConvCell* cc : active_cells(e, S)) ensure_packed(e, *cc);
    if (e->compute_mode == 1)
      for (ConvCell* cc : active_cells(e, S)) ensure_packed16(e, *cc);
    if (e->compute_mode == 1 && mv::wino_enabled())
      for (ConvCell* cc : active_cells(e, S)) ensure_packed_wino(e, *cc);
    if (e->compute_mode == 2)
      for (ConvCell* cc : active_cells(e, S)) ensure_packed_bf16(e, *cc);
    if (!S.wq_valid) {     // hidden2grid tap packs, from the CURRENT device weights
      const int C = e->cfg.hidden_size;
      hipLaunchKernelGGL(mv::pack_h2g_kernel, dim3(cdiv((size_t)C * 32, 256)), dim3(256), 0,
                         e->stream, S.out_cls_W->dev.p, S.wq_cls.p, C, 1);
      hipLaunchKernelGGL(mv::pack_h2g_kernel, dim3(cdiv((size_t)C * 32, 256)), dim3(256), 0,
                         e->stream, S.out_reg_W->dev.p, S.wq_reg.p, C, 2);
      S.wq_valid = true;
    }
    if (!S.sx_valid && sparse_x_on(e, S)) {
      const int C = e->cfg.hidden_size;
      hipLaunchKernelGGL(mv::cell_yx_kernel, dim3(cdiv((size_t)S.K, 256)), dim3(256), 0,
                         e->stream, S.sx_cellyx.p, S.H, S.W);
      hipLaunchKernelGGL(mv::sx_decoder_tables_kernel,
                         dim3(cdiv((size_t)(9 + 9 * 25) * 4 * C, 256)), dim3(256), 0, e->stream,
                         S.dec_cls.kernel->dev.p, S.dec_cls.biases->dev.p, S.emb_cls_W->dev.p,
                         S.emb_cls_b->dev.p, S.dec_cls.Cx, C, S.sx_dec_bias.p, S.sx_dec_corr.p,
                         e->cfg.activation);
      S.sx_valid = true;
    }
  }
}

// ------------------------------------------------------------------ launches

using mv::ConvLstmArgs;

// class-chain x operands as table terms (sparse_x.h): f16x3 / bf16 inference engines only
// (the training forward keeps the dense x: the backward pass needs it).  MV_SPARSE_X=0
// restores the dense operand for A/B runs.
void set_sparse_x(mv_engine* e, ScaleState& S, ConvLstmArgs& a, bool decoder,
                  const int32_t* hot, int hot_stride, int hot_div) {
  a.sx_bias = decoder ? S.sx_dec_bias.p : nullptr;
  a.sx_corr = decoder ? S.sx_dec_corr.p : S.sx_enc_corr.p;
  a.sx_hot = hot; a.sx_hot_stride = hot_stride; a.sx_hot_div = hot_div;
  a.sx_cellyx = S.sx_cellyx.p;
  a.sx_rad = decoder ? 2 : 1;
  a.sx_by_class = decoder ? 1 : 0;
}

ConvLstmArgs conv_problem(mv_engine* e, const ConvCell& cc, const float* x,
                          const float* h, const float* c, const int32_t* src_row_h,
                          const int32_t* src_row_c, float* h_out, float* c_out,
                          int rows, int H, int W, bool zero_state,
                          size_t x_row_stride = 0, bool want_h16 = true) {
  ConvLstmArgs a{};
  // the kernel forms element offsets in 32-bit registers
  const size_t xrs = x_row_stride ? x_row_stride : (size_t)H * W * cc.Cx;
  MV_REQUIRE((size_t)rows * H * W * e->cfg.hidden_size < ((size_t)1 << 31) &&
             (size_t)rows * xrs < ((size_t)1 << 31),
             "ConvLSTM state of %d rows exceeds the 2^31-element addressing of one "
             "launch; lower batch_size x beam_size", rows);
  a.x_row_stride = (int32_t)xrs;
  a.x = x; a.h = h; a.c = c; a.src_row_h = src_row_h; a.src_row_c = src_row_c;
  a.wpack = cc.wpack.p; a.bias = cc.biases->dev.p;
  a.h_out = h_out; a.c_out = c_out;
  a.rows = rows; a.H = H; a.W = W; a.Cx = cc.Cx; a.C = e->cfg.hidden_size;
  mv::convlstm_finish_args(a, zero_state);
  a.want_h16 = want_h16 ? 1 : 0;
  return a;
}

ConvCell* cell_of_bias(mv_engine* e, const float* bias) {
  for (int s = 0; s < e->cfg.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;                  // an unused scale's cells have no parameters
    for (ConvCell* cc : active_cells(e, S))
      if (cc->biases && cc->biases->dev.p == bias) return cc;
  }
  throw HipError{"internal: unknown ConvLSTM cell"};
}

ConvCell* cell_of_pack(mv_engine* e, const float* wpack) {
  for (int s = 0; s < e->cfg.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    for (ConvCell* cc : active_cells(e, S))
      if (cc->wpack.p == wpack) return cc;
  }
  throw HipError{"internal: unknown weight pack"};
}

// f16x3 compute mode: split the fp32 operands of every problem into two scaled
// fp16 planes (HBM-bound, ~2 % of the step), then one grouped launch of the
// fp16-MFMA kernel.
void run_conv_group_f16x3(mv_engine* e, const std::vector<ConvLstmArgs>& probs,
                          double flops, double bytes, double dense) {
  std::vector<mv::ConvLstm16Args> p16(probs.size());
  struct SplitItem { const float* in; _Float16* p0; _Float16* p1; int cells, C; };
  std::vector<SplitItem> splits;
  struct DynItem { const float* in; _Float16* p0; _Float16* p1; int cells, C; int32_t* bits; };
  std::vector<DynItem> dyn_splits;
  const bool bf16 = e->compute_mode == 2;
  for (size_t i = 0; i < probs.size(); ++i) {
    const ConvLstmArgs& a = probs[i];
    ConvCell* cc = cell_of_pack(e, a.wpack);
    mv::ConvLstm16Args& q = p16[i];
    // the kernel's epilogue lets a 32-cell wave tile span at most two images
    MV_REQUIRE(a.H * a.W >= 32, "f16x3 / bf16 compute modes need grids of at least 32 cells "
               "(%d x %d); use compute mode f32", a.H, a.W);
    MV_REQUIRE((double)a.rows * a.H * a.W * a.C * 4.0 < 4294967296.0,
               "f16x3 / bf16 compute modes: state tensor of %d rows exceeds 4 GB", a.rows);
    // the training forward stores the four gate activations [M][4C] through a buffer
    // resource with 32-bit byte offsets (convlstm_f16x3.h epilogue: num_records = 16 M C)
    MV_REQUIRE(!a.gates_out || (double)a.rows * a.H * a.W * a.C * 16.0 < 4294967296.0,
               "f16x3 / bf16 training forward: gate tensor of %d rows exceeds 4 GB "
               "(lower the per-GPU batch or use compute mode f32)", a.rows);
    q.f = a;
    q.wp16 = bf16 ? cc->wpb.p : cc->wp16.p;
    q.wx32 = bf16 ? cc->wx32u.p : cc->wx32.p;
    const size_t cells = (size_t)a.rows * a.H * a.W;
    q.n_xk = a.x_small ? 0 : mv::f16x3_xksteps(a.Cx);
    q.n_hk = a.zero_state ? 0 : 9 * (a.C / 16);
    q.w_ksteps = mv::f16x3_xksteps(a.Cx) + 9 * (a.C / 16);
    if (a.x_small) q.w_ksteps = 9 * (a.C / 16);
    if (bf16)       // an LDS stage of the bf16 kernel holds MV_BF16_UNITS row units of 3 k-steps
      MV_REQUIRE((q.n_xk / 3) % MV_BF16_UNITS == 0 && (q.n_hk / 3) % MV_BF16_UNITS == 0,
                 "bf16 mode: %d x / %d h k-steps do not fill whole LDS stages (emb_size and "
                 "scene_conv_dim must be multiples of 32)", q.n_xk, q.n_hk);
    q.x16 = nullptr; q.h16 = nullptr;
    q.x_plane_stride = q.h_plane_stride = 0;
    // The conv epilogue emits the operand planes of h' (assembled per wave in LDS,
    // 16-byte stores) when the next consumer of h' is a gate convolution; MV_EPI_PLANES=0
    // falls back to the separate split pass over the fp32 h'.
    static const bool epi = !(getenv("MV_EPI_PLANES") && atoi(getenv("MV_EPI_PLANES")) == 0);
    q.h16_out = nullptr;
    q.h16_out_stride = 0;
    e->plane_invalidate(a.h_out);
    if (epi && a.want_h16 && !a.gates_out) {
      size_t pst = 0;
      if (_Float16* po = e->plane_out(a.h_out, &pst)) {   // marks the planes valid
        q.h16_out = po;
        q.h16_out_stride = (int64_t)pst;
      }
    }
    q.f.skip_h32 = (a.skip_h32 && q.h16_out && !e->train) ? 1 : 0;
    auto ready = [&](const float* src) -> const mv_engine::PlaneBuf* {
      auto it = e->planes.find(src);
      return (it != e->planes.end() && it->second.valid) ? &it->second : nullptr;
    };
    if (!a.x_small && a.Cx > 0 && !a.sx_corr) {
      MV_REQUIRE((size_t)a.x_row_stride == (size_t)a.H * a.W * a.Cx,
                 "internal: f16x3 needs a contiguous x operand");
      const size_t n = cells * a.Cx;
      if (const auto* pb = ready(a.x)) {
        q.x16 = pb->p; q.x_plane_stride = (int64_t)pb->n;
      } else {
      const size_t pst = n + mv::kPlaneSlack + mv::kPlanePad;
      MV_REQUIRE(e->px16[i].n >= 2 * pst, "internal: f16x3 x plane scratch");
      _Float16* p0 = e->px16[i].p + mv::kPlanePad;
      q.x16 = p0; q.x_plane_stride = (int64_t)pst;
      if (e->dyn_x()) {       // unbounded activations: planes of 2^e x, e from max |x|
        MV_REQUIRE(e->xexp[i].p, "internal: x exponent scratch");
        q.x_exp = e->xexp[i].p + 64;
        dyn_splits.push_back(DynItem{a.x, p0, p0 + pst, (int)cells, a.Cx, e->xexp[i].p});
      } else {
      splits.push_back(SplitItem{a.x, p0, bf16 ? (_Float16*)nullptr : p0 + pst, (int)cells, a.Cx});
      }
      }
    }
    if (!a.zero_state) {
      const size_t n = cells * a.C;     // source rows == rows (beam: permuted, same count)
      if (const auto* pb = ready(a.h)) {
        q.h16 = pb->p; q.h_plane_stride = (int64_t)pb->n;
      } else {
      const size_t pst = n + mv::kPlaneSlack + mv::kPlanePad;
      MV_REQUIRE(e->ph16[i].n >= 2 * pst, "internal: f16x3 h plane scratch");
      _Float16* p0 = e->ph16[i].p + mv::kPlanePad;
      q.h16 = p0; q.h_plane_stride = (int64_t)pst;
      splits.push_back(SplitItem{a.h, p0, bf16 ? (_Float16*)nullptr : p0 + pst, (int)cells, a.C});
      }
    }
  }
  for (const DynItem& it : dyn_splits) {
    launch(e, "split_planes", 0, 12.0 * (double)it.cells * it.C, [&] {
      HIP_CHECK(hipMemsetAsync(it.bits, 0, 64 * sizeof(int32_t), e->stream));
      hipLaunchKernelGGL(mv::absmax_bits_kernel, dim3(256), dim3(256), 0, e->stream, it.in,
                         (size_t)it.cells * it.C, it.bits);
      hipLaunchKernelGGL(mv::split_planes_dyn_kernel,
                         dim3(mv::split_planes_blocks((size_t)it.cells, it.C)), dim3(256), 0,
                         e->stream, it.in, it.p0, it.p1, it.cells, it.C, it.bits, it.bits + 64);
    });
  }
  // operands no producer left as planes: one grouped split launch in front of the gate kernel
  for (size_t s0 = 0; s0 < splits.size(); s0 += mv::kSplitGroup) {
    mv::SplitGroup g{};
    double sbytes = 0;
    unsigned nb = 0;
    g.n = (int)std::min<size_t>(mv::kSplitGroup, splits.size() - s0);
    for (int j = 0; j < g.n; ++j) {
      const SplitItem& it = splits[s0 + j];
      g.in[j] = it.in; g.p0[j] = it.p0; g.p1[j] = it.p1; g.M[j] = it.cells; g.C[j] = it.C;
      nb += mv::split_planes_blocks((size_t)it.cells, it.C);
      g.blk_end[j] = nb;
      sbytes += (bf16 ? 6.0 : 8.0) * (double)it.cells * it.C;
    }
    launch(e, "split_planes", 0, sbytes, [&] {
      hipLaunchKernelGGL(mv::split_planes_group_kernel, dim3(nb), dim3(256), 0, e->stream, g);
    });
  }
  // f16x3: the Winograd F(2,3) form of the same step (two thirds of the MFMAs,
  // convlstm_wino.h) whenever every problem of the group fits its tiling
  const bool wino_mode = e->compute_mode == 1 && mv::wino_enabled();
  bool wino = wino_mode;
  std::vector<mv::ConvLstmWinoArgs> pw(p16.size());
  for (size_t i = 0; i < p16.size(); ++i) {
    ConvCell* cc = cell_of_pack(e, probs[i].wpack);
    if (!mv::wino_geometry_ok(p16[i].f) || !cc->wpw.p) wino = false;
    pw[i].b = p16[i];
    pw[i].wpw = cc->wpw.p;
    pw[i].w_hwio = cc->kernel->dev.p;
    pw[i].n_xc = p16[i].f.x_small ? 0 : p16[i].f.Cx / 16;
  }
  // ... and its F(3,3) form (five ninths, convlstm_wino3.h) when every problem fits THAT tiling
  // (any grid width: widths that do not divide 32 take its halo tiling) and the slots' buffers
  // hold the pre-transformed operands.  The input transform runs ONCE per operand, in a
  // pre-pass, instead of in every one of the C / 16 column-block workgroups of the gate kernel.
  bool wino3 = wino_mode && mv::wino3_enabled();
  for (size_t i = 0; i < p16.size() && wino3; ++i) {
    const ConvLstmArgs& a = p16[i].f;
    ConvCell* cc = cell_of_pack(e, probs[i].wpack);
    if (!mv::wino3_geometry_ok(a, p16[i]) || !cc->wpw3.p) wino3 = false;
    else if (!mv::wino3_halo_addressable(a)) wino3 = false;   // 32-bit lane offsets (HALO)
    else if (!a.zero_state && e->pv3h[i].n < mv::wino3_v_elems(a.rows, a.H, a.W, a.C)) wino3 = false;
    else if (!a.x_small && a.Cx > 0 && !a.sx_corr &&
             e->pv3x[i].n < mv::wino3_v_elems(a.rows, a.H, a.W, a.Cx)) wino3 = false;
  }
  std::vector<mv::Wn3TransformItem> tr3;
  double tr3_bytes = 0;
  if (wino3) {
    for (size_t i = 0; i < p16.size(); ++i) {
      const mv::ConvLstm16Args& q = p16[i];
      const ConvLstmArgs& a = q.f;
      const double cells = (double)a.rows * a.H * a.W;
      pw[i].wpw = cell_of_pack(e, probs[i].wpack)->wpw3.p;
      if (!a.zero_state) {
        tr3.push_back(mv::Wn3TransformItem{q.h16, q.h_plane_stride, e->pv3h[i].p, a.src_row_h,
                                           a.rows, a.H, a.W, a.C});
        pw[i].v3h = e->pv3h[i].p;
        tr3_bytes += cells * a.C * 4.0 * (1.0 + 5.0 / 3.0);
      }
      if (!a.x_small && a.Cx > 0 && !a.sx_corr) {
        tr3.push_back(mv::Wn3TransformItem{q.x16, q.x_plane_stride, e->pv3x[i].p, nullptr,
                                           a.rows, a.H, a.W, a.Cx});
        pw[i].v3x = e->pv3x[i].p;
        tr3_bytes += cells * a.Cx * 4.0 * (1.0 + 5.0 / 3.0);
      }
    }
  }
  if (!tr3.empty())
    launch(e, "wino3_transform", 0, tr3_bytes, [&] {
      mv::launch_wino3_transforms(tr3.data(), (int)tr3.size(), e->stream);
    });
  // fp16 MFMA products ISSUED per executed fp32 product: 3 in the direct form; in a Winograd
  // form 3 * (components * row tiles) / (3 * H) -- partial tiles count (9 rows = 5 pairs: 2.22,
  // not 2), weighted over the group by executed FLOPs
  double factor = e->compute_mode == 2 ? 1.0 : 3.0;
  if (wino || wino3) {
    double num = 0, den = 0;
    for (const auto& a : probs) {
      const double cx = a.sx_corr ? 0.0 : (double)a.Cx;
      const double fl = (double)a.rows * a.H * a.W * (cx + (a.zero_state ? 0 : a.C));
      // (the halo tiling issues 32 lanes for 30 owned triple-cells)
      const double per = wino3 ? 5.0 * ((a.H + 2) / 3) / a.H * (mv::wino3_needs_halo(a) ? 32.0 / 30.0 : 1.0)
                               : 4.0 * ((a.H + 1) / 2) / a.H;
      num += fl * per; den += fl;
    }
    factor = den > 0 ? num / den : (wino3 ? 5.0 / 3.0 : 2.0);
  }
  launch(e, "convlstm_step", flops, bytes, [&] {
    if (e->compute_mode == 2)
      mv::launch_convlstm_bf16_steps(p16.data(), (int)p16.size(), e->stream);
    else if (wino3)
      mv::launch_convlstm_wino3_steps(pw.data(), (int)pw.size(), e->stream);
    else if (wino)
      mv::launch_convlstm_wino_steps(pw.data(), (int)pw.size(), e->stream);
    else
      mv::launch_convlstm16_steps(p16.data(), (int)p16.size(), e->stream);
  }, dense, factor);
}

// One launch for up to four independent ConvLSTM steps (class / regression
// chain of each scale advance in lockstep).
void run_conv_group(mv_engine* e, const std::vector<ConvLstmArgs>& probs) {
  if (probs.empty()) return;
  double flops = 0, bytes = 0, dense = 0;
  for (const auto& a : probs) {
    const double M = (double)a.rows * a.H * a.W;
    // dense: the step as the reference computes it; executed: a zero-state step
    // (first encoder step) never multiplies the h half and never reads h, c
    dense += 2.0 * M * 9.0 * (a.Cx + a.C) * 4.0 * a.C;
    // sparse x: the x k-steps are not executed (table terms in the epilogue)
    const double cx = a.sx_corr ? 0.0 : (double)a.Cx;
    flops += 2.0 * M * 9.0 * (cx + (a.zero_state ? 0 : a.C)) * 4.0 * a.C;
    bytes += M * (cx + (a.zero_state ? 2.0 : 4.0) * a.C) * 4.0;   // x,(h,c) in; h,c out
  }
  if (e->compute_mode != 0) {
    run_conv_group_f16x3(e, probs, flops, bytes, dense);
    return;
  }
  if (e->cfg.convlstm_kernel != 3) {            // --convlstm_kernel 1 / 5 / ...: plain fp32 loops
    const double kk = (double)e->cfg.convlstm_kernel * e->cfg.convlstm_kernel / 9.0;
    launch(e, "convlstm_step", flops * kk, bytes, [&] {
      for (const auto& a : probs) {
        mv::ConvGenericArgs ga{};
        ga.f = a;
        ga.w = cell_of_bias(e, a.bias)->kernel->dev.p;
        ga.ksize = e->cfg.convlstm_kernel;
        mv::launch_convlstm_generic_step(ga, e->stream);
      }
    }, dense * kk, 0.0);
    return;
  }
  launch(e, "convlstm_step", flops, bytes, [&] {
    mv::launch_convlstm_steps(probs.data(), (int)probs.size(), e->stream);
  }, dense, 1.0);
}

void run_scene(mv_engine* e) {
  const mv_config& c = e->cfg;
  const int U = e->num_frames;
  const float* in = e->scene_feat.p;
  int Hi = c.scene_h, Wi = c.scene_w, Ci = c.scene_class;
  const int k = c.scene_conv_kernel;
  for (int i = 0; i < c.num_scales; ++i) {
    const int Ho = e->conv_h[i], Wo = e->conv_w[i], Co = c.scene_conv_dim;
    const int pad_h = std::max((Ho - 1) * 2 + k - Hi, 0);
    const int pad_w = std::max((Wo - 1) * 2 + k - Wi, 0);
    const size_t total = (size_t)U * Ho * Wo * Co;
    float* out = e->scene_conv[i].p;
    const float *w = e->scene_W[i]->dev.p, *b = e->scene_b[i]->dev.p;
    if (k == 1 && Co <= 64) {     // --scene_conv_kernel 1: the dense 1x1 projection, on MFMA
      const size_t M = (size_t)U * Ho * Wo;
      launch(e, "scene_proj1x1_mfma", 2.0 * M * Ci * Co,
             4.0 * (total + (double)M * Ci), [&] {
        hipLaunchKernelGGL(mv::scene_proj1x1_mfma_kernel, dim3(cdiv(M, 128)), dim3(256), 0,
                           e->stream, in, w, b, out, U, Hi, Wi, Ci, Ho, Wo, Co, c.activation);
      });
    } else {
    launch(e, "scene_conv_s2_tanh", 2.0 * total * k * k * Ci,
           4.0 * (total + (double)U * Hi * Wi * Ci), [&] {
      hipLaunchKernelGGL(mv::scene_conv_s2_tanh_kernel, dim3(cdiv(total, 256)),
                         dim3(256), 0, e->stream, in, w, b, out, U, Hi, Wi, Ci,
                         Ho, Wo, Co, k, pad_h / 2, pad_w / 2, c.activation);
    });
    }
    in = out; Hi = Ho; Wi = Wo; Ci = Co;
  }
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    const size_t total = (size_t)c.batch_size * S.K * c.scene_conv_dim;
    launch(e, "scene_mean", (double)total * c.obs_len,
           4.0 * total * (c.obs_len + 1), [&] {
      hipLaunchKernelGGL(mv::scene_mean_kernel, dim3(cdiv(total, 256)), dim3(256),
                         0, e->stream, e->scene_conv[s].p, e->obs_scene.p,
                         S.scene_mean.p, c.batch_size, c.obs_len, S.K,
                         c.scene_conv_dim);
    });
  }
}

struct Cursors {                 // which ping-pong buffer holds the live state
  int cls[MV_MAX_SCALES] = {0, 0};
  int reg[MV_MAX_SCALES] = {0, 0};
};

// MV_BEAM_SHARED_FIRST=0 restores the tiled first beam step for A/B runs (run_decoders_beam)
static bool beam_shared_first() {
  static const bool on =
      !(getenv("MV_BEAM_SHARED_FIRST") && atoi(getenv("MV_BEAM_SHARED_FIRST")) == 0);
  return on;
}

// Encoders of every enabled scale (dynamic_rnn from the zero state, T_o steps;
// code/pred_models.py:212-215, 232-234), all chains advanced in lockstep.
void run_encoders(mv_engine* e, Cursors& cur) {
  const mv_config& c = e->cfg;
  const int N = c.batch_size, T = c.obs_len, D = c.scene_conv_dim;
  for (int t = 0; t < T; ++t) {
    std::vector<ConvLstmArgs> probs;
    for (int s = 0; s < c.num_scales; ++s) {
      ScaleState& S = e->sc[s];
      if (!S.use) continue;
      const size_t total = (size_t)N * S.K * D;
      const bool sparse = sparse_x_on(e, S);
      const size_t nc = (size_t)N * 9 * 4 * c.hidden_size;   // table of one step
      if (sparse) {
        if (t == 0)      // the tables of all T_o steps in one launch
          launch(e, "sx_encoder_corr", 2.0 * nc * D * T, 4.0 * nc * T, [&] {
            hipLaunchKernelGGL(mv::sx_encoder_corr_kernel,
                               dim3(cdiv((size_t)4 * c.hidden_size, 256), 9,
                                    cdiv((size_t)N, mv::kSxRows) * T),
                               dim3(256), 0, e->stream, S.enc_cls.kernel->dev.p,
                               e->scene_conv[s].p, e->obs_scene.p, S.labels.p, N, T, -1, S.K, D,
                               c.hidden_size, S.sx_enc_corr.p);
          });
      } else {
      launch(e, "enc_class_input", 0, 4.0 * total, [&] {
        size_t pst = 0;
        _Float16* p16 = e->plane_out(S.xbuf_cls.p, &pst);
        hipLaunchKernelGGL(mv::enc_class_input_kernel, dim3(cdiv(total, 256)),
                           dim3(256), 0, e->stream, e->scene_conv[s].p,
                           e->obs_scene.p, S.labels.p, S.xbuf_cls.p, N, T, t, S.K, D, p16,
                           pst);
      });
      }
      // x = grid_obs_regress[:, t] is read in place through the row stride
      const size_t row = (size_t)S.K * 2;
      const int cc = cur.cls[s], cr = cur.reg[s];
      probs.push_back(conv_problem(e, S.enc_cls, S.xbuf_cls.p, S.cls_h[cc].p,
                                   S.cls_c[cc].p, nullptr, nullptr, S.cls_h[cc ^ 1].p,
                                   S.cls_c[cc ^ 1].p, N, S.H, S.W, t == 0, 0,
                                   /*want_h16=*/t + 1 < T || !c.use_gnn));
      // the class encoder's h' is read as fp32 only by the graph attention in front of the
      // first decoder step (and by the tiled-first-step A/B path of the beam decoder); the
      // regression encoder's never
      probs.back().skip_h32 =
          (t + 1 < T || (!c.use_gnn && (c.beam_size == 1 || beam_shared_first()))) ? 1 : 0;
      if (sparse) {
        set_sparse_x(e, S, probs.back(), false, S.labels.p + t, T, 1);
        probs.back().sx_corr = S.sx_enc_corr.p + (size_t)t * nc;
      }
      if (!c.use_single_decoder)     // single decoder: the regression encoder feeds nothing
        probs.push_back(conv_problem(e, S.enc_reg, S.obs_reg.p + (size_t)t * row,
                                     S.reg_h[cr].p, S.reg_c[cr].p, nullptr, nullptr,
                                     S.reg_h[cr ^ 1].p, S.reg_c[cr ^ 1].p, N, S.H, S.W,
                                     t == 0, (size_t)T * row));
      if (!c.use_single_decoder) probs.back().skip_h32 = 1;
      cur.cls[s] ^= 1; cur.reg[s] ^= 1;
    }
    run_conv_group(e, probs);
  }
}

// One attention pass per job; the LDS-tiled kernel takes up to two jobs per launch (the
// two grid scales of a greedy step are 62 + 22 us back to back, one round of workgroups
// each: together they fill the chip better).
// MV_GNN = v1 (one wave per cell) / v2 (LDS-tiled, one cell per thread) select the earlier
// kernels for A/B runs; default: the register-blocked third version.
static int gnn_version() {
  const char* v = getenv("MV_GNN");
  if (v && strcmp(v, "v1") == 0) return 1;
  if (v && strcmp(v, "v2") == 0) return 2;
  return 3;
}

struct GnnJob {
  ScaleState* S; const float* h; const int32_t* src_row; float* out; int rows, sm_div;
  const int32_t* row_ref = nullptr;
};

void run_gnn_jobs(mv_engine* e, const std::vector<GnnJob>& jobs) {
  const mv_config& c = e->cfg;
  static const int env_ver = gnn_version();
  // The third version addresses h, the scene means and nothing else through 32-bit byte
  // offsets and takes the scene channels in one 64-channel chunk; anything else runs on
  // the second.
  const size_t max_rows = (size_t)c.batch_size * (size_t)std::max(1, c.beam_size);
  auto v3_ok = [&](const GnnJob& J) {
    return (gnn_scene_dim(e) == 0 || gnn_scene_dim(e) == 64) &&
           max_rows * J.S->K * c.hidden_size * 4 < ((size_t)1 << 32);
  };
  for (size_t j0 = 0; j0 < jobs.size();) {
    const GnnJob& A = jobs[j0];
    const bool tiled = env_ver >= 2 && A.S->W <= 32 && c.hidden_size == 256 && c.scene_conv_dim <= 64;
    size_t nj = 1;
    if (tiled && j0 + 1 < jobs.size() && jobs[j0 + 1].S->W <= 32) nj = 2;
    const int ver = env_ver >= 3 && !(v3_ok(A) && (nj == 1 || v3_ok(jobs[j0 + 1]))) ? 2 : env_ver;
    mv::GnnGroup grp{};
    double flops = 0, bytes = 0;
    unsigned nblocks = 0;
    for (size_t j = 0; j < nj; ++j) {
      const GnnJob& J = jobs[j0 + j];
      const size_t cells = (size_t)J.rows * J.S->K;
      size_t pst = 0;
      _Float16* p16 = e->plane_out(J.out, &pst);
      // f16x3 inference: the only consumer of h + GNN(h) is the gate convolution, which
      // reads the operand planes -- the fp32 copy is not written at all
      const bool need_f32 = !(tiled && p16 && !e->train && e->compute_mode != 0);
      flops += cells * (9.0 * 2 * 2 * (c.hidden_size + gnn_scene_dim(e)) +
                        9.0 * 2 * c.hidden_size);
      bytes += 4.0 * cells * c.hidden_size *
                   (1.0 + (need_f32 ? 1.0 : 0.0) + (p16 ? 1.0 : 0.0)) +
               4.0 * (cells / J.sm_div) * gnn_scene_dim(e);
      int ng = 0;
      const unsigned nb = ver >= 3 ? mv::gnn_v3_blocks(cells, &ng) : mv::gnn_v2_blocks(cells, &ng);
      mv::GnnProblem& P = grp.p[j];
      P.h = J.h; P.scene_mean = J.S->scene_mean.p; P.src_row = J.src_row;
      P.out = need_f32 ? J.out : nullptr; P.p16 = p16; P.p16_stride = pst;
      P.M = J.rows; P.H = J.S->H; P.W = J.S->W; P.sm_div = J.sm_div; P.ngroups = ng;
      P.row_ref = J.row_ref;
      if (j == 0) grp.nblocks0 = nb;
      nblocks += nb;
    }
    if (nj == 1) grp.nblocks0 = nblocks;
    launch(e, "gnn_attend", flops, bytes, [&] {
      if (tiled && ver >= 3) {
        hipLaunchKernelGGL(mv::gnn_attend_v3_kernel, dim3(nblocks), dim3(mv::kGnn3Threads), 0,
                           e->stream, grp, c.hidden_size, gnn_scene_dim(e));
      } else if (tiled) {
        hipLaunchKernelGGL(mv::gnn_attend_v2_kernel, dim3(nblocks), dim3(mv::kGnnThreads), 0, e->stream,
                           grp, c.hidden_size, gnn_scene_dim(e));
      } else {
        const size_t cells = (size_t)A.rows * A.S->K;
        size_t pst = 0;
        _Float16* p16 = e->plane_out(A.out, &pst);
        if (c.hidden_size <= 256)
          hipLaunchKernelGGL(mv::gnn_attend_kernel<1>, dim3(cdiv(cells, 4)), dim3(256), 0,
                             e->stream, A.h, A.S->scene_mean.p, A.src_row, A.out, A.rows,
                             A.S->H, A.S->W, c.hidden_size, gnn_scene_dim(e), A.sm_div, p16, pst);
        else
          hipLaunchKernelGGL(mv::gnn_attend_kernel<2>, dim3(cdiv(cells, 4)), dim3(256), 0,
                             e->stream, A.h, A.S->scene_mean.p, A.src_row, A.out, A.rows,
                             A.S->H, A.S->W, c.hidden_size, gnn_scene_dim(e), A.sm_div, p16, pst);
      }
    });
    j0 += nj;
  }
}

void run_gnn(mv_engine* e, ScaleState& S, const float* h, const int32_t* src_row,
             float* out, int rows, int sm_div) {
  run_gnn_jobs(e, {GnnJob{&S, h, src_row, out, rows, sm_div}});
}

template <int P>
void run_hidden2grid(mv_engine* e, ScaleState& S, const float* h, const float* w,
                     float* out, size_t out_row_stride, int rows) {
  const size_t cells = (size_t)rows * S.K;
  const int C = e->cfg.hidden_size;
  launch(e, "hidden2grid", cells * 2.0 * 9 * C * P, 4.0 * cells * (C + P), [&] {
    hipLaunchKernelGGL(mv::hidden2grid_kernel<P>, dim3(cdiv(cells, 4)), dim3(256),
                       0, e->stream, h, w, out, out_row_stride, rows, S.H, S.W, C);
  });
}

// MV_TAIL=v1 selects the first-round decoder tail (hidden2grid convolved in place,
// separate argmax / embedding launches) for A/B runs
static bool tail_v2() {
  static const bool on = !(getenv("MV_TAIL") && strcmp(getenv("MV_TAIL"), "v1") == 0);
  return on;
}

void run_emb_onehot(mv_engine* e, ScaleState& S, const int32_t* ids, int stride,
                    float* out, int rows, int ids_div = 1) {
  const int E = e->cfg.emb_size;
  const size_t total = (size_t)rows * S.K * E;
  launch(e, "grid_emb_onehot", (double)total, 4.0 * total, [&] {
    size_t pst = 0;
    _Float16* p16 = e->plane_out(out, &pst);
    if (tail_v2() && E % 8 == 0)
      hipLaunchKernelGGL(mv::grid_emb_onehot8_kernel, dim3(cdiv(total / 8, 256)), dim3(256), 0,
                         e->stream, ids, stride, ids_div, S.emb_cls_W->dev.p,
                         S.emb_cls_b->dev.p, out, rows, S.H, S.W, E, p16, pst, e->cfg.activation);
    else
      hipLaunchKernelGGL(mv::grid_emb_onehot_kernel, dim3(cdiv(total, 256)),
                         dim3(256), 0, e->stream, ids, stride, ids_div, S.emb_cls_W->dev.p,
                         S.emb_cls_b->dev.p, out, rows, S.H, S.W, E, p16, pst, e->cfg.activation);
  });
}

// grid_emb on a dense P-channel map: the regression decoder's (dx, dy) maps (default
// weights), or -- class decoder fed its own logits / the ground-truth map (training
// without --train_w_onehot, teacher forcing) -- a 1-channel map with the class weights
void run_emb_dense(mv_engine* e, ScaleState& S, const float* x, size_t row_stride,
                   float* out, int rows, Param* W = nullptr, Param* b = nullptr, int P = 2) {
  const int E = e->cfg.emb_size;
  const size_t total = (size_t)rows * S.K * E;
  if (!W) { W = S.emb_reg_W; b = S.emb_reg_b; }
  launch(e, "grid_emb_dense", total * 2.0 * 9 * P, 4.0 * total, [&] {
    size_t pst = 0;
    _Float16* p16 = e->plane_out(out, &pst);
    hipLaunchKernelGGL(mv::grid_emb_dense_kernel, dim3(cdiv(total, 256)), dim3(256),
                       0, e->stream, x, row_stride, W->dev.p, b->dev.p, out, rows, S.H, S.W,
                       P, E, p16, pst, e->cfg.activation);
  });
}

// Regression decoder step t, always greedy and un-beamed
// (code/pred_models.py:298-305 -> grid_decoder :311-471 with input_onehot=False,
// use_gnn=False): input embedding + the conv problem; the caller launches it.
ConvLstmArgs reg_decoder_problem(mv_engine* e, int s, Cursors& cur, int t, int Tp,
                                 bool embed = true) {
  const mv_config& c = e->cfg;
  ScaleState& S = e->sc[s];
  const int N = c.batch_size, T = c.obs_len;
  const size_t orow = (size_t)Tp * S.K * 2;
  if (t == 0)  // first_input = obs_grid_reg[:, -1]
    run_emb_dense(e, S, S.obs_reg.p + (size_t)(T - 1) * S.K * 2, (size_t)T * S.K * 2,
                  S.xbuf_reg.p, N);
  else if (embed)   // hidden2grid output of the previous step (else: the tail embedded it)
    run_emb_dense(e, S, S.out_reg.p + (size_t)(t - 1) * S.K * 2, orow, S.xbuf_reg.p, N);
  const int cr = cur.reg[s];
  cur.reg[s] ^= 1;
  return conv_problem(e, S.dec_reg, S.xbuf_reg.p, S.reg_h[cr].p, S.reg_c[cr].p, nullptr,
                      nullptr, S.reg_h[cr ^ 1].p, S.reg_c[cr ^ 1].p, N, S.H, S.W, false);
}

void reg_decoder_output(mv_engine* e, int s, const Cursors& cur, int t, int Tp) {
  ScaleState& S = e->sc[s];
  const size_t orow = (size_t)Tp * S.K * 2;
  run_hidden2grid<2>(e, S, S.reg_h[cur.reg[s]].p, S.out_reg_W->dev.p,
                     S.out_reg.p + (size_t)t * S.K * 2, orow, e->cfg.batch_size);
}

// The decoder tail of step t for all chains (decode_tail.h): hidden2grid as one
// grouped GEMM launch reading every h' once, then one workgroup per (chain, row) for
// the 9-tap gather, the output row, the greedy argmax and the NEXT step's embedding.
// cls_rows / cls_out / cls_stride describe the class chain's logits destination
// (greedy: out_cls step t; beam: bm_logits of this time step, no argmax / embedding).
struct TailPlan {
  int s;
  const float* cls_h; int cls_rows; float* cls_out; int64_t cls_stride;
  bool cls_next;       // class chain: argmax + embedding of step t+1 (greedy only)
  const float* reg_h; float* reg_out; int64_t reg_stride; bool reg_next;
  int reg_rows = 0;    // 0: N (the un-beamed regression chain)
  // training forward: where the argmax ids and the next step's embeddings go (null: the
  // inference buffers S.ids / S.xbuf_cls / S.xbuf_reg with their operand planes)
  int32_t* cls_ids_out = nullptr; float* cls_x_out = nullptr; float* reg_x_out = nullptr;
  bool cls_embed = true;   // false: ids only (the embedding is made elsewhere)
};

void run_tail(mv_engine* e, const std::vector<TailPlan>& plans) {
  const mv_config& c = e->cfg;
  const int C = c.hidden_size, E = c.emb_size, N = c.batch_size;
  std::vector<mv::H2gQProblem> qp;
  std::vector<mv::TailProblem> tp;
  double qbytes = 0, qflops = 0, tbytes = 0;
  for (const TailPlan& pl : plans) {
    ScaleState& S = e->sc[pl.s];
    MV_REQUIRE((size_t)S.K * 2 <= 2048 && E % 16 == 0 && E <= 512,
               "decode tail: K %d / emb_size %d", S.K, E);
    const int reg_rows = pl.reg_rows ? pl.reg_rows : N;
    const size_t cc = (size_t)pl.cls_rows * S.K, cr = (size_t)reg_rows * S.K;
    qp.push_back(mv::H2gQProblem{pl.cls_h, S.wq_cls.p, S.q_cls.p, (int32_t)cc, 1});
    qp.push_back(mv::H2gQProblem{pl.reg_h, S.wq_reg.p, S.q_reg.p, (int32_t)cr, 2});
    qbytes += 4.0 * (cc * (C + 9.0) + cr * (C + 18.0));
    qflops += 2.0 * 9 * C * (cc + 2.0 * cr);
    mv::TailProblem a{};
    a.q = S.q_cls.p; a.out = pl.cls_out; a.out_row_stride = pl.cls_stride;
    a.rows = pl.cls_rows; a.H = S.H; a.W = S.W; a.P = 1; a.E = E; a.onehot = 1;
    a.act = c.activation;
    tbytes += 4.0 * cc * (9 + 1);
    if (pl.cls_next) {
      a.ids_out = pl.cls_ids_out ? pl.cls_ids_out : S.ids.p;
      if (pl.cls_x_out) {           // training: fp32 embedding into the time-major x buffer
        if (pl.cls_embed) {
          a.emb_w = S.emb_cls_W->dev.p; a.emb_b = S.emb_cls_b->dev.p;
          a.x_out = pl.cls_x_out;
          tbytes += 4.0 * cc * E;
        }
      } else if (!sparse_x_on(e, S)) {     // sparse x: the next step needs the id, not the embedding
        size_t pst = 0;
        a.emb_w = S.emb_cls_W->dev.p; a.emb_b = S.emb_cls_b->dev.p;
        a.x_out = S.xbuf_cls.p;
        a.x16 = e->plane_out(S.xbuf_cls.p, &pst); a.x16_stride = (int64_t)pst;
        tbytes += 4.0 * cc * E * (a.x16 ? 2 : 1);
      }
    }
    tp.push_back(a);
    mv::TailProblem b{};
    b.q = S.q_reg.p; b.out = pl.reg_out; b.out_row_stride = pl.reg_stride;
    b.rows = reg_rows; b.H = S.H; b.W = S.W; b.P = 2; b.E = E; b.onehot = 0;
    b.act = c.activation;
    tbytes += 4.0 * cr * (18 + 2);
    if (pl.reg_next) {
      size_t pst = 0;
      b.emb_w = S.emb_reg_W->dev.p; b.emb_b = S.emb_reg_b->dev.p;
      if (pl.reg_x_out) {
        b.x_out = pl.reg_x_out;
        tbytes += 4.0 * cr * E;
      } else {
      b.x_out = S.xbuf_reg.p;
      b.x16 = e->plane_out(S.xbuf_reg.p, &pst); b.x16_stride = (int64_t)pst;
      tbytes += 4.0 * cr * E * (b.x16 ? 2 : 1);
      }
    }
    tp.push_back(b);
  }
  MV_REQUIRE(qp.size() <= (size_t)mv::kTailMax, "decode tail: too many chains");
  launch(e, "hidden2grid", qflops, qbytes, [&] {
    mv::launch_h2g_q(qp.data(), (int)qp.size(), C, e->stream);
  });
  launch(e, "decode_tail", 0, tbytes, [&] {
    mv::launch_decode_tail(tp.data(), (int)tp.size(), e->stream);
  });
}

// Greedy decoders of every enabled scale in lockstep: class decoder
// (grid_decoder with input_onehot, use_gnn; code/pred_models.py:311-471) and
// regression decoder.
void run_decoders_greedy(mv_engine* e, Cursors& cur, int Tp) {
  const mv_config& c = e->cfg;
  const int N = c.batch_size, T = c.obs_len;
  const bool v2 = tail_v2();
  for (int t = 0; t < Tp; ++t) {
    std::vector<ConvLstmArgs> probs;
    if (c.use_gnn) {
      std::vector<GnnJob> jobs;
      for (int s = 0; s < c.num_scales; ++s)
        if (e->sc[s].use)
          jobs.push_back(GnnJob{&e->sc[s], e->sc[s].cls_h[cur.cls[s]].p, nullptr,
                                e->sc[s].cls_hg.p, N, 1});
      run_gnn_jobs(e, jobs);
    }
    for (int s = 0; s < c.num_scales; ++s) {
      ScaleState& S = e->sc[s];
      if (!S.use) continue;
      const int cc = cur.cls[s];
      const float* hin = c.use_gnn ? S.cls_hg.p : S.cls_h[cc].p;
      // sparse x: the embedding of a one-hot map enters the gate kernel as table terms
      const bool sparse = sparse_x_on(e, S) && (t == 0 || !c.class_feedback_dense);
      if (sparse)
        ;
      else if (t == 0)  // one_hot(last observed cell)
        run_emb_onehot(e, S, S.labels.p + (T - 1), T, S.xbuf_cls.p, N);
      else if (c.class_feedback_dense)   // raw logits of the previous step (:388-406)
        run_emb_dense(e, S, S.out_cls.p + (size_t)(t - 1) * S.K, (size_t)Tp * S.K,
                      S.xbuf_cls.p, N, S.emb_cls_W, S.emb_cls_b, 1);
      else if (!v2)
        run_emb_onehot(e, S, S.ids.p, 1, S.xbuf_cls.p, N);
      probs.push_back(conv_problem(e, S.dec_cls, S.xbuf_cls.p, hin, S.cls_c[cc].p,
                                   nullptr, nullptr, S.cls_h[cc ^ 1].p,
                                   S.cls_c[cc ^ 1].p, N, S.H, S.W, false, 0,
                                   /*want_h16=*/!c.use_gnn));
      if (sparse) {
        if (t == 0) set_sparse_x(e, S, probs.back(), true, S.labels.p + (T - 1), T, 1);
        else set_sparse_x(e, S, probs.back(), true, S.ids.p, 1, 1);
      }
      cur.cls[s] ^= 1;
      if (!c.use_single_decoder) probs.push_back(reg_decoder_problem(e, s, cur, t, Tp, !v2));
    }
    // longest tiles first: the dense-x problems (162 k-steps per tile) are dispatched
    // before the sparse-x ones (144), so the last, partly filled round of workgroups is
    // made of the short ones
    std::stable_sort(probs.begin(), probs.end(), [](const ConvLstmArgs& a, const ConvLstmArgs& b) {
      return (a.sx_corr == nullptr) > (b.sx_corr == nullptr);
    });
    run_conv_group(e, probs);
    if (v2) {
      std::vector<TailPlan> plans;
      for (int s = 0; s < c.num_scales; ++s) {
        ScaleState& S = e->sc[s];
        if (!S.use) continue;
        TailPlan pl{};
        pl.s = s;
        pl.cls_h = S.cls_h[cur.cls[s]].p; pl.cls_rows = N;
        pl.cls_out = S.out_cls.p + (size_t)t * S.K; pl.cls_stride = (int64_t)Tp * S.K;
        pl.cls_next = t + 1 < Tp && !c.class_feedback_dense;
        pl.reg_h = S.reg_h[cur.reg[s]].p;
        pl.reg_out = S.out_reg.p + (size_t)t * S.K * 2; pl.reg_stride = (int64_t)Tp * S.K * 2;
        pl.reg_next = t + 1 < Tp;
        if (c.use_single_decoder) {    // offsets from the class decoder's state (:287-296)
          pl.reg_h = pl.cls_h;
          pl.reg_next = false;
        }
        plans.push_back(pl);
      }
      run_tail(e, plans);
      continue;
    }
    for (int s = 0; s < c.num_scales; ++s) {
      ScaleState& S = e->sc[s];
      if (!S.use) continue;
      const size_t orow = (size_t)Tp * S.K;
      float* logits = S.out_cls.p + (size_t)t * S.K;
      run_hidden2grid<1>(e, S, S.cls_h[cur.cls[s]].p, S.out_cls_W->dev.p, logits, orow, N);
      if (t + 1 < Tp && !c.class_feedback_dense) {
        launch(e, "argmax_rows", 0, 4.0 * N * S.K, [&] {
          hipLaunchKernelGGL(mv::argmax_rows_kernel, dim3(N), dim3(64), 0, e->stream,
                             logits, orow, S.ids.p, N, S.K);
        });
      }
      if (c.use_single_decoder)
        run_hidden2grid<2>(e, S, S.cls_h[cur.cls[s]].p, S.out_reg_W->dev.p,
                           S.out_reg.p + (size_t)t * S.K * 2, (size_t)Tp * S.K * 2, N);
      else
        reg_decoder_output(e, s, cur, t, Tp);
    }
  }
}

__global__ void tile_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                 size_t row_elems4, int B, size_t total4) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const size_t r = idx / row_elems4, off = idx - r * row_elems4;
  reinterpret_cast<mv::f32x4_t*>(out)[idx] =
      reinterpret_cast<const mv::f32x4_t*>(in)[(r / B) * row_elems4 + off];
}

// logits[(n*B + b), :] = logits[(n*B), :] for b > 0 (the shared first beam step)
__global__ void tile_beam0_kernel(float* __restrict__ logits, int K, int B, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const size_t r = idx / K;
  if (r % B) logits[idx] = logits[(r - r % B) * K + (idx - r * K)];
}

__global__ void beam_backtrace_kernel(const int32_t* __restrict__ step_ids,
                                      const int32_t* __restrict__ step_parents,
                                      int32_t* __restrict__ out_ids,
                                      int32_t* __restrict__ trace, int N, int B,
                                      int T) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * B) return;
  const int n = idx / B, b = idx - n * B;
  int par = b;  // parents_0 = arange(B), code/pred_models.py:714-716
  for (int t = T - 1; t >= 0; --t) {
    const size_t o = ((size_t)t * N + n) * B + par;
    out_ids[((size_t)n * B + b) * T + t] = step_ids[o];
    trace[((size_t)n * B + b) * T + t] = par;
    par = step_parents[o];
  }
}

__global__ void beam_gather_logits_kernel(const float* __restrict__ step_logits,
                                          const int32_t* __restrict__ trace,
                                          float* __restrict__ out, int N, int B,
                                          int T, int K) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * B * T * K;
  if (idx >= total) return;
  const int k = idx % K;
  size_t r = idx / K;
  const int t = r % T; r /= T;
  const int b = r % B;
  const int n = r / B;
  const int par = trace[((size_t)n * B + b) * T + t];
  out[idx] = step_logits[(((size_t)t * N + n) * B + par) * K + k];
}

// Beam-search class decoder (grid_decoder_beam_search,
// code/pred_models.py:474-806) with the un-beamed regression decoder advanced
// in lockstep (its step t shares a launch with beam time t+1).
void run_decoders_beam(mv_engine* e, int s, Cursors& cur, int Tp) {
  const mv_config& c = e->cfg;
  ScaleState& S = e->sc[s];
  const int N = c.batch_size, T = c.obs_len, B = c.beam_size, K = S.K,
            C = c.hidden_size;
  const int R = N * B;
  // The reference tiles the encoder state and the first input over the beams (:497-502,
  // 531-532), so the first cell step (and the attention before it) sees B identical rows
  // per sample.  Rows are independent in every kernel of the step, so that step runs ONCE
  // per sample on the N encoder rows (bit-identical to the tiled computation): its logits
  // are copied to the B beam rows, and the first selection hands out state rows n instead
  // of n * B + parent.  MV_BEAM_SHARED_FIRST=0 restores the tiled first step for A/B runs.
  const bool shared_first = beam_shared_first();
  if (!shared_first) {
    const int cc = cur.cls[s];
    const size_t row4 = (size_t)K * C / 4, total4 = (size_t)R * row4;
    launch(e, "beam_tile_state", 0, 8.0 * total4 * 16, [&] {
      hipLaunchKernelGGL(tile_rows_kernel, dim3(cdiv(total4, 256)), dim3(256), 0,
                         e->stream, S.cls_h[cc].p, S.cls_h[cc ^ 1].p, row4, B, total4);
      hipLaunchKernelGGL(tile_rows_kernel, dim3(cdiv(total4, 256)), dim3(256), 0,
                         e->stream, S.cls_c[cc].p, S.cls_c[cc ^ 1].p, row4, B, total4);
    });
    e->plane_invalidate(S.cls_h[cc ^ 1].p);   // fp32 copy only: planes are re-split
    cur.cls[s] ^= 1;
  }
  HIP_CHECK(hipMemsetAsync(e->bm_lp[0].p, 0, (size_t)R * sizeof(float), e->stream));
  int lpi = 0;
  const int32_t* src = nullptr;  // state row indirection for the next cell step
  const bool sparse = sparse_x_on(e, S);
  // Graph attention BEFORE the parent gather: h + GNN(h) depends on the state row alone, and
  // the B beams of a sample descend from few distinct parents, so it is computed once per
  // state row that some surviving beam continues (beam_select marks them in bm_ref; the rest
  // are skipped) and the next cell step reads it through the parent indirection, like c.
  // Bit-identical to attention after the gather.  MV_BEAM_GNN_DEDUPE=0 gathers first.
  static const bool dedupe_env =
      !(getenv("MV_BEAM_GNN_DEDUPE") && atoi(getenv("MV_BEAM_GNN_DEDUPE")) == 0);
  const bool dedupe = dedupe_env && shared_first && c.use_gnn && K <= 64 * mv::kBeamRankJ &&
                      !(getenv("MV_BEAM_STEP") && strcmp(getenv("MV_BEAM_STEP"), "v1") == 0) &&
                      !(getenv("MV_GNN") && strcmp(getenv("MV_GNN"), "v1") == 0);
  for (int time = 0; time <= Tp; ++time) {
    // rows the state holds going INTO this iteration's kernels
    const bool one_per_sample = shared_first && time <= 1;
    const int rows_now = one_per_sample ? N : R;
    if (time > 0) {
      // cell step; h comes from the GNN buffer (identity rows) when use_gnn, c through
      // the parent indirection
      const int cc = cur.cls[s];
      const float* hin = c.use_gnn ? S.cls_hg.p : S.cls_h[cc].p;
      std::vector<ConvLstmArgs> probs;
      probs.push_back(conv_problem(e, S.dec_cls, S.xbuf_cls.p, hin, S.cls_c[cc].p,
                                   (c.use_gnn && !dedupe) ? nullptr : src, src,
                                   S.cls_h[cc ^ 1].p, S.cls_c[cc ^ 1].p, rows_now, S.H, S.W,
                                   false, 0, /*want_h16=*/!c.use_gnn));
      if (sparse) {
        if (time == 1)
          set_sparse_x(e, S, probs.back(), true, S.labels.p + (T - 1), T, one_per_sample ? 1 : B);
        else
          set_sparse_x(e, S, probs.back(), true, e->bm_ids.p + (size_t)(time - 2) * R, 1, 1);
      }
      cur.cls[s] ^= 1;
      const bool v2 = tail_v2();
      const bool single = c.use_single_decoder != 0;
      MV_REQUIRE(!single || v2, "use_single_decoder with beam search needs the v2 decoder tail");
      if (!single) probs.push_back(reg_decoder_problem(e, s, cur, time - 1, Tp, !v2));
      run_conv_group(e, probs);
      float* logits = e->bm_logits.p + (size_t)(time - 1) * R * K;
      // single decoder: the offsets of this step, decoded from every state row (traced back
      // along the beams after the loop)
      float* regstep = single ? e->bm_reg_steps.p + (size_t)(time - 1) * R * K * 2 : nullptr;
      // one row per sample: the logits land in beam 0's row of each sample
      const size_t lrow = one_per_sample ? (size_t)B * K : (size_t)K;
      if (v2) {
        TailPlan pl{};
        pl.s = s;
        pl.cls_h = S.cls_h[cur.cls[s]].p; pl.cls_rows = rows_now;
        pl.cls_out = logits; pl.cls_stride = lrow; pl.cls_next = false;   // beam_step selects
        if (single) {
          pl.reg_h = pl.cls_h; pl.reg_rows = rows_now;
          pl.reg_out = regstep; pl.reg_stride = (int64_t)lrow * 2; pl.reg_next = false;
        } else {
        pl.reg_h = S.reg_h[cur.reg[s]].p;
        pl.reg_out = S.out_reg.p + (size_t)(time - 1) * K * 2;
        pl.reg_stride = (int64_t)Tp * K * 2; pl.reg_next = time < Tp;
        }
        run_tail(e, {pl});
      } else {
        reg_decoder_output(e, s, cur, time - 1, Tp);
        run_hidden2grid<1>(e, S, S.cls_h[cur.cls[s]].p, S.out_cls_W->dev.p, logits,
                           lrow, rows_now);
      }
      if (one_per_sample) {
        const size_t total = (size_t)R * K;
        hipLaunchKernelGGL(tile_beam0_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                           e->stream, logits, K, B, total);
        if (single)
          hipLaunchKernelGGL(tile_beam0_kernel, dim3(cdiv(total * 2, 256)), dim3(256), 0,
                             e->stream, regstep, K * 2, B, total * 2);
      }
      int32_t* ids = e->bm_ids.p + (size_t)(time - 1) * R;
      int32_t* parents = e->bm_parents.p + (size_t)(time - 1) * R;
      if (dedupe)
        HIP_CHECK(hipMemsetAsync(e->bm_ref.p, 0, (size_t)R * sizeof(int32_t), e->stream));
      launch(e, "beam_step", 0, 4.0 * R * K, [&] {
        launch_beam_step(e->stream, logits, e->bm_lp[lpi].p, e->bm_cand.p, N, B, K, time,
                         c.diverse_beam, logf(c.diverse_gamma), c.fix_num_timestep,
                         e->bm_lp[lpi ^ 1].p, ids, parents, e->bm_src_row.p,
                         one_per_sample ? 1 : B, dedupe ? e->bm_ref.p : nullptr);
      });
      lpi ^= 1;
      src = e->bm_src_row.p;
      if (time == Tp) break;
      if (!sparse) run_emb_onehot(e, S, ids, 1, S.xbuf_cls.p, R);
    } else if (sparse) {
      // the embedded one-hot input enters the gate kernel as table terms (sparse_x.h)
    } else if (shared_first) {
      // one_hot(last observed cell) (:497-498, 531-532), one row per sample
      run_emb_onehot(e, S, S.labels.p + (T - 1), T, S.xbuf_cls.p, N, 1);
    } else {
      run_emb_onehot(e, S, S.labels.p + (T - 1), T, S.xbuf_cls.p, R, B);
    }
    if (c.use_gnn) {
      // time 0 (shared): N rows in, N rows out; afterwards R rows gathered through src
      // (which, after the first selection, indexes the N-row state)
      if (dedupe) {
        // on the state rows themselves (N of them up to the first selection)
        GnnJob job{&S, S.cls_h[cur.cls[s]].p, nullptr, S.cls_hg.p, rows_now,
                   one_per_sample ? 1 : B};
        job.row_ref = time >= 2 ? e->bm_ref.p : nullptr;
        run_gnn_jobs(e, {job});
      } else {
      const int out_rows = (shared_first && time == 0) ? N : R;
      run_gnn(e, S, S.cls_h[cur.cls[s]].p, src, S.cls_hg.p, out_rows,
              (shared_first && time == 0) ? 1 : B);
      }
    }
  }
  // back-trace (:689-806)
  hipLaunchKernelGGL(beam_backtrace_kernel, dim3(cdiv(R, 256)), dim3(256), 0,
                     e->stream, e->bm_ids.p, e->bm_parents.p, e->bm_out_ids.p,
                     e->bm_trace.p, N, B, Tp);
  const size_t total = (size_t)R * Tp * K;
  hipLaunchKernelGGL(beam_gather_logits_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                     e->stream, e->bm_logits.p, e->bm_trace.p, e->bm_out_logits.p,
                     N, B, Tp, K);
  if (c.use_single_decoder) {       // offsets along every beam: the same gather, 2K per row
    const size_t tot2 = (size_t)R * Tp * K * 2;
    hipLaunchKernelGGL(beam_gather_logits_kernel, dim3(cdiv(tot2, 256)), dim3(256), 0,
                       e->stream, e->bm_reg_steps.p, e->bm_trace.p, e->bm_out_reg.p,
                       N, B, Tp, K * 2);
  }
  // final logprobs are in bm_lp[lpi]
  if (lpi != 0)
    HIP_CHECK(hipMemcpyAsync(e->bm_lp[0].p, e->bm_lp[1].p, (size_t)R * sizeof(float),
                             hipMemcpyDeviceToDevice, e->stream));
}

void enqueue_forward(mv_engine* e, bool beam) {
  const mv_config& c = e->cfg;
  const int Tp = e->pred_len;
  run_scene(e);
  Cursors cur;
  run_encoders(e, cur);
  if (beam) {
    int s = 0;
    for (int i = 0; i < c.num_scales; ++i) if (e->sc[i].use) s = i;
    run_decoders_beam(e, s, cur, Tp);
  } else {
    run_decoders_greedy(e, cur, Tp);
  }
  HIP_CHECK(hipGetLastError());
}

// One forward = one `sess.run`.  In graph mode the ~150 (greedy) / ~120 (beam)
// launches of a forward are captured once per (mode, T_pred, U) into a hipGraph
// and replayed; every device pointer in it is engine-owned and stable.
void run_forward(mv_engine* e, bool beam) {
  MV_REQUIRE(e->inputs_ready, "no inputs uploaded (mv_upload_inputs)");
  ensure_params(e);
  if (beam)
    MV_REQUIRE(e->cfg.beam_size > 1, "engine was created with beam_size 1");
  if (!e->graph_mode || e->profiling) {
    enqueue_forward(e, beam);
    return;
  }
  const auto key = std::make_tuple(beam ? 1 : 0, e->pred_len, e->num_frames);
  auto it = e->graphs.find(key);
  if (it == e->graphs.end()) {
    hipGraph_t g = nullptr;
    HIP_CHECK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    try {
      enqueue_forward(e, beam);
    } catch (...) {
      (void)hipStreamEndCapture(e->stream, &g);
      if (g) (void)hipGraphDestroy(g);
      throw;
    }
    HIP_CHECK(hipStreamEndCapture(e->stream, &g));
    hipGraphExec_t ex = nullptr;
    hipError_t ie = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    HIP_CHECK(ie);
    it = e->graphs.emplace(key, ex).first;
  }
  HIP_CHECK(hipGraphLaunch(it->second, e->stream));
}

void upload_inputs(mv_engine* e, const mv_inputs* in) {
  const mv_config& c = e->cfg;
  const size_t N = c.batch_size, T = c.obs_len;
  MV_REQUIRE(in->obs_scene && in->scene_feat, "obs_scene / scene_feat is NULL");
  MV_REQUIRE(in->num_scene_frames >= 1 && (size_t)in->num_scene_frames <= N * T,
             "num_scene_frames %d not in [1, N*T_o=%zu]", in->num_scene_frames, N * T);
  MV_REQUIRE(in->pred_len >= 1 && in->pred_len <= c.max_pred_len,
             "pred_len %d not in [1, max_pred_len=%d]", in->pred_len, c.max_pred_len);
  for (size_t i = 0; i < N * T; ++i)
    MV_REQUIRE(in->obs_scene[i] >= 0 && in->obs_scene[i] < in->num_scene_frames,
               "obs_scene[%zu] = %d out of range [0,%d)", i, in->obs_scene[i],
               in->num_scene_frames);
  e->num_frames = in->num_scene_frames;
  e->pred_len = in->pred_len;
  HIP_CHECK(hipMemcpyAsync(e->obs_scene.p, in->obs_scene, N * T * sizeof(int32_t),
                           hipMemcpyHostToDevice, e->stream));
  HIP_CHECK(hipMemcpyAsync(e->scene_feat.p, in->scene_feat,
                           (size_t)e->num_frames * c.scene_h * c.scene_w *
                               c.scene_class * sizeof(float),
                           hipMemcpyHostToDevice, e->stream));
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    MV_REQUIRE(in->grid_obs_labels[s] && in->grid_obs_regress[s],
               "grid_obs_labels/grid_obs_regress[%d] is NULL for an enabled scale", s);
    for (size_t i = 0; i < N * T; ++i)
      MV_REQUIRE(in->grid_obs_labels[s][i] >= 0 && in->grid_obs_labels[s][i] < S.K,
                 "grid_obs_labels[%d][%zu] = %d out of range [0,%d)", s, i,
                 in->grid_obs_labels[s][i], S.K);
    HIP_CHECK(hipMemcpyAsync(S.labels.p, in->grid_obs_labels[s],
                             N * T * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    HIP_CHECK(hipMemcpyAsync(S.obs_reg.p, in->grid_obs_regress[s],
                             N * T * S.K * 2 * sizeof(float), hipMemcpyHostToDevice,
                             e->stream));
  }
  HIP_CHECK(hipStreamSynchronize(e->stream));
  e->inputs_ready = true;
}

// compact inputs: labels / scene indices as before, maps and masks expanded in HBM
void upload_inputs_compact(mv_engine* e, const mv_inputs_compact* in) {
  const mv_config& c = e->cfg;
  const size_t N = c.batch_size, T = c.obs_len;
  MV_REQUIRE(in->obs_scene && in->scene_feat_u8 && in->obs_xy,
             "obs_scene / scene_feat_u8 / obs_xy is NULL");
  MV_REQUIRE(in->num_scene_frames >= 1 && (size_t)in->num_scene_frames <= N * T,
             "num_scene_frames %d not in [1, N*T_o=%zu]", in->num_scene_frames, N * T);
  MV_REQUIRE(in->pred_len >= 1 && in->pred_len <= c.max_pred_len,
             "pred_len %d not in [1, max_pred_len=%d]", in->pred_len, c.max_pred_len);
  MV_REQUIRE(in->num_rows >= 0 && (size_t)in->num_rows <= N, "num_rows %d not in [0, N=%zu]",
             in->num_rows, N);
  for (size_t i = 0; i < N * T; ++i)
    MV_REQUIRE(in->obs_scene[i] >= 0 && in->obs_scene[i] < in->num_scene_frames,
               "obs_scene[%zu] = %d out of range [0,%d)", i, in->obs_scene[i],
               in->num_scene_frames);
  e->num_frames = in->num_scene_frames;
  e->pred_len = in->pred_len;
  HIP_CHECK(hipMemcpyAsync(e->obs_scene.p, in->obs_scene, N * T * sizeof(int32_t),
                           hipMemcpyHostToDevice, e->stream));
  const size_t nscene = (size_t)e->num_frames * c.scene_h * c.scene_w * c.scene_class;
  e->scene_u8.alloc(N * T * c.scene_h * c.scene_w * c.scene_class);
  HIP_CHECK(hipMemcpyAsync(e->scene_u8.p, in->scene_feat_u8, nscene, hipMemcpyHostToDevice,
                           e->stream));
  hipLaunchKernelGGL(mv::u8_to_f32_kernel, dim3(cdiv(nscene, 256)), dim3(256), 0, e->stream,
                     e->scene_u8.p, e->scene_feat.p, nscene);
  e->xy_dev.alloc(2 * N * std::max<size_t>(T, c.max_pred_len));
  HIP_CHECK(hipMemcpyAsync(e->xy_dev.p, in->obs_xy, 2 * N * T * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    MV_REQUIRE(in->grid_obs_labels[s], "grid_obs_labels[%d] is NULL for an enabled scale", s);
    MV_REQUIRE(S.centers.p, "mv_set_grid_centers(%d) has not been called", s);
    for (size_t i = 0; i < N * T; ++i)
      MV_REQUIRE(in->grid_obs_labels[s][i] >= 0 && in->grid_obs_labels[s][i] < S.K,
                 "grid_obs_labels[%d][%zu] = %d out of range [0,%d)", s, i,
                 in->grid_obs_labels[s][i], S.K);
    HIP_CHECK(hipMemcpyAsync(S.labels.p, in->grid_obs_labels[s],
                             N * T * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(mv::regress_from_xy_kernel, dim3(cdiv(N * T * S.K, 256)), dim3(256), 0,
                       e->stream, e->xy_dev.p, S.centers.p, S.obs_reg.p, (int)(N * T), (int)T,
                       S.K, in->num_rows);
  }
  HIP_CHECK(hipStreamSynchronize(e->stream));
  e->inputs_ready = true;
}

void download_outputs(mv_engine* e, mv_outputs* out) {
  const mv_config& c = e->cfg;
  const size_t N = c.batch_size, Tp = e->pred_len;
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    if (out->grid_pred_class[s])
      HIP_CHECK(hipMemcpyAsync(out->grid_pred_class[s], S.out_cls.p,
                               N * Tp * S.K * sizeof(float), hipMemcpyDeviceToHost,
                               e->stream));
    if (out->grid_pred_reg[s])
      HIP_CHECK(hipMemcpyAsync(out->grid_pred_reg[s], S.out_reg.p,
                               N * Tp * S.K * 2 * sizeof(float), hipMemcpyDeviceToHost,
                               e->stream));
  }
  HIP_CHECK(hipStreamSynchronize(e->stream));
}

// ---- pipelined greedy forward.  One `sess.run` of the reference is feed + compute + fetch,
// strictly in turn (code/pred_models.py:1761-1790).  An evaluation loop knows its next batch
// while the current one computes: mv_submit_greedy copies the caller's buffers into a pinned
// slot and queues H2D (copy stream) -> device staging -> [compute stream: D2D into the live
// input buffers, the forward, D2D of the outputs into the slot] -> D2H (copy stream) and
// returns; mv_collect_greedy waits for the OLDEST submission and hands its outputs over.
// With two slots the PCIe traffic of batches k+1 and k-1 runs under the kernels of batch k.
// Layout of a slot: obs_scene | scene_feat (N*T frames max) | per used scale labels,
// obs_regress || per used scale out_cls, out_reg (max_pred_len).
struct PipeLayout {
  size_t obs_scene = 0, scene_feat = 0, labels[MV_MAX_SCALES] = {0, 0},
         obs_reg[MV_MAX_SCALES] = {0, 0}, in_bytes = 0;
  size_t out_cls[MV_MAX_SCALES] = {0, 0}, out_reg[MV_MAX_SCALES] = {0, 0}, out_bytes = 0;
};
static PipeLayout pipe_layout(const mv_engine* e) {
  const mv_config& c = e->cfg;
  const size_t N = c.batch_size, T = c.obs_len, Tp = c.max_pred_len;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  PipeLayout L;
  size_t o = 0;
  L.obs_scene = o; o = al(o + N * T * sizeof(int32_t));
  L.scene_feat = o; o = al(o + N * T * c.scene_h * c.scene_w * c.scene_class * sizeof(float));
  for (int s = 0; s < c.num_scales; ++s) {
    if (!e->sc[s].use) continue;
    const size_t K = e->sc[s].K;
    L.labels[s] = o; o = al(o + N * T * sizeof(int32_t));
    L.obs_reg[s] = o; o = al(o + N * T * K * 2 * sizeof(float));
  }
  L.in_bytes = o;
  for (int s = 0; s < c.num_scales; ++s) {
    if (!e->sc[s].use) continue;
    const size_t K = e->sc[s].K;
    L.out_cls[s] = o; o = al(o + N * Tp * K * sizeof(float));
    L.out_reg[s] = o; o = al(o + N * Tp * K * 2 * sizeof(float));
  }
  L.out_bytes = o - L.in_bytes;
  return L;
}

void pipeline_create(mv_engine* e, int depth) {
  MV_REQUIRE(depth >= 1 && depth <= 8, "pipeline depth %d not in [1, 8]", depth);
  MV_REQUIRE(e->cfg.beam_size == 1, "the pipelined forward is the greedy one");
  MV_REQUIRE(e->pipe.empty(), "pipeline already created");
  const PipeLayout L = pipe_layout(e);
  HIP_CHECK(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&e->fetch_stream, hipStreamNonBlocking));
  e->pipe.resize(depth);
  for (auto& sl : e->pipe) {
    sl.in_bytes = L.in_bytes; sl.out_bytes = L.out_bytes;
    HIP_CHECK(hipHostMalloc(&sl.pin, L.in_bytes + L.out_bytes, hipHostMallocDefault));
    HIP_CHECK(hipMalloc((void**)&sl.dev, L.in_bytes + L.out_bytes));
    HIP_CHECK(hipEventCreateWithFlags(&sl.h2d, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&sl.d2h, hipEventDisableTiming));
  }
  e->pipe_head = e->pipe_tail = 0;
}

void pipeline_destroy(mv_engine* e) {
  for (auto& sl : e->pipe) {
    if (sl.pin) (void)hipHostFree(sl.pin);
    if (sl.dev) (void)hipFree(sl.dev);
    if (sl.h2d) (void)hipEventDestroy(sl.h2d);
    if (sl.done) (void)hipEventDestroy(sl.done);
    if (sl.d2h) (void)hipEventDestroy(sl.d2h);
  }
  e->pipe.clear();
  if (e->copy_stream) { (void)hipStreamDestroy(e->copy_stream); e->copy_stream = nullptr; }
  if (e->fetch_stream) { (void)hipStreamDestroy(e->fetch_stream); e->fetch_stream = nullptr; }
}

void pipeline_submit(mv_engine* e, const mv_inputs* in) {
  const mv_config& c = e->cfg;
  const size_t N = c.batch_size, T = c.obs_len;
  MV_REQUIRE(!e->pipe.empty(), "mv_pipeline_create has not been called");
  mv_engine::PipeSlot& sl = e->pipe[e->pipe_head % e->pipe.size()];
  MV_REQUIRE(!sl.busy, "pipeline full: %zu submissions not collected (mv_collect_greedy)",
             e->pipe.size());
  MV_REQUIRE(in->obs_scene && in->scene_feat, "obs_scene / scene_feat is NULL");
  MV_REQUIRE(in->num_scene_frames >= 1 && (size_t)in->num_scene_frames <= N * T,
             "num_scene_frames %d not in [1, N*T_o=%zu]", in->num_scene_frames, N * T);
  MV_REQUIRE(in->pred_len >= 1 && in->pred_len <= c.max_pred_len,
             "pred_len %d not in [1, max_pred_len=%d]", in->pred_len, c.max_pred_len);
  for (size_t i = 0; i < N * T; ++i)
    MV_REQUIRE(in->obs_scene[i] >= 0 && in->obs_scene[i] < in->num_scene_frames,
               "obs_scene[%zu] = %d out of range [0,%d)", i, in->obs_scene[i],
               in->num_scene_frames);
  const PipeLayout L = pipe_layout(e);
  char* pin = static_cast<char*>(sl.pin);
  const size_t sf_bytes = (size_t)in->num_scene_frames * c.scene_h * c.scene_w *
                          c.scene_class * sizeof(float);
  memcpy(pin + L.obs_scene, in->obs_scene, N * T * sizeof(int32_t));
  memcpy(pin + L.scene_feat, in->scene_feat, sf_bytes);
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    MV_REQUIRE(in->grid_obs_labels[s] && in->grid_obs_regress[s],
               "grid_obs_labels/grid_obs_regress[%d] is NULL for an enabled scale", s);
    for (size_t i = 0; i < N * T; ++i)
      MV_REQUIRE(in->grid_obs_labels[s][i] >= 0 && in->grid_obs_labels[s][i] < S.K,
                 "grid_obs_labels[%d][%zu] = %d out of range [0,%d)", s, i,
                 in->grid_obs_labels[s][i], S.K);
    memcpy(pin + L.labels[s], in->grid_obs_labels[s], N * T * sizeof(int32_t));
    memcpy(pin + L.obs_reg[s], in->grid_obs_regress[s], N * T * S.K * 2 * sizeof(float));
  }
  sl.num_frames = in->num_scene_frames; sl.pred_len = in->pred_len;
  // copy stream: the whole input block in one transfer (it must not start before the
  // slot's previous fetch has left the same pinned / staging buffers: collect waited d2h)
  HIP_CHECK(hipMemcpyAsync(sl.dev, sl.pin, L.in_bytes, hipMemcpyHostToDevice, e->copy_stream));
  HIP_CHECK(hipEventRecord(sl.h2d, e->copy_stream));
  // compute stream: staging -> live inputs, forward, outputs -> staging
  HIP_CHECK(hipStreamWaitEvent(e->stream, sl.h2d, 0));
  auto d2d = [&](void* dst, const void* src, size_t n) {
    HIP_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, e->stream));
  };
  d2d(e->obs_scene.p, sl.dev + L.obs_scene, N * T * sizeof(int32_t));
  d2d(e->scene_feat.p, sl.dev + L.scene_feat, sf_bytes);
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    d2d(S.labels.p, sl.dev + L.labels[s], N * T * sizeof(int32_t));
    d2d(S.obs_reg.p, sl.dev + L.obs_reg[s], N * T * S.K * 2 * sizeof(float));
  }
  e->num_frames = sl.num_frames;
  e->pred_len = sl.pred_len;
  e->inputs_ready = true;
  run_forward(e, false);
  const size_t Tp = sl.pred_len;
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    d2d(sl.dev + L.out_cls[s], S.out_cls.p, N * Tp * S.K * sizeof(float));
    d2d(sl.dev + L.out_reg[s], S.out_reg.p, N * Tp * S.K * 2 * sizeof(float));
  }
  HIP_CHECK(hipEventRecord(sl.done, e->stream));
  // fetch stream
  HIP_CHECK(hipStreamWaitEvent(e->fetch_stream, sl.done, 0));
  HIP_CHECK(hipMemcpyAsync(pin + L.in_bytes, sl.dev + L.in_bytes, L.out_bytes,
                           hipMemcpyDeviceToHost, e->fetch_stream));
  HIP_CHECK(hipEventRecord(sl.d2h, e->fetch_stream));
  sl.busy = true;
  e->pipe_head += 1;
}

void pipeline_collect(mv_engine* e, mv_outputs* out) {
  const mv_config& c = e->cfg;
  MV_REQUIRE(!e->pipe.empty(), "mv_pipeline_create has not been called");
  mv_engine::PipeSlot& sl = e->pipe[e->pipe_tail % e->pipe.size()];
  MV_REQUIRE(sl.busy, "mv_collect_greedy: nothing submitted");
  HIP_CHECK(hipEventSynchronize(sl.d2h));
  const PipeLayout L = pipe_layout(e);
  const char* pin = static_cast<const char*>(sl.pin);
  const size_t N = c.batch_size, Tp = sl.pred_len;
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    if (out->grid_pred_class[s])
      memcpy(out->grid_pred_class[s], pin + L.out_cls[s], N * Tp * S.K * sizeof(float));
    if (out->grid_pred_reg[s])
      memcpy(out->grid_pred_reg[s], pin + L.out_reg[s], N * Tp * S.K * 2 * sizeof(float));
  }
  sl.busy = false;
  e->pipe_tail += 1;
}

void download_beam(mv_engine* e, mv_beam_outputs* out) {
  const mv_config& c = e->cfg;
  const size_t N = c.batch_size, Tp = e->pred_len, B = c.beam_size;
  int s = 0;
  for (int i = 0; i < c.num_scales; ++i) if (e->sc[i].use) s = i;
  ScaleState& S = e->sc[s];
  const size_t K = S.K;
  if (out->logits)
    HIP_CHECK(hipMemcpyAsync(out->logits, e->bm_out_logits.p, N * B * Tp * K * sizeof(float),
                             hipMemcpyDeviceToHost, e->stream));
  if (out->ids)
    HIP_CHECK(hipMemcpyAsync(out->ids, e->bm_out_ids.p, N * B * Tp * sizeof(int32_t),
                             hipMemcpyDeviceToHost, e->stream));
  if (out->logprobs)
    HIP_CHECK(hipMemcpyAsync(out->logprobs, e->bm_lp[0].p, N * B * sizeof(float),
                             hipMemcpyDeviceToHost, e->stream));
  if (out->grid_reg) {
    if (c.use_single_decoder)       // per beam: [N*B, T, K, 2]
      HIP_CHECK(hipMemcpyAsync(out->grid_reg, e->bm_out_reg.p,
                               N * B * Tp * K * 2 * sizeof(float), hipMemcpyDeviceToHost,
                               e->stream));
    else
    HIP_CHECK(hipMemcpyAsync(out->grid_reg, S.out_reg.p, N * Tp * K * 2 * sizeof(float),
                             hipMemcpyDeviceToHost, e->stream));
  }
  if (out->best_beam)  // logits[:, 0] -> [N, T, K]: rows n*B of [N,B,T,K]
    HIP_CHECK(hipMemcpy2DAsync(out->best_beam, Tp * K * sizeof(float),
                               e->bm_out_logits.p, B * Tp * K * sizeof(float),
                               Tp * K * sizeof(float), N, hipMemcpyDeviceToHost,
                               e->stream));
  HIP_CHECK(hipStreamSynchronize(e->stream));
}

template <typename F>
int guarded(mv_engine* e, F&& fn) {
  try {
    if (e) HIP_CHECK(hipSetDevice(e->device));
    fn();
    return 0;
  } catch (const HipError& err) {
    if (e) e->err = err.msg; else g_create_error = err.msg;
    return 1;
  } catch (const std::exception& ex) {
    if (e) e->err = ex.what(); else g_create_error = ex.what();
    return 2;
  }
}

}  // namespace

#include "engine_train.h"

namespace {

// RAII device buffer helpers for the single-kernel entry points
struct OpCtx {
  int device;
  hipStream_t stream = nullptr;
  explicit OpCtx(int dev) : device(dev) {
    HIP_CHECK(hipSetDevice(dev));
    HIP_CHECK(hipStreamCreate(&stream));
  }
  ~OpCtx() { if (stream) (void)hipStreamDestroy(stream); }
  template <typename T>
  void up(DevBuf<T>& b, const T* src, size_t n) {
    b.alloc(n ? n : 1);
    if (n) HIP_CHECK(hipMemcpy(b.p, src, n * sizeof(T), hipMemcpyHostToDevice));
  }
  template <typename T>
  void down(T* dst, const DevBuf<T>& b, size_t n) {
    HIP_CHECK(hipStreamSynchronize(stream));
    if (n) HIP_CHECK(hipMemcpy(dst, b.p, n * sizeof(T), hipMemcpyDeviceToHost));
  }
};

}  // namespace

// ===================================================================== C ABI

extern "C" {

int mv_abi_version(void) { return MV_ABI_VERSION; }

const char* mv_last_error(mv_handle h) {
  return h ? h->err.c_str() : g_create_error.c_str();
}

int mv_create(const mv_config* cfg, int device, mv_handle* out) {
  if (!cfg || !out) { g_create_error = "mv_create: NULL argument"; return 1; }
  *out = nullptr;
  mv_engine* e = nullptr;
  int rc = guarded(nullptr, [&] {
    validate_config(*cfg);
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    MV_REQUIRE(device >= 0 && device < ndev, "device %d not in [0,%d)", device, ndev);
    HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    MV_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
               "device %d is %s; this library contains gfx950 (MI355X) code only",
               device, prop.gcnArchName);
    e = new mv_engine();
    e->cfg = *cfg;
    e->device = device;
    HIP_CHECK(hipStreamCreate(&e->stream));
    build_param_table(e);
    alloc_buffers(e);
    if (cfg->beam_size > 1) {
      size_t K = 0;
      for (int s = 0; s < cfg->num_scales; ++s)
        if (e->sc[s].use) K = e->sc[s].K;
      ensure_beam_step_lds(device, ((size_t)2 * cfg->beam_size * K + 512) * sizeof(float));
    }
  });
  if (rc != 0) { delete e; return rc; }
  *out = e;
  return 0;
}

int mv_destroy(mv_handle h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
  if (h->fetch_stream) (void)hipStreamSynchronize(h->fetch_stream);
  pipeline_destroy(h);
  h->drop_graphs();
  if (h->comm) {
    if (h->comm->stream) (void)hipStreamSynchronize(h->comm->stream);
    if (h->comm->comm) (void)mv::rccl().CommDestroy(h->comm->comm);
    if (h->comm->ready) (void)hipEventDestroy(h->comm->ready);
    if (h->comm->done) (void)hipEventDestroy(h->comm->done);
    if (h->comm->stream) (void)hipStreamDestroy(h->comm->stream);
    delete h->comm;
  }
  delete h->train;
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int mv_num_params(mv_handle h) { return h ? (int)h->params.size() : -1; }

int mv_param_info(mv_handle h, int32_t i, char* name_out, int32_t name_cap,
                  int64_t* shape_out) {
  if (!h || i < 0 || i >= (int)h->params.size()) return -1;
  Param* p = h->params[i].get();
  if (name_out && name_cap > 0) {
    strncpy(name_out, p->name.c_str(), name_cap - 1);
    name_out[name_cap - 1] = 0;
  }
  if (shape_out)
    for (size_t d = 0; d < 4; ++d) shape_out[d] = d < p->shape.size() ? p->shape[d] : 0;
  return (int)p->shape.size();
}

int mv_set_param(mv_handle h, const char* tf_name, const float* data,
                 const int64_t* shape, int32_t rank) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(tf_name && data && shape, "mv_set_param: NULL argument");
    auto it = h->by_name.find(tf_name);
    MV_REQUIRE(it != h->by_name.end(), "unknown parameter '%s'", tf_name);
    Param* p = it->second;
    MV_REQUIRE(rank == (int)p->shape.size(), "parameter %s: rank %d, expected %zu",
               tf_name, rank, p->shape.size());
    for (int d = 0; d < rank; ++d)
      MV_REQUIRE(shape[d] == p->shape[d], "parameter %s: dim %d is %lld, expected %lld",
                 tf_name, d, (long long)shape[d], (long long)p->shape[d]);
    const size_t n = p->elems();
    p->host.assign(data, data + n);
    p->dev.alloc(n);
    HIP_CHECK(hipMemcpy(p->dev.p, data, n * sizeof(float), hipMemcpyHostToDevice));
    p->set = true;
    h->drop_graphs();   // captured launches hold the old device pointers
    h->train_packs_valid = false;
    for (int s = 0; s < h->cfg.num_scales; ++s) h->sc[s].wq_valid = h->sc[s].sx_valid = false;
    // invalidate the packed copy of a ConvLSTM kernel
    for (int s = 0; s < h->cfg.num_scales; ++s) {
      ScaleState& S = h->sc[s];
      for (ConvCell* cc : {&S.enc_cls, &S.enc_reg, &S.dec_cls, &S.dec_reg})
        if (cc->kernel == p) {
          cc->wpack.release(); cc->wp16.release(); cc->wx32.release();
          cc->wpb.release(); cc->wx32u.release(); cc->wpw.release(); cc->wpw3.release();
          cc->host_stale = false;
        }
    }
  });
}

int mv_get_param(mv_handle h, const char* tf_name, float* out, int64_t capacity) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(tf_name && out, "mv_get_param: NULL argument");
    auto it = h->by_name.find(tf_name);
    MV_REQUIRE(it != h->by_name.end(), "unknown parameter '%s'", tf_name);
    Param* p = it->second;
    MV_REQUIRE(p->set, "parameter %s not set", tf_name);
    MV_REQUIRE((size_t)capacity >= p->elems(), "buffer too small for %s", tf_name);
    HIP_CHECK(hipMemcpy(out, p->dev.p, p->elems() * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int mv_upload_inputs(mv_handle h, const mv_inputs* in) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(in, "mv_upload_inputs: NULL inputs");
    upload_inputs(h, in);
  });
}

int mv_set_grid_centers(mv_handle h, int32_t scale, const double* centers) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(scale >= 0 && scale < h->cfg.num_scales && h->sc[scale].use,
               "mv_set_grid_centers: scale %d is not an enabled scale", scale);
    MV_REQUIRE(centers, "mv_set_grid_centers: NULL centers");
    ScaleState& S = h->sc[scale];
    S.centers.alloc((size_t)S.K * 2);
    HIP_CHECK(hipMemcpy(S.centers.p, centers, (size_t)S.K * 2 * sizeof(double),
                        hipMemcpyHostToDevice));
  });
}

int mv_upload_inputs_compact(mv_handle h, const mv_inputs_compact* in) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(in, "mv_upload_inputs_compact: NULL inputs");
    upload_inputs_compact(h, in);
  });
}

int mv_run_greedy_resident(mv_handle h) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->cfg.beam_size == 1, "engine was created for beam search");
    run_forward(h, false);
  });
}

int mv_run_beam_resident(mv_handle h) {
  if (!h) return 1;
  return guarded(h, [&] { run_forward(h, true); });
}

int mv_synchronize(mv_handle h) {
  if (!h) return 1;
  return guarded(h, [&] {
    HIP_CHECK(hipStreamSynchronize(h->stream));
    drain_events(h);
  });
}

int mv_download_outputs(mv_handle h, mv_outputs* out) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(out, "NULL outputs");
    download_outputs(h, out);
  });
}

int mv_download_beam_outputs(mv_handle h, mv_beam_outputs* out) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(out, "NULL outputs");
    download_beam(h, out);
  });
}

int mv_forward_greedy(mv_handle h, const mv_inputs* in, mv_outputs* out) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(in && out, "mv_forward_greedy: NULL argument");
    MV_REQUIRE(h->cfg.beam_size == 1, "engine was created for beam search");
    upload_inputs(h, in);
    run_forward(h, false);
    download_outputs(h, out);
    drain_events(h);
  });
}

int mv_pipeline_create(mv_handle h, int32_t depth) {
  if (!h) return 1;
  return guarded(h, [&] { pipeline_create(h, depth); });
}

int mv_submit_greedy(mv_handle h, const mv_inputs* in) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(in, "mv_submit_greedy: NULL argument");
    pipeline_submit(h, in);
  });
}

int mv_collect_greedy(mv_handle h, mv_outputs* out, int32_t* pred_len) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(out, "mv_collect_greedy: NULL argument");
    if (pred_len && !h->pipe.empty())
      *pred_len = h->pipe[h->pipe_tail % h->pipe.size()].pred_len;
    pipeline_collect(h, out);
    drain_events(h);
  });
}

int mv_forward_beam(mv_handle h, const mv_inputs* in, mv_beam_outputs* out) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(in && out, "mv_forward_beam: NULL argument");
    upload_inputs(h, in);
    run_forward(h, true);
    download_beam(h, out);
    drain_events(h);
  });
}

int mv_set_profiling(mv_handle h, int32_t enabled) {
  if (!h) return 1;
  h->profiling = enabled != 0;
  return 0;
}

int mv_train_init(mv_handle h, const mv_train_config* tc) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(tc, "mv_train_init: NULL config");
    MV_REQUIRE(tc->optimizer >= 0 && tc->optimizer <= 3, "Optimizer not implemented: %d "
               "(0 adadelta, 1 momentum, 2 adam, 3 rmsprop; reference pred_models.py:1667-1681)",
               tc->optimizer);
    MV_REQUIRE(tc->class_feedback >= 0 && tc->class_feedback <= 2, "class_feedback %d",
               tc->class_feedback);
    MV_REQUIRE(tc->keep_prob > 0.f && tc->keep_prob <= 1.f, "keep_prob %g not in (0, 1]",
               tc->keep_prob);
    if (tc->use_soft_grid_class)
      MV_REQUIRE(tc->soft_kernel_size == 3 || tc->soft_kernel_size == 5,
                 "soft_kernel_size %d (3 or 5)", tc->soft_kernel_size);
    MV_REQUIRE(h->cfg.beam_size == 1, "training needs a greedy (beam_size 1) engine "
               "(reference pred_models.py:261)");
    MV_REQUIRE(!(h->compute_mode == 2 && h->cfg.activation != 0),
               "training in compute mode 2 (bf16) needs activation_func tanh: relu / lrelu "
               "models train in mode 1 (f16x3); see mv_config.activation");
    if (!h->train) {
      h->train = new mv_train_holder();
      train_alloc(h);
      // sparse_x_on() turns false once a training state exists (the backward pass needs
      // the dense x operand), and the graph key does not carry that: a forward captured
      // before this call would keep replaying the sparse-x launches against tables that
      // train_apply no longer rebuilds.  Forget the captures and the derived tables.
      h->drop_graphs();
      for (int s = 0; s < h->cfg.num_scales; ++s) h->sc[s].wq_valid = h->sc[s].sx_valid = false;
    }
    TrainState& t = h->train->st;
    t.tc = *tc;
    if (t.slots_for != tc->optimizer) {
      // slot initial values as TF creates them: zeros, except RMSProp's `rms` = ones
      HIP_CHECK(hipMemsetAsync(t.accum.p, 0, t.total_elems * sizeof(float), h->stream));
      HIP_CHECK(hipMemsetAsync(t.accum_update.p, 0, t.total_elems * sizeof(float), h->stream));
      if (tc->optimizer == 3)
        hipLaunchKernelGGL(mv::fill_kernel, dim3(cdiv(t.total_elems, 256)), dim3(256), 0,
                           h->stream, t.accum.p, 1.0f, t.total_elems);
      t.beta1_power = 0.9f; t.beta2_power = 0.999f;
      HIP_CHECK(hipStreamSynchronize(h->stream));
      t.slots_for = tc->optimizer;
    }
  });
}

int mv_train_forward_backward(mv_handle h, const mv_inputs* in, const mv_targets* tg,
                              mv_losses* out) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE((in == nullptr) == (tg == nullptr),
               "mv_train_forward_backward: give both inputs and targets, or neither "
               "(resident)");
    train_fwd_bwd(h, in, tg, out);
  });
}

int mv_upload_targets(mv_handle h, const mv_targets* tg) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->train, "mv_train_init has not been called");
    MV_REQUIRE(tg, "mv_upload_targets: NULL targets");
    MV_REQUIRE(h->inputs_ready, "mv_upload_inputs first (it fixes T_pred)");
    upload_targets(h, tg);
    TS(h).targets_ready = true;
  });
}

int mv_upload_targets_compact(mv_handle h, const mv_targets_compact* tg) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->train, "mv_train_init has not been called");
    MV_REQUIRE(tg, "mv_upload_targets_compact: NULL targets");
    MV_REQUIRE(h->inputs_ready, "mv_upload_inputs first (it fixes T_pred)");
    upload_targets_compact(h, tg);
    TS(h).targets_ready = true;
  });
}

int mv_train_apply(mv_handle h, float grad_scale) {
  if (!h) return 1;
  return guarded(h, [&] { train_apply(h, grad_scale); });
}

int mv_train_step(mv_handle h, const mv_inputs* in, const mv_targets* tg, mv_losses* out) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE((in == nullptr) == (tg == nullptr),
               "mv_train_step: give both inputs and targets, or neither (resident)");
    train_fwd_bwd(h, in, tg, out);
    // with a communicator the gradients are the SUM over the ranks (reduced inside
    // train_fwd_bwd, overlapped with the backward pass): mean, then clip + optimizer
    train_apply(h, h->comm ? 1.0f / (float)h->comm->world : 1.0f);
  });
}

int mv_grad_buffer(mv_handle h, float** device_ptr, int64_t* elems) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->train, "mv_train_init has not been called");
    MV_REQUIRE(device_ptr && elems, "mv_grad_buffer: NULL argument");
    *device_ptr = TS(h).grad.p;
    *elems = (int64_t)TS(h).total_elems;
  });
}

static Param* find_param(mv_handle h, const char* tf_name) {
  MV_REQUIRE(tf_name, "NULL parameter name");
  auto it = h->by_name.find(tf_name);
  MV_REQUIRE(it != h->by_name.end(), "unknown parameter '%s'", tf_name);
  return it->second;
}

int mv_get_grad(mv_handle h, const char* tf_name, float* out, int64_t capacity) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->train && TS(h).have_grads, "no gradients (mv_train_forward_backward)");
    Param* p = find_param(h, tf_name);
    MV_REQUIRE(out && (size_t)capacity >= p->elems(), "buffer too small for %s", tf_name);
    HIP_CHECK(hipMemcpy(out, grad_of(h, p), p->elems() * sizeof(float),
                        hipMemcpyDeviceToHost));
  });
}

int mv_get_global_step(mv_handle h, int64_t* step) {
  if (!h || !step || !h->train) return 1;
  *step = h->train->st.global_step;
  return 0;
}

int mv_set_global_step(mv_handle h, int64_t step) {
  if (!h || !h->train) return 1;
  h->train->st.global_step = step;
  return 0;
}

int mv_comm_unique_id(uint8_t* id_out) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(id_out, "mv_comm_unique_id: NULL buffer");
    mv::RcclApi& r = mv::rccl();
    MV_REQUIRE(r.ok, "RCCL unavailable: %s", r.error.c_str());
    ncclUniqueId id;
    ncclResult_t rc = r.GetUniqueId(&id);
    MV_REQUIRE(rc == ncclSuccess, "ncclGetUniqueId: %s", r.GetErrorString(rc));
    static_assert(sizeof(id) == MV_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
  });
}

int mv_allreduce_init(mv_handle h, int32_t rank, int32_t world, const uint8_t* unique_id) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(unique_id && world >= 1 && rank >= 0 && rank < world,
               "mv_allreduce_init: bad arguments (rank %d of %d)", rank, world);
    MV_REQUIRE(!h->comm, "mv_allreduce_init: communicator already initialised");
    mv::RcclApi& r = mv::rccl();
    MV_REQUIRE(r.ok, "RCCL unavailable: %s", r.error.c_str());
    std::unique_ptr<mv::Comm> c(new mv::Comm());
    c->rank = rank; c->world = world;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t rc = r.CommInitRank(&c->comm, world, id, rank);   // device: hipSetDevice above
    MV_REQUIRE(rc == ncclSuccess, "ncclCommInitRank(rank %d of %d): %s", rank, world,
               r.GetErrorString(rc));
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    h->comm = c.release();
  });
}

int mv_allreduce_info(mv_handle h, int32_t* rank, int32_t* world, int32_t* buckets,
                      double* bytes) {
  if (!h || !h->comm) return 1;
  if (rank) *rank = h->comm->rank;
  if (world) *world = h->comm->world;
  if (buckets) *buckets = h->comm->buckets_last;
  if (bytes) *bytes = h->comm->bytes_last;
  return 0;
}

// ---- SimAug training extras: targeted attacks on the scene features
int mv_attack_begin(mv_handle h) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->train, "mv_train_init has not been called");
    MV_REQUIRE(h->inputs_ready, "no inputs uploaded");
    TrainState& t = TS(h);
    const size_t n = (size_t)h->num_frames * h->cfg.scene_h * h->cfg.scene_w * h->cfg.scene_class;
    t.scene_clean.alloc((size_t)h->cfg.batch_size * h->cfg.obs_len * h->cfg.scene_h *
                        h->cfg.scene_w * h->cfg.scene_class);
    HIP_CHECK(hipMemcpyAsync(t.scene_clean.p, h->scene_feat.p, n * sizeof(float),
                             hipMemcpyDeviceToDevice, h->stream));
    t.want_dscene = true;
    t.have_dscene = false;
  });
}

int mv_attack_end(mv_handle h) {
  if (!h || !h->train) return 1;
  TS(h).want_dscene = false;
  return 0;
}

int mv_set_scene_feat(mv_handle h, const float* scene_feat) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(scene_feat && h->inputs_ready, "mv_set_scene_feat: NULL / no inputs uploaded");
    const size_t n = (size_t)h->num_frames * h->cfg.scene_h * h->cfg.scene_w * h->cfg.scene_class;
    HIP_CHECK(hipMemcpy(h->scene_feat.p, scene_feat, n * sizeof(float), hipMemcpyHostToDevice));
  });
}

int mv_get_scene_feat(mv_handle h, float* out) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(out && h->inputs_ready, "mv_get_scene_feat: NULL / no inputs uploaded");
    const size_t n = (size_t)h->num_frames * h->cfg.scene_h * h->cfg.scene_w * h->cfg.scene_class;
    HIP_CHECK(hipStreamSynchronize(h->stream));
    HIP_CHECK(hipMemcpy(out, h->scene_feat.p, n * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int mv_get_scene_grad(mv_handle h, float* out) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->train && TS(h).have_dscene, "no input gradient (mv_attack_begin, then "
               "mv_train_forward_backward)");
    const size_t n = (size_t)h->num_frames * h->cfg.scene_h * h->cfg.scene_w * h->cfg.scene_class;
    HIP_CHECK(hipStreamSynchronize(h->stream));
    HIP_CHECK(hipMemcpy(out, TS(h).dscene.p, n * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int mv_attack_step(mv_handle h, float epsilon, float step) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->train && TS(h).have_dscene && TS(h).scene_clean.p,
               "mv_attack_step: mv_attack_begin + mv_train_forward_backward first");
    const size_t n = (size_t)h->num_frames * h->cfg.scene_h * h->cfg.scene_w * h->cfg.scene_class;
    hipLaunchKernelGGL(mv::adv_step_kernel, dim3(cdiv(n, 256)), dim3(256), 0, h->stream,
                       h->scene_feat.p, TS(h).dscene.p, TS(h).scene_clean.p, epsilon, step, n);
    HIP_CHECK(hipGetLastError());
  });
}

int mv_scene_mix(mv_handle h, const float* other, float weight) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->train && h->inputs_ready, "mv_scene_mix: training engine with inputs");
    const size_t n = (size_t)h->num_frames * h->cfg.scene_h * h->cfg.scene_w * h->cfg.scene_class;
    TrainState& t = TS(h);
    const float* src = t.scene_clean.p;        // other == NULL: mix with the clean features
    DevBuf<float> tmp;
    if (other) {
      tmp.alloc(n);
      HIP_CHECK(hipMemcpy(tmp.p, other, n * sizeof(float), hipMemcpyHostToDevice));
      src = tmp.p;
    }
    MV_REQUIRE(src, "mv_scene_mix: no clean copy (mv_attack_begin)");
    hipLaunchKernelGGL(mv::mix_kernel, dim3(cdiv(n, 256)), dim3(256), 0, h->stream,
                       h->scene_feat.p, src, weight, n);
    HIP_CHECK(hipStreamSynchronize(h->stream));
  });
}

int mv_get_sample_losses(mv_handle h, int32_t scale, float* out) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->train && TS(h).have_grads, "no losses (mv_train_forward_backward)");
    MV_REQUIRE(scale >= 0 && scale < h->cfg.num_scales && h->sc[scale].use && out,
               "mv_get_sample_losses: scale %d", scale);
    const int N = h->cfg.batch_size, Tp = h->pred_len;
    TrainState& t = TS(h);
    hipLaunchKernelGGL(mv::loss_rows_mean_kernel, dim3(cdiv(N, 256)), dim3(256), 0, h->stream,
                       t.sc[scale].loss_row.p, t.scratch.p, Tp, N);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    HIP_CHECK(hipMemcpy(out, t.scratch.p, N * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int mv_set_label_mixup(mv_handle h, const int32_t* const* obs_labels2,
                       const int32_t* const* pred_labels2, float weight,
                       const float* sample_weight) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->train, "mv_train_init has not been called");
    MV_REQUIRE(obs_labels2 && pred_labels2, "mv_set_label_mixup: NULL argument");
    MV_REQUIRE(weight >= 0.f && weight <= 1.f, "mv_set_label_mixup: weight %g not in [0, 1]",
               (double)weight);
    TrainState& t = TS(h);
    const int N = h->cfg.batch_size, To = h->cfg.obs_len, Tp = h->pred_len;
    for (int s = 0; s < h->cfg.num_scales; ++s) {
      if (!h->sc[s].use) continue;
      MV_REQUIRE(obs_labels2[s] && pred_labels2[s], "mv_set_label_mixup: scale %d is NULL", s);
      const int K = h->sc[s].K;
      for (int i = 0; i < N * To; ++i)
        MV_REQUIRE(obs_labels2[s][i] >= 0 && obs_labels2[s][i] < K,
                   "mv_set_label_mixup: observed label %d out of range", obs_labels2[s][i]);
      for (int i = 0; i < N * Tp; ++i)
        MV_REQUIRE(pred_labels2[s][i] >= 0 && pred_labels2[s][i] < K,
                   "mv_set_label_mixup: future label %d out of range", pred_labels2[s][i]);
      HIP_CHECK(hipMemcpy(t.sc[s].obs_labels2.p, obs_labels2[s], (size_t)N * To * sizeof(int32_t),
                          hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(t.sc[s].pred_labels2.p, pred_labels2[s],
                          (size_t)N * Tp * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if (sample_weight)
      HIP_CHECK(hipMemcpy(t.sample_w.p, sample_weight, (size_t)N * sizeof(float),
                          hipMemcpyHostToDevice));
    t.mix_on = true; t.mix_sw = sample_weight != nullptr; t.mix_w = weight;
  });
}

int mv_clear_label_mixup(mv_handle h) {
  if (!h || !h->train) return 1;
  TS(h).mix_on = TS(h).mix_sw = false;
  TS(h).mix_w = 1.f;
  return 0;
}

int mv_set_dropout_seed(mv_handle h, uint32_t seed) {
  if (!h || !h->train) return 1;
  h->train->st.dropout_seed = seed;
  return 0;
}

int mv_get_opt_scalars(mv_handle h, float* beta1_power, float* beta2_power) {
  if (!h || !h->train || !beta1_power || !beta2_power) return 1;
  *beta1_power = h->train->st.beta1_power;
  *beta2_power = h->train->st.beta2_power;
  return 0;
}

int mv_set_opt_scalars(mv_handle h, float beta1_power, float beta2_power) {
  if (!h || !h->train) return 1;
  h->train->st.beta1_power = beta1_power;
  h->train->st.beta2_power = beta2_power;
  return 0;
}

int mv_get_opt_slot(mv_handle h, const char* tf_name, int32_t slot, float* out,
                    int64_t capacity) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->train, "mv_train_init has not been called");
    MV_REQUIRE(slot == 0 || slot == 1, "slot must be 0 or 1 (see multiverse_hip.h)");
    Param* p = find_param(h, tf_name);
    MV_REQUIRE(out && (size_t)capacity >= p->elems(), "buffer too small for %s", tf_name);
    const float* src = (slot == 0 ? TS(h).accum.p : TS(h).accum_update.p) +
                       TS(h).goff[param_index(h, p)];
    HIP_CHECK(hipMemcpy(out, src, p->elems() * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int mv_set_opt_slot(mv_handle h, const char* tf_name, int32_t slot, const float* data,
                    int64_t elems) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(h->train, "mv_train_init has not been called");
    MV_REQUIRE(slot == 0 || slot == 1, "slot must be 0 or 1 (see multiverse_hip.h)");
    Param* p = find_param(h, tf_name);
    MV_REQUIRE(data && (size_t)elems == p->elems(), "size mismatch for %s", tf_name);
    float* dst = (slot == 0 ? TS(h).accum.p : TS(h).accum_update.p) +
                 TS(h).goff[param_index(h, p)];
    HIP_CHECK(hipMemcpy(dst, data, p->elems() * sizeof(float), hipMemcpyHostToDevice));
  });
}

int mv_set_compute_mode(mv_handle h, int32_t mode) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(mode >= 0 && mode <= 2, "compute mode %d (0 = fp32 MFMA, 1 = f16x3, 2 = bf16)",
               mode);
    MV_REQUIRE(!(mode == 2 && h->train && h->cfg.activation != 0),
               "compute mode 2 (bf16) on a TRAINING engine needs activation_func tanh: relu / "
               "lrelu models train in mode 1 (f16x3); see mv_config.activation");
    MV_REQUIRE(mode == 0 || h->cfg.convlstm_kernel == 3,
               "compute mode %d needs convlstm_kernel 3 (%d given: the matrix-pipe gate kernels "
               "are 3 x 3 stencils; mode 0 runs the generic fp32 loops)", mode,
               h->cfg.convlstm_kernel);
    if (mode != 0) {
      // operand-plane scratch per group slot: even slots class-sized (N*B rows),
      // odd slots regression-sized (N rows), largest enabled grid
      const mv_config& c = h->cfg;
      size_t K = 0;
      for (int s = 0; s < c.num_scales; ++s)
        if (h->sc[s].use) K = std::max(K, (size_t)h->sc[s].K);
      const size_t xc = (size_t)std::max(c.scene_conv_dim, c.emb_size);
      for (int i = 0; i < mv::kMaxGroup; ++i) {
        const size_t rows = (size_t)c.batch_size * ((i % 2 == 0) ? c.beam_size : 1);
        // [pad | plane 0 | pad | plane 1], pads zero (out-of-image taps read them)
        h->px16[i].alloc(2 * (rows * K * xc + mv::kPlaneSlack + mv::kPlanePad));
        h->ph16[i].alloc(2 * (rows * K * c.hidden_size + mv::kPlaneSlack + mv::kPlanePad));
        HIP_CHECK(hipMemset(h->px16[i].p, 0, h->px16[i].n * sizeof(_Float16)));
        if (c.activation != 0) h->xexp[i].alloc(65);    // never inside a graph capture
        if (mode == 1 && mv::wino_enabled() && mv::wino3_enabled() && c.activation == 0) {
          size_t vx = 0, vh = 0;
          for (int s = 0; s < c.num_scales; ++s) {
            if (!h->sc[s].use || h->sc[s].H < 3) continue;
            vx = std::max(vx, mv::wino3_v_elems((int)rows, h->sc[s].H, h->sc[s].W, (int)((xc + 15) / 16 * 16)));
            vh = std::max(vh, mv::wino3_v_elems((int)rows, h->sc[s].H, h->sc[s].W, c.hidden_size));
          }
          if (vx) h->pv3x[i].alloc(vx);
          if (vh) h->pv3h[i].alloc(vh);
        }
        HIP_CHECK(hipMemset(h->ph16[i].p, 0, h->ph16[i].n * sizeof(_Float16)));
      }
    }
    if (mode != 0 && h->planes.empty()) {
      for (int s = 0; s < h->cfg.num_scales; ++s) {
        ScaleState& S = h->sc[s];
        if (!S.use) continue;
        for (DevBuf<float>* b : {&S.cls_h[0], &S.cls_h[1], &S.reg_h[0], &S.reg_h[1],
                                 &S.cls_hg, &S.xbuf_cls, &S.xbuf_reg}) {
          if (!b->p) continue;
          h->plane_store.emplace_back(new DevBuf<_Float16>());
          DevBuf<_Float16>& pb = *h->plane_store.back();
          pb.alloc(2 * (b->n + mv::kPlaneSlack + mv::kPlanePad));
          HIP_CHECK(hipMemset(pb.p, 0, pb.n * sizeof(_Float16)));
          // p -> first element of plane 0; plane stride n + slack + pad puts a zero
          // pad in front of plane 1 as well (slack: the last partial 32-cell tile row)
          h->planes[b->p] = mv_engine::PlaneBuf{pb.p + mv::kPlanePad,
                                                b->n + mv::kPlaneSlack + mv::kPlanePad, false};
          if (b == &S.xbuf_cls || b == &S.xbuf_reg) h->xbufs.insert(b->p);
        }
      }
    }
    for (auto& kv : h->planes) kv.second.valid = false;
    if (h->compute_mode != mode) {
      h->drop_graphs();
      // the backward's weight packs are per mode (f16x3 planes / Winograd form / one bf16
      // plane): the next training step re-packs them from the current device weights
      h->train_packs_valid = false;
    }
    h->compute_mode = mode;
  });
}

int mv_set_graph_mode(mv_handle h, int32_t enabled) {
  if (!h) return 1;
  h->graph_mode = enabled != 0;
  return 0;
}

int mv_reset_kernel_stats(mv_handle h) {
  if (!h) return 1;
  return guarded(h, [&] {
    HIP_CHECK(hipStreamSynchronize(h->stream));
    drain_events(h);
    h->stats.clear();
  });
}

int mv_num_kernel_stats(mv_handle h) { return h ? (int)h->stats.size() : -1; }

int mv_kernel_stat_dense_flops(mv_handle h, int32_t i, double* flops_dense) {
  if (!h || i < 0 || i >= (int)h->stats.size() || !flops_dense) return 1;
  *flops_dense = h->stats[i].flops_dense;
  return 0;
}

int mv_kernel_stat_mfma_flops(mv_handle h, int32_t i, double* flops_mfma) {
  if (!h || i < 0 || i >= (int)h->stats.size() || !flops_mfma) return 1;
  *flops_mfma = h->stats[i].flops_mfma;
  return 0;
}

int mv_kernel_stat(mv_handle h, int32_t i, char* name_out, int32_t name_cap,
                   int64_t* launches, double* total_ms, double* flops, double* bytes) {
  if (!h || i < 0 || i >= (int)h->stats.size()) return 1;
  const KernelStat& s = h->stats[i];
  if (name_out && name_cap > 0) {
    strncpy(name_out, s.name.c_str(), name_cap - 1);
    name_out[name_cap - 1] = 0;
  }
  if (launches) *launches = s.launches;
  if (total_ms) *total_ms = s.total_ms;
  if (flops) *flops = s.flops;
  if (bytes) *bytes = s.bytes;
  return 0;
}

static int time_resident(mv_handle h, int32_t iters, float* ms_out, bool beam) {
  if (!h) return 1;
  return guarded(h, [&] {
    MV_REQUIRE(iters >= 1 && ms_out, "bad arguments");
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(a, h->stream));
    for (int i = 0; i < iters; ++i) run_forward(h, beam);
    HIP_CHECK(hipEventRecord(b, h->stream));
    HIP_CHECK(hipEventSynchronize(b));
    HIP_CHECK(hipEventElapsedTime(ms_out, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    drain_events(h);
  });
}

int mv_time_greedy_resident(mv_handle h, int32_t iters, float* ms_out) {
  return time_resident(h, iters, ms_out, false);
}
int mv_time_beam_resident(mv_handle h, int32_t iters, float* ms_out) {
  return time_resident(h, iters, ms_out, true);
}

// ------------------------------------------------------- single-kernel ops

int mv_op_convlstm_step(int device, const float* x, const float* c, const float* h,
                        const float* kernel, const float* biases, int32_t M,
                        int32_t H, int32_t W, int32_t Cx, int32_t C, float* c_out,
                        float* h_out) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(C % mv::kChBlock == 0 && C % mv::kBK == 0, "C %d must be a multiple of 32", C);
    MV_REQUIRE(mv::convlstm_cx_supported(Cx), "Cx %d unsupported (multiple of 32, or <= 3)", Cx);
    OpCtx ctx(device);
    const size_t cells = (size_t)M * H * W;
    DevBuf<float> dx, dc, dh, dw, db, dco, dho;
    ctx.up(dx, x, cells * Cx);
    ctx.up(db, biases, (size_t)4 * C);
    std::vector<float> packed(mv::convlstm_wpack_elems(Cx, C));
    mv::pack_convlstm_weights(kernel, Cx, C, packed.data());
    ctx.up(dw, packed.data(), packed.size());
    const bool zero = (c == nullptr && h == nullptr);
    if (!zero) {
      MV_REQUIRE(c && h, "c and h must both be given or both be NULL");
      ctx.up(dc, c, cells * C);
      ctx.up(dh, h, cells * C);
    }
    dco.alloc(cells * C); dho.alloc(cells * C);
    mv::ConvLstmArgs a{};
    a.x = dx.p; a.h = dh.p; a.c = dc.p; a.wpack = dw.p; a.bias = db.p;
    a.h_out = dho.p; a.c_out = dco.p;
    a.rows = M; a.H = H; a.W = W; a.Cx = Cx; a.C = C;
    mv::convlstm_finish_args(a, zero);
    mv::launch_convlstm_steps(&a, 1, ctx.stream);
    HIP_CHECK(hipGetLastError());
    ctx.down(c_out, dco, cells * C);
    ctx.down(h_out, dho, cells * C);
  });
}

namespace {
// planes (hi + lo) / 256 of an [M][C] tensor in the tiled operand layout -> fp32 [M][C]
__global__ void decode_planes_kernel(const _Float16* __restrict__ p0,
                                     const _Float16* __restrict__ p1, float* __restrict__ out,
                                     size_t M, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * (size_t)C) return;
  const size_t m = i / C;
  const int ch = (int)(i - m * C);
  const size_t o = mv::plane_index((long long)m, ch, C);
  out[i] = ((float)p0[o] + (float)p1[o]) * (1.0f / 256.0f);
}
}  // namespace

int mv_op_convlstm_step16(int device, int32_t variant, const float* x, const float* c,
                          const float* h, const float* kernel, const float* biases,
                          int32_t M, int32_t H, int32_t W, int32_t Cx, int32_t C,
                          float* c_out, float* h_out, float* h16_out) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(variant >= 1 && variant <= 3, "variant %d: 1 = direct f16x3, 2 = Winograd F(2,3), 3 = Winograd F(3,3)", variant);
    MV_REQUIRE(C % mv::kChBlock == 0 && C % mv::kBK == 0, "C %d must be a multiple of 32", C);
    MV_REQUIRE(mv::f16x3_cx_supported(Cx), "Cx %d unsupported (multiple of 16, or <= 3)", Cx);
    MV_REQUIRE(H * W >= 32, "grids of at least 32 cells");
    OpCtx ctx(device);
    const size_t cells = (size_t)M * H * W;
    const bool small = Cx > 0 && 9 * Cx <= mv::kBK;
    const int Cx16 = small ? 0 : Cx;
    const int Cin = Cx + C, N4 = 4 * C;
    DevBuf<float> dx, dc, dh, dw, db, dco, dho, dk, dwx, dplanes;
    DevBuf<_Float16> px, ph, pho, wp;
    ctx.up(dx, x, cells * Cx);
    ctx.up(db, biases, (size_t)4 * C);
    ctx.up(dk, kernel, (size_t)9 * Cin * N4);
    std::vector<float> packed(mv::convlstm_wpack_elems(Cx, C));
    mv::pack_convlstm_weights(kernel, Cx, C, packed.data());
    ctx.up(dw, packed.data(), packed.size());
    const bool zero = (c == nullptr && h == nullptr);
    if (!zero) {
      MV_REQUIRE(c && h, "c and h must both be given or both be NULL");
      ctx.up(dc, c, cells * C);
      ctx.up(dh, h, cells * C);
    }
    dco.alloc(cells * C); dho.alloc(cells * C);
    // operand planes: [pad | plane 0 | slack][pad | plane 1 | slack], zero-filled
    auto make_planes = [&](DevBuf<_Float16>& buf, const float* src, int Cc, size_t* stride) {
      const size_t pst = cells * Cc + mv::kPlaneSlack + mv::kPlanePad;
      buf.alloc(2 * pst + mv::kPlanePad);
      HIP_CHECK(hipMemsetAsync(buf.p, 0, (2 * pst + mv::kPlanePad) * sizeof(_Float16), ctx.stream));
      _Float16* p0 = buf.p + mv::kPlanePad;
      if (src)
        hipLaunchKernelGGL(mv::split_planes_kernel, dim3(mv::split_planes_blocks(cells, Cc)),
                           dim3(256), 0, ctx.stream, src, p0, p0 + pst, (int)cells, Cc);
      *stride = pst;
      return p0;
    };
    mv::ConvLstm16Args q{};
    mv::ConvLstmArgs& a = q.f;
    a.x = dx.p; a.h = dh.p; a.c = dc.p; a.wpack = dw.p; a.bias = db.p;
    a.h_out = dho.p; a.c_out = dco.p;
    a.rows = M; a.H = H; a.W = W; a.Cx = Cx; a.C = C;
    mv::convlstm_finish_args(a, zero);
    size_t xst = 0, hst = 0, ost = 0;
    if (Cx16 > 0) { q.x16 = make_planes(px, dx.p, Cx16, &xst); q.x_plane_stride = (int64_t)xst; }
    if (!zero) { q.h16 = make_planes(ph, dh.p, C, &hst); q.h_plane_stride = (int64_t)hst; }
    q.h16_out = make_planes(pho, nullptr, C, &ost);
    q.h16_out_stride = (int64_t)ost;
    q.n_xk = small ? 0 : mv::f16x3_xksteps(Cx);
    q.n_hk = zero ? 0 : 9 * (C / 16);
    q.w_ksteps = small ? 9 * (C / 16) : mv::f16x3_xksteps(Cx) + 9 * (C / 16);
    if (variant == 1) {
      std::vector<_Float16> p16(mv::f16x3_wpack_elems(Cx16, C));
      if (small) {
        std::vector<float> wh((size_t)9 * C * N4);
        for (int t = 0; t < 9; ++t)
          memcpy(&wh[(size_t)t * C * N4], &kernel[((size_t)t * Cin + Cx) * N4],
                 (size_t)C * N4 * sizeof(float));
        mv::pack_f16x3_weights(wh.data(), 0, C, p16.data());
        const int nch = mv::convlstm_xchunks(Cx) + 9 * (C / mv::kBK);
        std::vector<float> wx((size_t)(C / mv::kChBlock) * mv::kBN * mv::kBK);
        for (int cb = 0; cb < C / mv::kChBlock; ++cb)
          for (int i = 0; i < mv::kBN * mv::kBK; ++i)
            wx[(size_t)cb * mv::kBN * mv::kBK + i] =
                packed[((size_t)cb * nch + 0) * mv::kBN * mv::kBK + i] * 65536.0f;
        ctx.up(dwx, wx.data(), wx.size());
        q.wx32 = dwx.p;
      } else {
        mv::pack_f16x3_weights(kernel, Cx, C, p16.data());
      }
      ctx.up(wp, p16.data(), p16.size());
      q.wp16 = wp.p;
      mv::launch_convlstm16_steps(&q, 1, ctx.stream);
    } else if (variant == 3) {
      MV_REQUIRE(mv::wino3_geometry_ok(a, q), "Winograd F(3,3) form: H %d >= 3", H);
      MV_REQUIRE(mv::wino3_halo_addressable(a), "Winograd F(3,3) form, halo tiling (W %d does "
                 "not divide 32): an operand of 2 GiB or more is not addressable", W);
      const size_t halves = mv::wino3_wpack_elems(Cx16, C, mv::kW3Nrb);
      wp.alloc(halves);
      hipLaunchKernelGGL(mv::pack_wino3_kernel, dim3(cdiv(halves / 2, 256)), dim3(256), 0,
                         ctx.stream, dk.p, wp.p, Cx, Cx16, C, mv::kW3Nrb, halves / 2);
      mv::ConvLstmWinoArgs wq{};
      wq.b = q; wq.wpw = wp.p; wq.w_hwio = dk.p; wq.n_xc = Cx16 / 16;
      // the pre-transformed operands, as the engine hands them over
      DevBuf<_Float16> v3x, v3h;
      {
        std::vector<mv::Wn3TransformItem> tr;
        if (!zero) {
          v3h.alloc(mv::wino3_v_elems(M, H, W, C));
          tr.push_back(mv::Wn3TransformItem{q.h16, q.h_plane_stride, v3h.p, nullptr, M, H, W, C});
          wq.v3h = v3h.p;
        }
        if (Cx16 > 0) {
          v3x.alloc(mv::wino3_v_elems(M, H, W, Cx16));
          tr.push_back(mv::Wn3TransformItem{q.x16, q.x_plane_stride, v3x.p, nullptr, M, H, W, Cx16});
          wq.v3x = v3x.p;
        }
        mv::launch_wino3_transforms(tr.data(), (int)tr.size(), ctx.stream);
      }
      mv::launch_convlstm_wino3_steps(&wq, 1, ctx.stream);
      HIP_CHECK(hipStreamSynchronize(ctx.stream));    // v3x / v3h die with this scope
    } else {
      MV_REQUIRE(mv::wino_geometry_ok(a), "Winograd form: W %d must divide 32, H >= 2", W);
      const size_t halves = mv::wino_wpack_elems(Cx16, C);
      wp.alloc(halves);
      hipLaunchKernelGGL(mv::pack_wino_kernel, dim3(cdiv(halves / 2, 256)), dim3(256), 0,
                         ctx.stream, dk.p, wp.p, Cx, Cx16, C, halves / 2);
      mv::ConvLstmWinoArgs wq{};
      wq.b = q; wq.wpw = wp.p; wq.w_hwio = dk.p; wq.n_xc = Cx16 / 16;
      mv::launch_convlstm_wino_steps(&wq, 1, ctx.stream);
    }
    HIP_CHECK(hipGetLastError());
    if (h16_out) {
      dplanes.alloc(cells * C);
      hipLaunchKernelGGL(decode_planes_kernel, dim3(cdiv(cells * C, 256)), dim3(256), 0,
                         ctx.stream, q.h16_out, q.h16_out + ost, dplanes.p, cells, C);
      HIP_CHECK(hipGetLastError());
      ctx.down(h16_out, dplanes, cells * C);
    }
    ctx.down(c_out, dco, cells * C);
    ctx.down(h_out, dho, cells * C);
  });
}

int mv_op_gnn(int device, const float* h, const float* scene_mean, int32_t M,
              int32_t H, int32_t W, int32_t C, int32_t D, float* out) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(C % 64 == 0 && C <= 512 && D >= 0 && D <= 128,
               "gnn: C a multiple of 64 up to 512, D <= 128");
    OpCtx ctx(device);
    const size_t cells = (size_t)M * H * W;
    DevBuf<float> dh, ds, dout;
    ctx.up(dh, h, cells * C);
    ctx.up(ds, scene_mean, cells * D);
    dout.alloc(cells * C);
    int ver = gnn_version();              // read per call: the kernel test runs every version
    if (ver >= 3 && !((D == 0 || D == 64) && cells * C * 4 < ((size_t)1 << 32))) ver = 2;
    if (ver >= 2 && W <= 32 && C == 256) {
      int ng = 0;
      const unsigned nb = ver >= 3 ? mv::gnn_v3_blocks(cells, &ng) : mv::gnn_v2_blocks(cells, &ng);
      mv::GnnGroup grp{};
      grp.p[0] = mv::GnnProblem{dh.p, ds.p, nullptr, dout.p, nullptr, 0, M, H, W, 1, ng, nullptr};
      grp.nblocks0 = nb;
      if (ver >= 3)
        hipLaunchKernelGGL(mv::gnn_attend_v3_kernel, dim3(nb), dim3(mv::kGnn3Threads), 0, ctx.stream,
                           grp, C, D);
      else
        hipLaunchKernelGGL(mv::gnn_attend_v2_kernel, dim3(nb), dim3(mv::kGnnThreads), 0, ctx.stream,
                           grp, C, D);
    } else {
      if (C <= 256)
        hipLaunchKernelGGL(mv::gnn_attend_kernel<1>, dim3(cdiv(cells, 4)), dim3(256), 0,
                           ctx.stream, dh.p, ds.p, (const int32_t*)nullptr, dout.p, M, H,
                           W, C, D, 1, (_Float16*)nullptr, (size_t)0);
      else
        hipLaunchKernelGGL(mv::gnn_attend_kernel<2>, dim3(cdiv(cells, 4)), dim3(256), 0,
                           ctx.stream, dh.p, ds.p, (const int32_t*)nullptr, dout.p, M, H,
                           W, C, D, 1, (_Float16*)nullptr, (size_t)0);
    }
    HIP_CHECK(hipGetLastError());
    ctx.down(out, dout, cells * C);
  });
}

int mv_op_hidden2grid(int device, const float* h, const float* w, int32_t M,
                      int32_t H, int32_t W, int32_t C, int32_t P, float* out) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(C % 4 == 0 && (P == 1 || P == 2), "hidden2grid: C %% 4 == 0, P in {1,2}");
    OpCtx ctx(device);
    const size_t cells = (size_t)M * H * W;
    DevBuf<float> dh, dw, dout;
    ctx.up(dh, h, cells * C);
    ctx.up(dw, w, (size_t)9 * C * P);
    dout.alloc(cells * P);
    if (P == 1)
      hipLaunchKernelGGL(mv::hidden2grid_kernel<1>, dim3(cdiv(cells, 4)), dim3(256), 0,
                         ctx.stream, dh.p, dw.p, dout.p, (size_t)H * W, M, H, W, C);
    else
      hipLaunchKernelGGL(mv::hidden2grid_kernel<2>, dim3(cdiv(cells, 4)), dim3(256), 0,
                         ctx.stream, dh.p, dw.p, dout.p, (size_t)H * W * 2, M, H, W, C);
    HIP_CHECK(hipGetLastError());
    ctx.down(out, dout, cells * P);
  });
}

int mv_op_beam_step(int device, const float* logits, const float* prev_logprob,
                    int32_t N, int32_t B, int32_t K, int32_t time, int32_t diverse,
                    float gamma, int32_t fix_num_timestep, float* new_logprob,
                    int32_t* ids, int32_t* parents) {
  return guarded(nullptr, [&] {
    OpCtx ctx(device);
    const size_t lds = ((size_t)2 * B * K + 512) * sizeof(float);
    ensure_beam_step_lds(device, lds);
    DevBuf<float> dl, dp, dn;
    DevBuf<int32_t> di, dpa;
    ctx.up(dl, logits, (size_t)N * B * K);
    ctx.up(dp, prev_logprob, (size_t)N * B);
    dn.alloc((size_t)N * B); di.alloc((size_t)N * B); dpa.alloc((size_t)N * B);
    DevBuf<float> dc;
    dc.alloc((size_t)N * B * K);
    launch_beam_step(ctx.stream, dl.p, dp.p, dc.p, N, B, K, time, diverse, logf(gamma),
                     fix_num_timestep, dn.p, di.p, dpa.p, (int32_t*)nullptr, B);
    ctx.down(new_logprob, dn, (size_t)N * B);
    ctx.down(ids, di, (size_t)N * B);
    ctx.down(parents, dpa, (size_t)N * B);
  });
}

int mv_op_convlstm_bwd(int device, const float* x, const float* c, const float* h,
                       const float* kernel, const float* biases, const float* dh_new,
                       const float* dc_new, int32_t M, int32_t H, int32_t W, int32_t Cx,
                       int32_t C, float* dx, float* dh, float* dc, float* dkernel,
                       float* dbiases) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(C % 128 == 0 && C <= 512, "convlstm_bwd: C 128, 256, 384 or 512");
    MV_REQUIRE(mv::convlstm_cx_supported(Cx), "Cx %d unsupported", Cx);
    OpCtx ctx(device);
    const size_t cells = (size_t)M * H * W;
    DevBuf<float> dx_, dc_, dh_, dw, db, dco, dho, dg, ddh, ddc, dwd, dxo, dho2, part, dW,
        dB, tmp;
    dx_.alloc(cells * Cx ? cells * Cx : 1, mv::kWgradPad);
    if (cells * Cx)
      HIP_CHECK(hipMemcpy(dx_.p, x, cells * Cx * sizeof(float), hipMemcpyHostToDevice));
    ctx.up(db, biases, (size_t)4 * C);
    std::vector<float> packed(mv::convlstm_wpack_elems(Cx, C));
    mv::pack_convlstm_weights(kernel, Cx, C, packed.data());
    ctx.up(dw, packed.data(), packed.size());
    std::vector<float> packedT(mv::convlstm_dgrad_wpack_elems(Cx, C));
    mv::pack_convlstm_dgrad_weights(kernel, Cx, C, packedT.data());
    ctx.up(dwd, packedT.data(), packedT.size());
    const bool zero = (c == nullptr && h == nullptr);
    dc_.alloc(cells * C); dh_.alloc(cells * C, mv::kWgradPad);
    if (!zero) {
      MV_REQUIRE(c && h, "c and h must both be given or both be NULL");
      HIP_CHECK(hipMemcpy(dc_.p, c, cells * C * sizeof(float), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(dh_.p, h, cells * C * sizeof(float), hipMemcpyHostToDevice));
    } else {
      HIP_CHECK(hipMemset(dc_.p, 0, cells * C * sizeof(float)));
      HIP_CHECK(hipMemset(dh_.p, 0, cells * C * sizeof(float)));
    }
    dco.alloc(cells * C); dho.alloc(cells * C); dg.alloc(cells * 4 * C, mv::kWgradPad);
    ctx.up(ddh, dh_new, cells * C);
    ctx.up(ddc, dc_new, cells * C);
    // forward with saved gate activations
    mv::ConvLstmArgs a{};
    a.x = dx_.p; a.h = dh_.p; a.c = dc_.p; a.wpack = dw.p; a.bias = db.p;
    a.h_out = dho.p; a.c_out = dco.p; a.gates_out = dg.p;
    a.rows = M; a.H = H; a.W = W; a.Cx = Cx; a.C = C;
    mv::convlstm_finish_args(a, zero);
    mv::launch_convlstm_steps(&a, 1, ctx.stream);
    // pointwise backward: gates -> G in place, ddc -> d c
    const size_t total = cells * C;
    hipLaunchKernelGGL(mv::lstm_gate_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                       ctx.stream, dg.p, dc_.p, dco.p, ddh.p, ddc.p, total, C);
    // dgrad
    dxo.alloc(cells * (Cx ? Cx : 1)); dho2.alloc(cells * C);
    mv::ConvLstmArgs d{};
    mv::convlstm_dgrad_args(d, dg.p, dwd.p, dho2.p, dxo.p, M, H, W, Cx, C, true, Cx > 0);
    mv::launch_convlstm_dgrads(&d, 1, ctx.stream);
    // wgrad (device-side packs are checked against the host packs on the way)
    mv::WgradArgs wa{};
    wa.x = Cx ? dx_.p : nullptr; wa.h = dh_.p; wa.g = dg.p;
    wa.R = M; wa.H = H; wa.W = W; wa.Cx = Cx; wa.C = C;
    mv::wgrad_plan(wa, 3072);
    part.alloc(mv::wgrad_partial_elems(wa));
    wa.partial = part.p;
    mv::launch_convlstm_wgrad(wa, ctx.stream);
    const size_t ncols = (size_t)9 * (Cx + C) * 4 * C;
    dW.alloc(ncols);
    hipLaunchKernelGGL(mv::colsum_kernel, dim3(1, cdiv(ncols, 256)), dim3(256), 0,
                       ctx.stream, part.p, dW.p, (size_t)wa.nsplit, ncols,
                       (size_t)wa.nsplit);
    dB.alloc((size_t)4 * C);
    hipLaunchKernelGGL(mv::colsum_kernel, dim3(1, cdiv((size_t)4 * C, 256)), dim3(256), 0,
                       ctx.stream, dg.p, dB.p, cells, (size_t)4 * C, cells);
    // device packs == host packs (the training step repacks on the device)
    {
      DevBuf<float> wsrc, p1, p2;
      ctx.up(wsrc, kernel, (size_t)9 * (Cx + C) * 4 * C);
      const int nx = mv::convlstm_xchunks(Cx), nch = nx + 9 * (C / mv::kBK);
      p1.alloc(packed.size()); p2.alloc(packedT.size());
      hipLaunchKernelGGL(mv::pack_fwd_kernel, dim3(cdiv(packed.size(), 256)), dim3(256), 0,
                         ctx.stream, wsrc.p, p1.p, Cx, C, nx, nch,
                         (Cx > 0 && 9 * Cx <= mv::kBK) ? 1 : 0, packed.size());
      hipLaunchKernelGGL(mv::pack_dgrad_kernel, dim3(cdiv(packedT.size(), 256)), dim3(256),
                         0, ctx.stream, wsrc.p, p2.p, Cx, C, 9 * (4 * C / mv::kBK),
                         packedT.size());
      std::vector<float> c1(packed.size()), c2(packedT.size());
      ctx.down(c1.data(), p1, c1.size());
      ctx.down(c2.data(), p2, c2.size());
      MV_REQUIRE(memcmp(c1.data(), packed.data(), c1.size() * 4) == 0,
                 "device forward weight pack differs from the host pack");
      MV_REQUIRE(memcmp(c2.data(), packedT.data(), c2.size() * 4) == 0,
                 "device dgrad weight pack differs from the host pack");
    }
    HIP_CHECK(hipGetLastError());
    if (dx && Cx) ctx.down(dx, dxo, cells * Cx);
    ctx.down(dh, dho2, cells * C);
    ctx.down(dc, ddc, cells * C);
    ctx.down(dkernel, dW, ncols);
    ctx.down(dbiases, dB, (size_t)4 * C);
  });
}

int mv_op_gnn_bwd(int device, const float* h, const float* scene_mean, const float* g,
                  int32_t M, int32_t H, int32_t W, int32_t C, int32_t D, float* dh,
                  float* dscene_mean) {
  return guarded(nullptr, [&] {
    MV_REQUIRE(C % 64 == 0 && C <= 512 && D >= 0 && D <= 128,
               "gnn_bwd: C a multiple of 64 up to 512, D <= 128");
    OpCtx ctx(device);
    const size_t cells = (size_t)M * H * W;
    DevBuf<float> dh_, ds_, dg_, a, de, n, odh, ods;
    ctx.up(dh_, h, cells * C);
    ctx.up(ds_, scene_mean, cells * D);
    ctx.up(dg_, g, cells * C);
    a.alloc(cells * 9); de.alloc(cells * 9); n.alloc(cells);
    odh.alloc(cells * C); ods.alloc(cells * (D ? D : 1));
    if (C <= 256) {
      hipLaunchKernelGGL(mv::gnn_bwd_a_kernel<1>, dim3(cdiv(cells, 4)), dim3(256), 0, ctx.stream,
                         dh_.p, ds_.p, dg_.p, a.p, de.p, n.p, M, H, W, C, D);
      hipLaunchKernelGGL(mv::gnn_bwd_b_kernel<1>, dim3(cdiv(cells, 4)), dim3(256), 0, ctx.stream,
                         dh_.p, ds_.p, dg_.p, a.p, de.p, n.p, odh.p, ods.p, M, H, W, C, D, 0);
    } else {
      hipLaunchKernelGGL(mv::gnn_bwd_a_kernel<2>, dim3(cdiv(cells, 4)), dim3(256), 0, ctx.stream,
                         dh_.p, ds_.p, dg_.p, a.p, de.p, n.p, M, H, W, C, D);
      hipLaunchKernelGGL(mv::gnn_bwd_b_kernel<2>, dim3(cdiv(cells, 4)), dim3(256), 0, ctx.stream,
                         dh_.p, ds_.p, dg_.p, a.p, de.p, n.p, odh.p, ods.p, M, H, W, C, D, 0);
    }
    HIP_CHECK(hipGetLastError());
    ctx.down(dh, odh, cells * C);
    if (dscene_mean && D) ctx.down(dscene_mean, ods, cells * D);
  });
}

// Debug probe (not part of the public header): XCC id of every workgroup of a
// 1-D launch of `nblocks` x 256 threads -- checks the "linear id % 8 -> XCD"
// dispatch pattern the XCD-aware block maps rely on for speed.
__global__ void xcc_probe_kernel(int32_t* out) {
  if (threadIdx.x == 0) {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    out[blockIdx.x] = (int32_t)(v & 0xf);
  }
}

int mv_debug_xcc_map(int device, int32_t nblocks, int32_t* out) {
  return guarded(nullptr, [&] {
    OpCtx ctx(device);
    DevBuf<int32_t> d;
    d.alloc(nblocks);
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(nblocks), dim3(256), 0, ctx.stream, d.p);
    HIP_CHECK(hipGetLastError());
    ctx.down(out, d, (size_t)nblocks);
  });
}

}  // extern "C"
