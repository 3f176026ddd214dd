// engine state: device buffers, parameters, per-scale state, mv_engine -- part of the ONE translation unit engine.hip (included from there, in order;
// not a stand-alone header).
#pragma once

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
  DevBuf<_Float16> wpbt;    // bf16 mode on the row-triple tile (convlstm_wino3.h BF16D)
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
  // relu / lrelu models on the fp16 / bf16 pipe: the x operands of the gate convolutions are
  // unbounded, so their planes carry a per-tensor exponent (max |x| as float bits [64] |
  // exponent [1]) instead of the fixed 2^8; the producers do not emit planes for these buffers.
  // In bf16 mode (round 6) the x k-steps of such models run on the LEADING fp16 plane of that
  // split (11 significand bits under the exponent, fp16 weights for the x rows, fp16 MFMA)
  // instead of one bf16 plane (8 bits): a pixel-offset embedding of hundreds rounded to bf16
  // cost the regression decoder's kernel gradient its direction (cosine 0.96)
  DevBuf<int32_t> xexp[mv::kMaxGroup];
  std::set<const float*> xbufs;
  bool dyn_x() const { return compute_mode != 0 && cfg.activation != 0; }
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
