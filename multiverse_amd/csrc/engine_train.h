// Training step of the engine (included by engine.hip after the inference
// path): the role of `Trainer.step` = sess.run([loss, train_op, wd_loss,
// pred_grid_loss]) at reference code/pred_models.py:1719-1742 over
//   Model.build_forward (:123-308, is_train, --train_w_onehot wiring),
//   Model.build_loss (:961-1040),
//   tf.gradients + clip_by_value + AdadeltaOptimizer.apply_gradients (:1667-1717).
//
// Memory plan (288 GB HBM: keep everything, recompute nothing).  Per scale and
// branch (class / regression) the h and c states of all T_o + T_p + 1 time
// slots live in ONE time-major buffer [slot][N][K][C], so that the h operand of
// every step of a cell is a contiguous [T*N] image batch for the batched wgrad.
// Per cell the saved gate activations [T][N][K][4C] are overwritten IN PLACE by
// the gate gradients G during the backward sweep.
#pragma once

namespace {

struct TrainChain {               // one ConvLSTM cell over its T steps
  ConvCell* cell = nullptr;
  int T = 0, Cx = 0;
  DevBuf<float> xs;               // [T][N][K][Cx]   x operand of every step
  DevBuf<float> dxs;              // [T][N][K][Cx]   d x (then d pre-activation)
  DevBuf<float> gates;            // [T][N][K][4C]   activations -> G in place
  DevBuf<float> wdpack;           // dgrad weight pack
  DevBuf<_Float16> wd16;          // f16x3 compute mode: the same as two fp16 planes
  DevBuf<_Float16> wdw;           // ... and in Winograd F(2,3) form (convlstm_wino.h, dgrad)
  DevBuf<_Float16> wdb;           // bf16 compute mode: one bf16 plane (pack_bf16_dgrad_kernel)
};

// bf16 compute mode (BASELINE.json configs[4]: reduced-precision operands, fp32 accumulate):
// the backward runs on one plane per operand as well -- dgrad on bf16 planes of G and of the
// kernel, wgrad on the leading fp16 plane of each operand -- one MFMA per product instead of
// the f16x3 split's three.  MV_BF16_BWD=0 keeps the backward on the f16x3 split (rounds 2-3).
// Models with unbounded activations (--activation_func relu / lrelu): one 8-bit-mantissa plane of
// a pixel-offset embedding of hundreds moved the regression decoder's gates enough that its
// kernel gradient came out at cosine 0.96 against the fp32 oracle whichever backward ran (tanh
// models: 0.99997).  Since round 6 the x k-steps of such models run as an f16x3 split of the x
// part alone in the FORWARD (convlstm_f16x3.h xpasses; engine_state.h dyn_x): cosine 0.99995
// (tests/test_gpu_bf16.py::test_bf16_with_unbounded_activations).
static bool bf16_bwd_enabled(const mv_engine*) {
  static const bool on = !(getenv("MV_BF16_BWD") && atoi(getenv("MV_BF16_BWD")) == 0);
  return on;
}

struct TrainScale {
  DevBuf<float> hs[2], cs[2];     // branch 0 = class, 1 = regression: [To+Tp+1][N][K][C]
  TrainChain enc[2], dec[2];
  DevBuf<float> hg;               // [Tp][N][K][C]  h + GNN(h), class decoder h operand
  DevBuf<float> logits;           // [Tp][N][K]     time-major class logits
  DevBuf<float> regio;            // [Tp+1][N][K][2] slot 0 = obs_reg[:, -1]; slot t+1 = out_reg[t]
  DevBuf<int32_t> ids;            // [Tp][N] class-decoder input id per step (slot 0 unused)
  DevBuf<float> onehot;           // [Tp][N][K]
  DevBuf<float> dlogits, dreg;    // [Tp][N][K], [Tp][N][K][2]
  DevBuf<float> loss_row, loss_elem;
  DevBuf<float> dh_a[2], dh_b[2], dc[2];   // [N][K][C]
  DevBuf<float> gnn_a, gnn_de, gnn_n;      // [N*K*9] x2, [N*K]
  DevBuf<float> dsmean;           // [N][K][D]
  DevBuf<int32_t> pred_labels;    // [N][Tp]
  DevBuf<float> pred_reg;         // [N][Tp][K][2]
  // switches beyond the published run (soft labels, teacher forcing, masked regression)
  DevBuf<float> gt_cls;           // [Tp][N][K] one-hot / soft ground-truth class maps
  DevBuf<float> reg_in;           // [Tp][N][K][2] teacher forcing: regression-decoder inputs
  DevBuf<int32_t> fg_count;       // [1] cells with gt_cls > 0
  // SimAug label mixup (multi-view experiment 3)
  DevBuf<int32_t> obs_labels2, pred_labels2;   // [N][To], [N][Tp]: the selected extra view
  DevBuf<float> mix_cls;          // [Tp][N][K] mixed-up class targets
};

struct TrainState {
  mv_train_config tc{};
  int64_t global_step = 0;
  size_t total_elems = 0;
  std::vector<size_t> goff;       // per param: offset into the flat buffers
  DevBuf<float> grad, accum, accum_update;
  TrainScale sc[MV_MAX_SCALES];
  DevBuf<float> dys[MV_MAX_SCALES], dpre_sc[MV_MAX_SCALES];   // scene stack backward
  DevBuf<float> single_w;         // --use_single_decoder: one scale's d decode_reg W
  bool mix_on = false, mix_sw = false;   // label mixup in force / with per-sample weights
  float mix_w = 1.f;              // beta weight of the ORIGINAL labels
  DevBuf<float> sample_w;         // [N] focal weights (double_weighting)
  DevBuf<float> dscene;           // [U, SH, SW, SC] d loss / d scene_feat (SimAug attacks)
  DevBuf<float> scene_clean;      // the clean features an attack perturbs
  bool want_dscene = false, have_dscene = false;
  DevBuf<float> partial;          // split-K partials / reduction scratch
  DevBuf<float> scratch;          // second-stage scratch
  DevBuf<float> losses;           // [2*MV_MAX_SCALES + 1 + nW] device scalars
  // f16x3 dgrad: planes of the gate gradients of the current step per group slot,
  // max |G| bits and scale exponents per (launch slot, step)
  DevBuf<_Float16> g16[mv::kMaxGroup];
  bool g16_fused[mv::kMaxGroup] = {};   // bf16 backward: lstm_gate_bwd wrote the plane itself
  DevBuf<float> dpart0[mv::kMaxGroup], dpart1[mv::kMaxGroup];   // dgrad split-K partials
  DevBuf<int32_t> gmax, gexp;
  // f16x3 wgrad: cell-contiguous operand planes of one chain at a time
  DevBuf<_Float16> gt16, at16[3], xt16[3];
  DevBuf<float> bias_part;        // [Mrow / 64][4C] per-block column sums of G
  DevBuf<int32_t> chain_exp;
  size_t mrow_max = 0;
  bool wgrad16_attr = false;
  bool have_grads = false;
  bool targets_ready = false;
  mv_losses last{};
  uint32_t dropout_seed = 0;      // keep_prob < 1: seed of the next step's masks
  float beta1_power = 0.9f, beta2_power = 0.999f;   // Adam (TF non-slot variables)
  int slots_for = -1;             // optimizer the slot buffers were initialised for
  bool needs_gt() const {
    return tc.use_soft_grid_class || tc.mask_grid_regression || tc.class_feedback == 2;
  }
};

}  // namespace

struct mv_train_holder { TrainState st; };

namespace {

inline TrainState& TS(mv_engine* e) { return e->train->st; }

size_t param_index(mv_engine* e, const Param* p) {
  for (size_t i = 0; i < e->params.size(); ++i)
    if (e->params[i].get() == p) return i;
  throw HipError{"internal: unknown parameter"};
}
float* grad_of(mv_engine* e, const Param* p) {
  return TS(e).grad.p + TS(e).goff[param_index(e, p)];
}

void train_alloc(mv_engine* e) {
  TrainState& t = TS(e);
  const mv_config& c = e->cfg;
  const size_t N = c.batch_size, To = c.obs_len, Tp = c.max_pred_len, C = c.hidden_size,
               D = c.scene_conv_dim, E = c.emb_size;
  t.goff.clear();
  size_t off = 0;
  for (auto& p : e->params) {
    t.goff.push_back(off);
    off += (p->elems() + 63) & ~(size_t)63;
  }
  t.total_elems = off;
  t.grad.alloc(off); t.accum.alloc(off); t.accum_update.alloc(off);
  HIP_CHECK(hipMemset(t.grad.p, 0, off * sizeof(float)));
  HIP_CHECK(hipMemset(t.accum.p, 0, off * sizeof(float)));
  HIP_CHECK(hipMemset(t.accum_update.p, 0, off * sizeof(float)));
  size_t max_partial = 1 << 20;
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    TrainScale& R = t.sc[s];
    const size_t K = S.K, NK = N * K, slots = To + Tp + 1;
    for (int b = 0; b < 2; ++b) {
      R.hs[b].alloc(slots * NK * C, mv::kWgradPad); R.cs[b].alloc(slots * NK * C);
      HIP_CHECK(hipMemset(R.hs[b].p, 0, NK * C * sizeof(float)));   // slot 0: zero state
      HIP_CHECK(hipMemset(R.cs[b].p, 0, NK * C * sizeof(float)));
      R.dh_a[b].alloc(NK * C); R.dh_b[b].alloc(NK * C); R.dc[b].alloc(NK * C);
    }
    auto chain = [&](TrainChain& ch, ConvCell* cell, size_t T, bool need_dx) {
      ch.cell = cell; ch.T = (int)T; ch.Cx = cell->Cx;
      ch.xs.alloc(T * NK * cell->Cx, mv::kWgradPad);
      if (need_dx) ch.dxs.alloc(T * NK * cell->Cx);
      ch.gates.alloc(T * NK * 4 * C, mv::kWgradPad);
      ch.wdpack.alloc(mv::convlstm_dgrad_wpack_elems(cell->Cx, (int)C));
      mv::WgradArgs wa{};
      wa.R = (int)(T * N); wa.H = S.H; wa.W = S.W; wa.Cx = cell->Cx; wa.C = (int)C;
      mv::wgrad_plan(wa, 3072);
      // (the row-triple form of the f16x3 wgrad keeps 15 partial taps per split instead of 9)
      max_partial = std::max(max_partial, (mv::wgrad_partial_elems(wa) * 15 + 8) / 9);
    };
    chain(R.enc[0], &S.enc_cls, To, true);
    chain(R.dec[0], &S.dec_cls, Tp, true);
    if (!c.use_single_decoder) {     // single decoder: no regression chains at all
      chain(R.enc[1], &S.enc_reg, To, false);
      chain(R.dec[1], &S.dec_reg, Tp, true);
    }
    if (c.use_gnn) {
      R.hg.alloc(Tp * NK * C, mv::kWgradPad);
      R.gnn_a.alloc(NK * 9); R.gnn_de.alloc(NK * 9); R.gnn_n.alloc(NK);
      R.dsmean.alloc(NK * D);
    }
    R.logits.alloc(Tp * NK); R.regio.alloc((Tp + 1) * NK * 2);
    R.ids.alloc(Tp * N); R.onehot.alloc(Tp * NK);
    R.dlogits.alloc(Tp * NK); R.dreg.alloc(Tp * NK * 2);
    R.loss_row.alloc(Tp * N); R.loss_elem.alloc(Tp * NK * 2);
    R.pred_labels.alloc(N * Tp); R.pred_reg.alloc(N * Tp * K * 2);
    R.gt_cls.alloc(Tp * NK); R.reg_in.alloc(Tp * NK * 2); R.fg_count.alloc(1);
    R.obs_labels2.alloc(N * To); R.pred_labels2.alloc(N * Tp); R.mix_cls.alloc(Tp * NK);
    // small-conv wgrad partials: blocks x 9 x Ci*Co (<= 512)
    max_partial = std::max(max_partial, (size_t)4096 * 9 * 512);
    max_partial = std::max(max_partial, (size_t)64 * c.scene_conv_kernel * c.scene_conv_kernel * std::max<size_t>(D, c.scene_class) * D);
    // bias column sums: slabs x 4C
    max_partial = std::max(max_partial, (size_t)1088 * 4 * C);   // run_colsum: 1024 + 32 fold rows
    (void)E;
  }
  for (int i = 0; i < c.num_scales; ++i) {
    const size_t n = (size_t)N * To * e->conv_h[i] * e->conv_w[i] * D;
    t.dys[i].alloc(n); t.dpre_sc[i].alloc(n);
  }
  {
    size_t K = 0;
    for (int s = 0; s < c.num_scales; ++s)
      if (e->sc[s].use) K = std::max(K, (size_t)e->sc[s].K);
    for (int i = 0; i < mv::kMaxGroup; ++i) {
      t.g16[i].alloc(2 * (N * K * 4 * C + mv::kPlaneSlack + mv::kPlanePad));
      HIP_CHECK(hipMemset(t.g16[i].p, 0, t.g16[i].n * sizeof(_Float16)));
    }
    t.gmax.alloc(mv::kMaxGroup * 64 * 64);   // [slot][64 spread addresses]
    t.chain_exp.alloc(256);   // [0] G exp, [1] h exp (8), [2] x exp, [64..127] max|x| bits
    { const int32_t eight = 8;
      HIP_CHECK(hipMemcpy(t.chain_exp.p + 1, &eight, sizeof(eight), hipMemcpyHostToDevice)); }
    t.mrow_max = ((size_t)std::max(To, Tp) * N * K + 63) / 64 * 64;
    t.gexp.alloc(mv::kMaxGroup * 64);
  }
  t.partial.alloc(max_partial);
  t.scratch.alloc((size_t)1 << 20);
  t.single_w.alloc((size_t)9 * C * 2);
  t.sample_w.alloc(N);
  t.losses.alloc(64);
}

// out[0] = scale * sum(in[0 .. n)), fixed order: long inputs go through per-chunk workgroups
// first (train_kernels.h chunk_sum_kernel), the partials through the one-workgroup kernel.
void run_long_sum(mv_engine* e, TrainState& t, const float* in, size_t n, float* out,
                  float scale) {
  if (n <= 4 * (size_t)mv::kSumChunk) {
    hipLaunchKernelGGL(mv::reduce_sum_kernel, dim3(1), dim3(256), 0, e->stream, in, n, out,
                       scale, 0);
    return;
  }
  const size_t nblk = cdiv(n, mv::kSumChunk);
  MV_REQUIRE(nblk <= t.partial.n, "internal: long-sum partials");
  hipLaunchKernelGGL(mv::chunk_sum_kernel, dim3(nblk), dim3(256), 0, e->stream, in, n,
                     t.partial.p);
  hipLaunchKernelGGL(mv::reduce_sum_kernel, dim3(1), dim3(256), 0, e->stream, t.partial.p, nblk,
                     out, scale, 0);
}

// deterministic column sum of X [rows, ncols] into out [ncols]: slabs of rows are summed in
// parallel, then the partial rows are folded 32 at a time until one is left (a fixed tree:
// the result does not depend on timing).  Round 3: the fold of up to 1 024 partial rows used
// to be ONE serial loop per column in a single workgroup per 256 columns (240 us per bias
// gradient, 2 ms per training step); now every pass is at most 32 dependent adds.
void run_colsum(mv_engine* e, const float* x, size_t rows, size_t ncols, float* out,
                float* tmp /* >= 1056*ncols */) {
  size_t nslab = std::min<size_t>(1024, (rows + 63) / 64);
  if (nslab < 1) nslab = 1;
  const size_t rps = (rows + nslab - 1) / nslab;
  nslab = (rows + rps - 1) / rps;
  const unsigned gy = cdiv(ncols, 256);
  if (nslab == 1) {
    hipLaunchKernelGGL(mv::colsum_kernel, dim3(1, gy), dim3(256), 0, e->stream, x, out,
                       rows, ncols, rows);
    return;
  }
  if (ncols <= 128 && 256 % ncols == 0)       // narrow: all 256 threads on 1 KB runs
    hipLaunchKernelGGL(mv::colsum_narrow_kernel, dim3((unsigned)nslab), dim3(256), 0,
                       e->stream, x, tmp, rows, (int)ncols, rps);
  else
    hipLaunchKernelGGL(mv::colsum_kernel, dim3((unsigned)nslab, gy), dim3(256), 0,
                       e->stream, x, tmp, rows, ncols, rps);
  // fold the nslab partial rows: tmp rows [0, nslab) -> rows [1024, 1024 + ceil(nslab/32))
  const float* cur = tmp;
  size_t n = nslab;
  float* nxt = tmp + (size_t)1024 * ncols;
  while (n > 32) {
    const size_t m = (n + 31) / 32;
    hipLaunchKernelGGL(mv::colsum_kernel, dim3((unsigned)m, gy), dim3(256), 0, e->stream, cur,
                       nxt, n, ncols, (size_t)32);
    cur = nxt; n = m;
    nxt = (cur == tmp) ? tmp + (size_t)1024 * ncols : tmp;   // m <= 32 rows: fits either end
  }
  hipLaunchKernelGGL(mv::colsum_kernel, dim3(1, gy), dim3(256), 0, e->stream, cur, out, n,
                     ncols, n);
}

void run_small_dgrad(mv_engine* e, const float* dout, size_t dout_rs, const float* w,
                     float* din, size_t din_rs, int M, int H, int W, int Ci, int Co,
                     bool accumulate) {
  const size_t total = (size_t)M * H * W * Ci;
  const bool vec4 = (Ci % 4 == 0) && (Co == 1 || Co == 2) && (din_rs % 4 == 0) &&
                    total / 4 < 0xffffffffull;
  launch(e, "conv3x3_small_dgrad", 2.0 * total * 9 * Co, 4.0 * total * 2, [&] {
    if (vec4 && Co == 1)
      hipLaunchKernelGGL(mv::conv3x3_small_dgrad4_kernel<1>, dim3(cdiv(total / 4, 256)),
                         dim3(256), 0, e->stream, dout, dout_rs, w, din, din_rs, M, H, W, Ci,
                         accumulate ? 1 : 0);
    else if (vec4)
      hipLaunchKernelGGL(mv::conv3x3_small_dgrad4_kernel<2>, dim3(cdiv(total / 4, 256)),
                         dim3(256), 0, e->stream, dout, dout_rs, w, din, din_rs, M, H, W, Ci,
                         accumulate ? 1 : 0);
    else
      hipLaunchKernelGGL(mv::conv3x3_small_dgrad_kernel, dim3(cdiv(total, 256)), dim3(256),
                         0, e->stream, dout, dout_rs, w, din, din_rs, M, H, W, Ci, Co,
                         accumulate ? 1 : 0);
  });
}

// dW [9][Ci][Co] of a small 3x3 conv over R images
void run_small_wgrad(mv_engine* e, const float* in, const float* dout, float* dw, int R,
                     int H, int W, int Ci, int Co) {
  TrainState& t = TS(e);
  const bool h2g_form = Ci > Co && Ci <= 512 && (Co == 1 || Co == 2);   // hidden2grid
  MV_REQUIRE(Ci * Co <= 512 || h2g_form, "small wgrad: Ci*Co %d > 512", Ci * Co);
  const long long cells = (long long)R * H * W;
  const int P = Ci * Co;
  const int G = std::max(1, 256 / P);                 // cell groups per workgroup
  long long nblk = std::min<long long>(4096, (cells + 63) / 64);
  const int cpb = (int)((cells + nblk - 1) / nblk);
  nblk = (cells + cpb - 1) / cpb;
  const int threads = ((P * G + 63) / 64) * 64;
  const size_t ncols = (size_t)9 * P;
  const size_t lds = G > 1 ? (size_t)G * 9 * P * sizeof(float) : 0;
  MV_REQUIRE((size_t)nblk * ncols <= t.partial.n && 64 * ncols <= t.scratch.n,
             "internal: small wgrad partial buffer");
  // algorithmic bytes: the input and the output gradient are each read ONCE (the nine taps
  // re-read the same cells from LDS / L1), plus the per-block partial tiles written and
  // folded.  (Rounds 1-3 counted the input nine times: a "roofline fraction" of 1.05.)
  launch(e, "conv3x3_small_wgrad", 2.0 * cells * 9 * Ci * Co,
         4.0 * cells * ((double)Ci + Co) + 8.0 * (double)nblk * ncols, [&] {
    const size_t lds2 = ((size_t)(cpb + 2 * W + 2) * Co + cpb) * sizeof(float);
    MV_REQUIRE(Ci * Co <= 512 || lds2 <= 48 * 1024, "small wgrad: LDS of the hidden2grid form");
    if (h2g_form && G == 1 && lds2 <= 48 * 1024) {
      if (Co == 1)
        hipLaunchKernelGGL(mv::h2g_wgrad_kernel<1>, dim3((unsigned)nblk), dim3(256), lds2,
                           e->stream, in, dout, t.partial.p, R, H, W, Ci, cpb);
      else
        hipLaunchKernelGGL(mv::h2g_wgrad_kernel<2>, dim3((unsigned)nblk), dim3(256), lds2,
                           e->stream, in, dout, t.partial.p, R, H, W, Ci, cpb);
    } else if (Ci > Co)
      hipLaunchKernelGGL(mv::conv3x3_small_wgrad_kernel<true>, dim3((unsigned)nblk),
                         dim3(threads), lds, e->stream, in, dout, t.partial.p, R, H, W, Ci,
                         Co, cpb, G);
    else
      hipLaunchKernelGGL(mv::conv3x3_small_wgrad_kernel<false>, dim3((unsigned)nblk),
                         dim3(threads), lds, e->stream, in, dout, t.partial.p, R, H, W, Ci,
                         Co, cpb, G);
    // two-level fold of the nblk partial rows (fixed order)
    const size_t rps = ((size_t)nblk + 63) / 64, nslab = ((size_t)nblk + rps - 1) / rps;
    hipLaunchKernelGGL(mv::colsum_kernel, dim3((unsigned)nslab, cdiv(ncols, 256)), dim3(256),
                       0, e->stream, t.partial.p, t.scratch.p, (size_t)nblk, ncols, rps);
    hipLaunchKernelGGL(mv::colsum_kernel, dim3(1, cdiv(ncols, 256)), dim3(256), 0,
                       e->stream, t.scratch.p, dw, nslab, ncols, nslab);
  });
}

void run_pack(mv_engine* e, TrainChain& ch) {
  const int C = e->cfg.hidden_size, Cx = ch.Cx;
  ConvCell& cc = *ch.cell;
  if (e->cfg.convlstm_kernel != 3) {       // generic taps read the HWIO kernel itself
    cc.host_stale = true;
    return;
  }
  {
    const size_t total = mv::convlstm_wpack_elems(Cx, C);
    const int nx = mv::convlstm_xchunks(Cx), nch = nx + 9 * (C / mv::kBK);
    cc.wpack.alloc(total);
    hipLaunchKernelGGL(mv::pack_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                       e->stream, cc.kernel->dev.p, cc.wpack.p, Cx, C, nx, nch,
                       (Cx > 0 && 9 * Cx <= mv::kBK) ? 1 : 0, total);
  }
  {
    const size_t total = mv::convlstm_dgrad_wpack_elems(Cx, C);
    const int nch = 9 * (4 * C / mv::kBK);
    hipLaunchKernelGGL(mv::pack_dgrad_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                       e->stream, cc.kernel->dev.p, ch.wdpack.p, Cx, C, nch, total);
  }
  cc.host_stale = true;     // the host copy no longer matches the device weights
  if (e->compute_mode != 0) {   // matrix-pipe planes of the updated kernel, on the device
    const bool small = Cx > 0 && 9 * Cx <= mv::kBK;
    const int Cx16 = small ? 0 : Cx;
    if (e->compute_mode == 1) {
      const size_t halves = mv::f16x3_wpack_elems(Cx16, C);
      cc.wp16.alloc(halves);
      const size_t threads = halves / 2;
      hipLaunchKernelGGL(mv::pack_f16x3_kernel, dim3(cdiv(threads, 256)), dim3(256), 0,
                         e->stream, cc.kernel->dev.p, cc.wp16.p, Cx, Cx16, C, threads);
      pack_wino_forms(e, cc);     // releases both Winograd packs, re-packs what is enabled
      cc.wpb.release(); cc.wpbt.release(); cc.wx32u.release();
    } else {                    // bf16 forward (the backward's packs: below)
      const bool xf16 = e->cfg.activation != 0 && Cx16 > 0;
      const size_t halves = mv::bf16_wpack_elems(Cx16, C, xf16);
      cc.wpb.alloc(halves);
      hipLaunchKernelGGL(mv::pack_bf16_kernel, dim3(cdiv(halves, 256)), dim3(256), 0,
                         e->stream, cc.kernel->dev.p, cc.wpb.p, Cx, Cx16, C, halves,
                         xf16 ? 1 : 0);
      pack_bf16t(e, cc);
      cc.wp16.release(); cc.wx32.release(); cc.wpw.release(); cc.wpw3.release();
    }
    if (e->compute_mode == 2 && bf16_bwd_enabled(e)) {
      const size_t db = mv::bf16_dgrad_wpack_elems(Cx, C);
      ch.wdb.alloc(db);
      hipLaunchKernelGGL(mv::pack_bf16_dgrad_kernel, dim3(cdiv(db, 256)), dim3(256), 0,
                         e->stream, cc.kernel->dev.p, ch.wdb.p, Cx, C, db);
      ch.wd16.release(); ch.wdw.release();
    } else {
      ch.wdb.release();
      const size_t dh = mv::f16x3_dgrad_wpack_elems(Cx, C);
      ch.wd16.alloc(dh);
      hipLaunchKernelGGL(mv::pack_f16x3_dgrad_kernel, dim3(cdiv(dh / 2, 256)), dim3(256), 0,
                         e->stream, cc.kernel->dev.p, ch.wd16.p, Cx, C, dh / 2);
      if (e->compute_mode == 1 && mv::wino_enabled() && mv::wino_dgrad_enabled()) {
        const size_t dw = mv::wino_dgrad_wpack_elems(Cx, C);
        ch.wdw.alloc(dw);
        hipLaunchKernelGGL(mv::pack_wino_dgrad_kernel, dim3(cdiv(dw / 2, 256)), dim3(256), 0,
                           e->stream, cc.kernel->dev.p, ch.wdw.p, Cx, C, dw / 2);
      } else {
        ch.wdw.release();
      }
    }
    if (small) {
      const size_t n = (size_t)(C / mv::kChBlock) * mv::kBN * mv::kBK;
      const int nch = mv::convlstm_xchunks(Cx) + 9 * (C / mv::kBK);
      DevBuf<float>& dst = e->compute_mode == 1 ? cc.wx32 : cc.wx32u;
      dst.alloc(n);
      hipLaunchKernelGGL(mv::scale_xchunk_kernel, dim3(cdiv(n, 256)), dim3(256), 0, e->stream,
                         cc.wpack.p, dst.p, nch, n, e->compute_mode == 1 ? 65536.0f : 1.0f);
    }
  } else {
    cc.wp16.release(); cc.wx32.release();   // rebuilt lazily if the mode is switched on
    cc.wpb.release(); cc.wpbt.release(); cc.wx32u.release(); cc.wpw.release(); cc.wpw3.release();
  }
}

void train_pack_all(mv_engine* e) {
  for (int s = 0; s < e->cfg.num_scales; ++s) {
    if (!e->sc[s].use) continue;
    TrainScale& R = TS(e).sc[s];
    const int nb = e->cfg.use_single_decoder ? 1 : 2;
    for (int b = 0; b < nb; ++b) { run_pack(e, R.enc[b]); run_pack(e, R.dec[b]); }
  }
}

void upload_targets(mv_engine* e, const mv_targets* tg) {
  const mv_config& c = e->cfg;
  const size_t N = c.batch_size, Tp = e->pred_len;
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    MV_REQUIRE(tg->grid_pred_labels[s] && tg->grid_pred_regress[s],
               "grid_pred_labels/grid_pred_regress[%d] is NULL for an enabled scale", s);
    for (size_t i = 0; i < N * Tp; ++i)
      MV_REQUIRE(tg->grid_pred_labels[s][i] >= 0 && tg->grid_pred_labels[s][i] < S.K,
                 "grid_pred_labels[%d][%zu] = %d out of range [0,%d)", s, i,
                 tg->grid_pred_labels[s][i], S.K);
    TrainScale& R = TS(e).sc[s];
    HIP_CHECK(hipMemcpyAsync(R.pred_labels.p, tg->grid_pred_labels[s],
                             N * Tp * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    HIP_CHECK(hipMemcpyAsync(R.pred_reg.p, tg->grid_pred_regress[s],
                             N * Tp * S.K * 2 * sizeof(float), hipMemcpyHostToDevice,
                             e->stream));
  }
  HIP_CHECK(hipStreamSynchronize(e->stream));
}

void upload_targets_compact(mv_engine* e, const mv_targets_compact* tg) {
  const mv_config& c = e->cfg;
  const size_t N = c.batch_size, Tp = e->pred_len;
  MV_REQUIRE(tg->pred_xy, "pred_xy is NULL");
  MV_REQUIRE(tg->num_rows >= 0 && (size_t)tg->num_rows <= N, "num_rows %d not in [0, N=%zu]",
             tg->num_rows, N);
  e->xy_dev.alloc(2 * N * std::max<size_t>(c.obs_len, c.max_pred_len));
  HIP_CHECK(hipMemcpyAsync(e->xy_dev.p, tg->pred_xy, 2 * N * Tp * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    MV_REQUIRE(tg->grid_pred_labels[s], "grid_pred_labels[%d] is NULL for an enabled scale", s);
    MV_REQUIRE(S.centers.p, "mv_set_grid_centers(%d) has not been called", s);
    for (size_t i = 0; i < N * Tp; ++i)
      MV_REQUIRE(tg->grid_pred_labels[s][i] >= 0 && tg->grid_pred_labels[s][i] < S.K,
                 "grid_pred_labels[%d][%zu] = %d out of range [0,%d)", s, i,
                 tg->grid_pred_labels[s][i], S.K);
    TrainScale& R = TS(e).sc[s];
    HIP_CHECK(hipMemcpyAsync(R.pred_labels.p, tg->grid_pred_labels[s],
                             N * Tp * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(mv::regress_from_xy_kernel, dim3(cdiv(N * Tp * S.K, 256)), dim3(256),
                       0, e->stream, e->xy_dev.p, S.centers.p, R.pred_reg.p, (int)(N * Tp),
                       (int)Tp, S.K, tg->num_rows);
  }
  HIP_CHECK(hipStreamSynchronize(e->stream));
}

ConvLstmArgs train_problem(mv_engine* e, TrainChain& ch, const float* x, const float* h,
                           const float* c, float* h_out, float* c_out, float* gates,
                           int H, int W, bool zero_state) {
  ConvLstmArgs a = conv_problem(e, *ch.cell, x, h, c, nullptr, nullptr, h_out, c_out,
                                e->cfg.batch_size, H, W, zero_state);
  a.gates_out = gates;
  return a;
}

// ------------------------------------------------------------ forward (is_train)
// The training forward runs in the engine's compute mode (f16x3: the gate
// convolutions on the fp16 matrix pipe, saving the same fp32 gate activations);
// dgrad and wgrad stay on the fp32 MFMA.
// keep_prob < 1: input dropout of one cell call, in place on its x operand (forward) or
// on d x (backward); draw = chain offset + step, numbered in the reference's call order
// (per used scale: class encoder, regression encoder, class decoder, regression decoder)
int dropout_stream(mv_engine* e, int s, int chain, int step) {
  const int To = e->cfg.obs_len, Tp = e->pred_len;
  int used = 0;
  for (int i = 0; i < s; ++i) used += e->sc[i].use ? 1 : 0;
  const int base = used * (2 * To + 2 * Tp);
  const int off[4] = {0, To, 2 * To, 2 * To + Tp};
  return base + off[chain] + step;
}
void run_dropout(mv_engine* e, float* x, size_t n, int s, int chain, int step) {
  TrainState& t = TS(e);
  if (!(t.tc.keep_prob < 1.0f) || n == 0) return;
  MV_REQUIRE(n < ((size_t)1 << 32), "dropout: tensor of %zu elements", n);
  const uint32_t thr = (uint32_t)lrintf(t.tc.keep_prob * 16777216.0f);
  launch(e, "dropout", 0, 8.0 * n, [&] {
    hipLaunchKernelGGL(mv::dropout_kernel, dim3(cdiv(n, 256)), dim3(256), 0, e->stream, x, n,
                       t.dropout_seed, (uint32_t)dropout_stream(e, s, chain, step), thr,
                       1.0f / t.tc.keep_prob);
  });
}

// ground-truth class maps / teacher-forcing inputs / fg count of this batch's targets
void train_prepare_targets(mv_engine* e) {
  const mv_config& c = e->cfg;
  TrainState& t = TS(e);
  const int N = c.batch_size, Tp = e->pred_len;
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    TrainScale& R = t.sc[s];
    const size_t NK = (size_t)N * S.K;
    if (t.needs_gt()) {
      mv::SoftKernel sk{};
      const int ks = t.tc.use_soft_grid_class ? t.tc.soft_kernel_size : 0;
      for (int i = 0; i < ks * ks; ++i) sk.k[i] = t.tc.soft_kernel[i];
      hipLaunchKernelGGL(mv::gt_class_maps_kernel, dim3(cdiv((size_t)Tp * NK, 256)), dim3(256),
                         0, e->stream, R.pred_labels.p, R.gt_cls.p, Tp, N, S.H, S.W, ks, sk);
    }
    if (t.tc.mask_grid_regression) {
      HIP_CHECK(hipMemsetAsync(R.fg_count.p, 0, sizeof(int32_t), e->stream));
      hipLaunchKernelGGL(mv::count_positive_kernel, dim3(256), dim3(256), 0, e->stream,
                         R.gt_cls.p, (size_t)Tp * NK, R.fg_count.p);
    }
    if (t.tc.reg_teacher_forcing) {       // [N][Tp][K][2] -> time-major
      const size_t tot = (size_t)N * Tp * S.K * 2;
      hipLaunchKernelGGL(mv::transpose_nt_kernel, dim3(cdiv(tot, 256)), dim3(256), 0,
                         e->stream, R.pred_reg.p, R.reg_in.p, N, Tp, (size_t)S.K * 2);
    }
  }
}

void train_forward(mv_engine* e) {
  const mv_config& c = e->cfg;
  TrainState& t = TS(e);
  const int N = c.batch_size, To = c.obs_len, Tp = e->pred_len, C = c.hidden_size,
            D = c.scene_conv_dim, E = c.emb_size;
  const int cls_fb = t.tc.class_feedback;        // 0 one-hot, 1 dense logits, 2 ground truth
  const bool reg_tf = t.tc.reg_teacher_forcing != 0;
  const int nb = c.use_single_decoder ? 1 : 2;   // branches run: class (+ regression)
  train_prepare_targets(e);
  run_scene(e);
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    TrainScale& R = t.sc[s];
    if (c.use_single_decoder) continue;
    // regression-encoder x operand, time-major
    const size_t tot = (size_t)N * To * S.K * 2;
    hipLaunchKernelGGL(mv::transpose_nt_kernel, dim3(cdiv(tot, 256)), dim3(256), 0,
                       e->stream, S.obs_reg.p, R.enc[1].xs.p, N, To, (size_t)S.K * 2);
  }
  // encoders, all chains in lockstep (dynamic_rnn from the zero state)
  for (int ts = 0; ts < To; ++ts) {
    std::vector<ConvLstmArgs> probs;
    for (int s = 0; s < c.num_scales; ++s) {
      ScaleState& S = e->sc[s];
      if (!S.use) continue;
      TrainScale& R = t.sc[s];
      const size_t NKC = (size_t)N * S.K * C;
      float* xc = R.enc[0].xs.p + (size_t)ts * N * S.K * D;
      const size_t total = (size_t)N * S.K * D;
      launch(e, "enc_class_input", 0, 4.0 * total, [&] {
        hipLaunchKernelGGL(mv::enc_class_input_kernel, dim3(cdiv(total, 256)), dim3(256),
                           0, e->stream, e->scene_conv[s].p, e->obs_scene.p, S.labels.p,
                           xc, N, To, ts, S.K, D, (_Float16*)nullptr, (size_t)0,
                           t.mix_on ? R.obs_labels2.p : (const int32_t*)nullptr, t.mix_w);
      });
      run_dropout(e, xc, total, s, 0, ts);
      if (nb == 2)
        run_dropout(e, R.enc[1].xs.p + (size_t)ts * N * S.K * 2, (size_t)N * S.K * 2, s, 1, ts);
      for (int b = 0; b < nb; ++b) {
        const float* x = b == 0 ? xc : R.enc[1].xs.p + (size_t)ts * N * S.K * 2;
        probs.push_back(train_problem(
            e, R.enc[b], x, R.hs[b].p + ts * NKC, R.cs[b].p + ts * NKC,
            R.hs[b].p + (ts + 1) * NKC, R.cs[b].p + (ts + 1) * NKC,
            R.enc[b].gates.p + (size_t)ts * 4 * NKC, S.H, S.W, ts == 0));
      }
    }
    run_conv_group(e, probs);
  }
  // decoders (grid_decoder under raw_rnn, input_onehot for the class decoder)
  bool tail_embedded = false;      // the previous step's tail wrote this step's embeddings
  for (int ts = 0; ts < Tp; ++ts) {
    std::vector<ConvLstmArgs> probs;
    for (int s = 0; s < c.num_scales; ++s) {
      ScaleState& S = e->sc[s];
      if (!S.use) continue;
      TrainScale& R = t.sc[s];
      const size_t NK = (size_t)N * S.K, NKC = NK * C;
      const int slot = To + ts;
      // class decoder
      const float* hin = R.hs[0].p + slot * NKC;
      if (c.use_gnn) {
        run_gnn(e, S, hin, nullptr, R.hg.p + ts * NKC, N, 1);
        hin = R.hg.p + ts * NKC;
      }
      float* xc = R.dec[0].xs.p + (size_t)ts * NK * E;
      if (ts == 0 && t.mix_on) {
        // label mixup: the first input is grid_emb of the MIXED map of the last observed
        // step (obs_grid_class[:, -1], SimAug/code/pred_models.py:616-636), kept in slot 0
        // of `onehot` for the embedding's wgrad
        hipLaunchKernelGGL(mv::twohot_map_kernel, dim3(cdiv(NK, 256)), dim3(256), 0, e->stream,
                           S.labels.p + (To - 1), R.obs_labels2.p + (To - 1), To, 0, t.mix_w,
                           R.onehot.p, 1, N, S.K);
        run_emb_dense(e, S, R.onehot.p, (size_t)S.K, xc, N, S.emb_cls_W, S.emb_cls_b, 1);
      } else if (ts == 0) run_emb_onehot(e, S, S.labels.p + (To - 1), To, xc, N);
      else if (cls_fb == 0) {
        if (!tail_embedded) run_emb_onehot(e, S, R.ids.p + (size_t)ts * N, 1, xc, N);
      }
      else   // dense input map: the previous step's logits, or the step's ground truth
        run_emb_dense(e, S, cls_fb == 1 ? R.logits.p + (size_t)(ts - 1) * NK
                                        : R.gt_cls.p + (size_t)ts * NK,
                      (size_t)S.K, xc, N, S.emb_cls_W, S.emb_cls_b, 1);
      run_dropout(e, xc, NK * E, s, 2, ts);
      probs.push_back(train_problem(e, R.dec[0], xc, hin, R.cs[0].p + slot * NKC,
                                    R.hs[0].p + (slot + 1) * NKC,
                                    R.cs[0].p + (slot + 1) * NKC,
                                    R.dec[0].gates.p + (size_t)ts * 4 * NKC, S.H, S.W,
                                    false));
      if (nb == 1) continue;       // single decoder: offsets come from the class states
      // regression decoder
      float* xr = R.dec[1].xs.p + (size_t)ts * NK * E;
      if (ts == 0) {   // regio slot 0 = obs_grid_reg[:, -1]
        const size_t row = (size_t)S.K * 2;
        HIP_CHECK(hipMemcpy2DAsync(R.regio.p, row * sizeof(float),
                                   S.obs_reg.p + (size_t)(To - 1) * row,
                                   (size_t)To * row * sizeof(float), row * sizeof(float),
                                   N, hipMemcpyDeviceToDevice, e->stream));
      }
      if (reg_tf && ts > 0)       // teacher forcing: grid_pred_regress[:, ts] (:398)
        run_emb_dense(e, S, R.reg_in.p + (size_t)ts * NK * 2, (size_t)S.K * 2, xr, N);
      else if (!(tail_embedded && ts > 0))
        run_emb_dense(e, S, R.regio.p + (size_t)ts * NK * 2, (size_t)S.K * 2, xr, N);
      run_dropout(e, xr, NK * E, s, 3, ts);
      probs.push_back(train_problem(e, R.dec[1], xr, R.hs[1].p + slot * NKC,
                                    R.cs[1].p + slot * NKC, R.hs[1].p + (slot + 1) * NKC,
                                    R.cs[1].p + (slot + 1) * NKC,
                                    R.dec[1].gates.p + (size_t)ts * 4 * NKC, S.H, S.W,
                                    false));
    }
    run_conv_group(e, probs);
    // Decoder tail of the step on the inference kernels (round 3): hidden2grid of every chain
    // as ONE grouped GEMM launch (h2g_q) + one gather / argmax / next-embedding kernel
    // (decode_tail) instead of two hidden2grid convolutions, an argmax and two embedding
    // launches per scale.  The embeddings of step ts+1 land in the time-major x buffers the
    // backward pass reads; dense class feedback and teacher forcing keep their own embedding
    // launches at the head of the next iteration.  MV_TRAIN_TAIL=v1 restores the old launches.
    static const bool tail2 = tail_v2() &&
        !(getenv("MV_TRAIN_TAIL") && strcmp(getenv("MV_TRAIN_TAIL"), "v1") == 0);
    bool tail_fits = E % 16 == 0 && E <= 512;   // run_tail's own limits (decode_tail LDS)
    for (int s = 0; s < c.num_scales; ++s)
      if (e->sc[s].use && (size_t)e->sc[s].K * 2 > 2048) tail_fits = false;
    if (tail2 && tail_fits) {
      std::vector<TailPlan> plans;
      for (int s = 0; s < c.num_scales; ++s) {
        ScaleState& S = e->sc[s];
        if (!S.use) continue;
        TrainScale& R = t.sc[s];
        const size_t NK = (size_t)N * S.K, NKC = NK * C;
        const int slot = To + ts + 1;
        const bool more = ts + 1 < Tp;
        TailPlan pl{};
        pl.s = s;
        pl.cls_h = R.hs[0].p + slot * NKC; pl.cls_rows = N;
        pl.cls_out = R.logits.p + (size_t)ts * NK; pl.cls_stride = (int64_t)S.K;
        pl.cls_next = more && cls_fb == 0;
        pl.cls_ids_out = R.ids.p + (size_t)(ts + 1) * N;
        pl.cls_x_out = R.dec[0].xs.p + (size_t)(ts + 1) * NK * E;
        pl.reg_h = R.hs[nb - 1].p + slot * NKC;
        pl.reg_out = R.regio.p + (size_t)(ts + 1) * NK * 2; pl.reg_stride = (int64_t)S.K * 2;
        pl.reg_next = more && nb == 2 && !reg_tf;
        pl.reg_x_out = nb == 2 ? R.dec[1].xs.p + (size_t)(ts + 1) * NK * E : nullptr;
        plans.push_back(pl);
      }
      run_tail(e, plans);
      tail_embedded = true;
      continue;
    }
    for (int s = 0; s < c.num_scales; ++s) {
      ScaleState& S = e->sc[s];
      if (!S.use) continue;
      TrainScale& R = t.sc[s];
      const size_t NK = (size_t)N * S.K, NKC = NK * C;
      const int slot = To + ts + 1;
      float* lg = R.logits.p + (size_t)ts * NK;
      run_hidden2grid<1>(e, S, R.hs[0].p + slot * NKC, S.out_cls_W->dev.p, lg,
                         (size_t)S.K, N);
      if (ts + 1 < Tp && cls_fb == 0) {
        launch(e, "argmax_rows", 0, 4.0 * NK, [&] {
          hipLaunchKernelGGL(mv::argmax_rows_kernel, dim3(N), dim3(64), 0, e->stream, lg,
                             (size_t)S.K, R.ids.p + (size_t)(ts + 1) * N, N, S.K);
        });
      }
      run_hidden2grid<2>(e, S, R.hs[nb - 1].p + slot * NKC, S.out_reg_W->dev.p,
                         R.regio.p + (size_t)(ts + 1) * NK * 2, (size_t)S.K * 2, N);
    }
  }
}

// ------------------------------------------------------------ losses
void train_losses(mv_engine* e) {
  const mv_config& c = e->cfg;
  TrainState& t = TS(e);
  const int N = c.batch_size, Tp = e->pred_len;
  int li = 0;
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    TrainScale& R = t.sc[s];
    const size_t NK = (size_t)N * S.K;
    const float cs = t.tc.grid_loss_weight / (float)((size_t)N * Tp);
    launch(e, "ce_loss", 0, 8.0 * Tp * NK, [&] {
      if (t.mix_on) {
        // mixed-up targets w * one_hot(label) + one_hot(extra view's label) * (1 - w)
        // (SimAug/code/pred_models.py:1371-1390), optionally weighted per sample (:1391-1398)
        hipLaunchKernelGGL(mv::twohot_map_kernel, dim3(cdiv((size_t)Tp * NK, 256)), dim3(256), 0,
                           e->stream, R.pred_labels.p, R.pred_labels2.p, Tp, 1, t.mix_w,
                           R.mix_cls.p, Tp, N, S.K);
        hipLaunchKernelGGL(mv::ce_soft_loss_kernel, dim3(Tp * N), dim3(64), 0, e->stream,
                           R.logits.p, R.mix_cls.p, R.loss_row.p, R.dlogits.p, S.K, cs);
        if (t.mix_sw)
          hipLaunchKernelGGL(mv::scale_rows_kernel, dim3(cdiv((size_t)Tp * NK, 256)), dim3(256),
                             0, e->stream, R.loss_row.p, R.dlogits.p, t.sample_w.p, Tp * N, N,
                             S.K);
      } else if (t.tc.use_soft_grid_class)
        hipLaunchKernelGGL(mv::ce_soft_loss_kernel, dim3(Tp * N), dim3(64), 0, e->stream,
                           R.logits.p, R.gt_cls.p, R.loss_row.p, R.dlogits.p, S.K, cs);
      else
        hipLaunchKernelGGL(mv::ce_loss_kernel, dim3(Tp * N), dim3(64), 0, e->stream,
                           R.logits.p, R.pred_labels.p, R.loss_row.p, R.dlogits.p, Tp, N,
                           S.K, cs);
      hipLaunchKernelGGL(mv::reduce_sum_kernel, dim3(1), dim3(256), 0, e->stream,
                         R.loss_row.p, (size_t)Tp * N, t.losses.p + li, cs, 0);
    });
    const size_t nel = (size_t)Tp * NK * 2;
    const float rs = t.tc.grid_reg_loss_weight / (float)nel;
    launch(e, "huber_loss", 0, 16.0 * nel, [&] {
      if (t.tc.mask_grid_regression) {
        hipLaunchKernelGGL(mv::huber_masked_loss_kernel, dim3(cdiv(nel, 256)), dim3(256), 0,
                           e->stream, R.regio.p + NK * 2, R.pred_reg.p, R.gt_cls.p,
                           R.fg_count.p, R.loss_elem.p, R.dreg.p, Tp, N, S.K,
                           t.tc.grid_reg_loss_weight);
        run_long_sum(e, t, R.loss_elem.p, nel, t.scratch.p, 1.0f);
        hipLaunchKernelGGL(mv::masked_mean_kernel, dim3(1), dim3(64), 0, e->stream,
                           t.scratch.p, R.fg_count.p, t.losses.p + li + 1,
                           t.tc.grid_reg_loss_weight);
      } else {
        hipLaunchKernelGGL(mv::huber_loss_kernel, dim3(cdiv(nel, 256)), dim3(256), 0,
                           e->stream, R.regio.p + NK * 2, R.pred_reg.p, R.loss_elem.p,
                           R.dreg.p, Tp, N, S.K * 2, rs);
        run_long_sum(e, t, R.loss_elem.p, nel, t.losses.p + li + 1, rs);
      }
    });
    li += 2;
  }
}

// ------------------------------------------------------------ backward
// group_slot >= 0 (compute mode 2 with the bf16 backward): the step's G also leaves as the bf16
// operand plane of the dgrad problem that will take that slot of the group (t.g16[slot]), and
// run_dgrad_group_f16x3 skips its split pass.  MV_BF16_FUSED_SPLIT=0: the separate pass.
void run_gate_bwd(mv_engine* e, float* gates, const float* c_prev, const float* c_new,
                  const float* dh, float* dc, size_t cells, int C,
                  int32_t* gmax_bits = nullptr, int group_slot = -1) {
  const size_t total = cells * C;
  static const bool fuse_on = !(getenv("MV_BF16_FUSED_SPLIT") && atoi(getenv("MV_BF16_FUSED_SPLIT")) == 0);
  _Float16* plane = nullptr;
  if (fuse_on && group_slot >= 0 && group_slot < mv::kMaxGroup && C % 32 == 0 &&
      e->compute_mode == 2 && bf16_bwd_enabled(e)) {
    TrainState& t = TS(e);
    const size_t n = cells * 4 * (size_t)C;
    if (t.g16[group_slot].n >= 2 * (n + mv::kPlaneSlack + mv::kPlanePad)) {
      plane = t.g16[group_slot].p + mv::kPlanePad;
      t.g16_fused[group_slot] = true;
    }
  }
  launch(e, "lstm_gate_bwd", 30.0 * total, 4.0 * total * (plane ? 15 : 13), [&] {
    if (plane)
      hipLaunchKernelGGL(mv::lstm_gate_bwd4_plane_kernel,
                         dim3((unsigned)(((cells + 31) / 32) * (size_t)(C / 32))), dim3(256), 0,
                         e->stream, gates, c_prev, c_new, dh, dc, (long long)cells, C, plane,
                         gmax_bits);
    else if (C % 4 == 0)
      hipLaunchKernelGGL(mv::lstm_gate_bwd4_kernel, dim3(cdiv(total / 4, 256)), dim3(256), 0,
                         e->stream, gates, c_prev, c_new, dh, dc, total / 4, C, gmax_bits);
    else
    hipLaunchKernelGGL(mv::lstm_gate_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                       e->stream, gates, c_prev, c_new, dh, dc, total, C, gmax_bits);
  });
}

// f16x3 compute mode: G of every problem -> two fp16 planes under its own
// power-of-two scale, then the grouped dgrad launch on the fp16 matrix pipe.
void run_dgrad_group_f16x3(mv_engine* e, const std::vector<ConvLstmArgs>& probs,
                           const std::vector<TrainChain*>& chains,
                           const std::vector<int>& slots, double fl, double by) {
  TrainState& t = TS(e);
  std::vector<mv::ConvLstm16Args> p16(probs.size());
  // the Winograd F(2,3) form of the same convolution (two thirds of the MFMAs) when every
  // problem of the group fits its tiling; MV_WINO_DGRAD=0 keeps the direct kernel
  const bool bf = e->compute_mode == 2 && bf16_bwd_enabled(e);
  bool wino = e->compute_mode == 1 && mv::wino_enabled() && mv::wino_dgrad_enabled();
  for (size_t i = 0; i < probs.size() && wino; ++i) {
    const ConvLstmArgs& a = probs[i];
    if (!(a.W > 0 && 32 % a.W == 0 && a.H >= 2 && a.out0_cols % 64 == 0 && (a.C / 16) % 8 == 0) ||
        !chains[i]->wdw.p)
      wino = false;
  }
  std::vector<mv::ConvLstmWinoArgs> pw(wino ? probs.size() : 0);
  for (size_t i = 0; i < probs.size(); ++i) {
    const ConvLstmArgs& a = probs[i];
    mv::ConvLstm16Args& q = p16[i];
    q = mv::ConvLstm16Args{};
    q.f = a;
    const size_t n = (size_t)a.rows * a.H * a.W * a.C;      // a.C == 4C here
    const size_t pst = n + mv::kPlaneSlack + mv::kPlanePad;
    const size_t gcells = (size_t)a.rows * a.H * a.W;
    MV_REQUIRE(t.g16[i].n >= 2 * pst, "internal: G plane scratch");
    _Float16* p0 = t.g16[i].p + mv::kPlanePad;
    const bool fused = bf && t.g16_fused[i];       // the plane came out of lstm_gate_bwd
    t.g16_fused[i] = false;
    if (!fused)
    launch(e, "split_planes", 0, (bf ? 6.0 : 8.0) * n, [&] {
      if (bf)
        hipLaunchKernelGGL(mv::split_plane_bf16_kernel, dim3(mv::split_planes_blocks(gcells, a.C)),
                           dim3(256), 0, e->stream, a.h, p0, (int)gcells, a.C);
      else
      hipLaunchKernelGGL(mv::split_planes_dyn_kernel, dim3(mv::split_planes_blocks(gcells, a.C)),
                         dim3(256), 0, e->stream, a.h, p0, p0 + pst, (int)gcells, a.C,
                         t.gmax.p + (size_t)slots[i] * 64, t.gexp.p + slots[i]);
    });
    q.h16 = p0; q.h_plane_stride = (int64_t)pst;
    q.x16 = nullptr; q.x_plane_stride = 0;
    q.wp16 = bf ? chains[i]->wdb.p : chains[i]->wd16.p;
    MV_REQUIRE(q.wp16, "internal: dgrad weight pack of the compute mode is missing");
    q.n_xk = 0; q.n_hk = 9 * (a.C / 16); q.w_ksteps = q.n_hk;
    q.g_exp = bf ? nullptr : t.gexp.p + slots[i];
    const size_t M = (size_t)a.rows * a.H * a.W;
    if (wino) {
      // split-K of the Winograd form: (d h column block, slice) combos = 8 = the XCDs, the d x
      // blocks a region of their own with eight slices (convlstm_dgrad_wino_kernel)
      const int ncm = a.out0_cols / 64;
      const bool need_dx = a.out1 && a.out1_cols > 0;
      int nkm = std::max(1, 8 / ncm);
      while ((a.C / 16) % nkm != 0) nkm /= 2;
      const int nkx = need_dx ? 8 : 1;
      mv::ConvLstmWinoArgs& w = pw[i];
      w.nks_main = nkm; w.nks_x = nkx;
      if (nkm > 1) { t.dpart0[i].alloc((size_t)nkm * M * a.out0_cols); q.part0 = t.dpart0[i].p; }
      if (need_dx) {
        t.dpart1[i].alloc((size_t)nkx * M * ((a.out1_cols + 3) / 4 * 4));
        q.part1 = t.dpart1[i].p;
      }
      w.b = q;
      w.b.f.n_colblocks = need_dx ? mv::wino_dgrad_colblocks(a.out1_cols, a.out0_cols) : ncm;
      w.wpw = chains[i]->wdw.p;
      w.n_xc = 0;
      continue;
    }
    // split-K over four channel-group ranges (see convlstm16_dgrad_dispatch)
    const int nstages = q.n_hk / 3;
    // (the bf16 dgrad is a third as long: whole-K tiles, no partial sums -- 1 208 vs 1 202 /
    // 1 178 traj/s with 2 / 4 slices at batch 64, profiles/r4u_*)
    static const int ks_set = getenv("MV_DGRAD_KSLICES") ? atoi(getenv("MV_DGRAD_KSLICES")) : 0;
    const int ks_env = ks_set > 0 ? ks_set : (bf ? 1 : 4);
    if (bf)
      MV_REQUIRE(nstages % MV_BF16_UNITS == 0, "internal: bf16 dgrad stage count %d", nstages);
    if (ks_env > 1 && nstages % (2 * ks_env) == 0) {
      q.n_kslice = ks_env;
      t.dpart0[i].alloc((size_t)ks_env * M * a.out0_cols);
      q.part0 = t.dpart0[i].p;
      if (a.out1_cols > 0) {
        t.dpart1[i].alloc((size_t)ks_env * M * ((a.out1_cols + 3) / 4 * 4));
        q.part1 = t.dpart1[i].p;
      }
    }
  }
  launch(e, "convlstm_dgrad", fl, by, [&] {
    if (wino)
      mv::launch_convlstm_wino_dgrads(pw.data(), (int)pw.size(), e->stream);
    else
      mv::launch_convlstm16_dgrads(p16.data(), (int)p16.size(), e->stream, bf);
  }, -1.0, wino ? 2.0 : (bf ? 1.0 : 3.0));
  mv::SumSlicesArgs sa{};
  unsigned blocks = 0;
  double sbytes = 0;
  for (size_t i = 0; i < p16.size(); ++i) {
    const mv::ConvLstm16Args& q = p16[i];
    const int ns0 = wino ? pw[i].nks_main : q.n_kslice, ns1 = wino ? pw[i].nks_x : q.n_kslice;
    const size_t M = (size_t)q.f.rows * q.f.H * q.f.W;
    sa.nslice = 1;
    for (int o = 0; o < 2; ++o) {
      const int ns = o ? ns1 : ns0;
      if (ns <= 1) continue;
      float* out = o ? q.f.out1 : q.f.out0;
      const size_t n = M * (size_t)(o ? q.f.out1_cols : q.f.out0_cols);
      if (!out || n == 0) continue;
      MV_REQUIRE(n % 4 == 0, "internal: dgrad slice sum needs a multiple of 4 elements");
      sa.part[sa.nseg] = o ? q.part1 : q.part0;
      sa.out[sa.nseg] = out;
      sa.n[sa.nseg] = n;
      sa.nslice_seg[sa.nseg] = ns;
      blocks += cdiv(n / 4, 256);
      sa.block_end[sa.nseg] = blocks;
      sbytes += 4.0 * n * (ns + 1);
      ++sa.nseg;
    }
  }
  if (sa.nseg > 0)
    launch(e, "dgrad_slice_sum", 0, sbytes, [&] {
      hipLaunchKernelGGL(mv::sum_slices_kernel, dim3(blocks), dim3(256), 0, e->stream, sa);
    });
}

void run_dgrad_group(mv_engine* e, const std::vector<ConvLstmArgs>& probs,
                     const std::vector<double>& flops,
                     const std::vector<TrainChain*>& chains, const std::vector<int>& slots) {
  if (probs.empty()) return;
  double fl = 0, by = 0;
  for (size_t i = 0; i < probs.size(); ++i) {
    fl += flops[i];
    by += (double)probs[i].rows * probs[i].H * probs[i].W * (probs[i].C + 288.0) * 4.0;
  }
  if (e->compute_mode != 0) {
    run_dgrad_group_f16x3(e, probs, chains, slots, fl, by);
    return;
  }
  if (e->cfg.convlstm_kernel != 3) {
    launch(e, "convlstm_dgrad", fl, by, [&] {
      for (size_t i = 0; i < probs.size(); ++i) {
        const ConvLstmArgs& a = probs[i];        // convlstm_dgrad_args: h = G, C = 4C
        mv::ConvGenericDgradArgs ga{};
        ga.g = a.h; ga.w = chains[i]->cell->kernel->dev.p;
        ga.dh = a.out0; ga.dx = a.out1;
        ga.rows = a.rows; ga.H = a.H; ga.W = a.W; ga.Cx = a.out1_cols; ga.C = a.out0_cols;
        ga.ksize = e->cfg.convlstm_kernel;
        mv::launch_convlstm_generic_dgrad(ga, e->stream);
      }
    }, -1.0, 0.0);
    return;
  }
  launch(e, "convlstm_dgrad", fl, by, [&] {
    mv::launch_convlstm_dgrads(probs.data(), (int)probs.size(), e->stream);
  }, -1.0, 1.0);
}

// In-library gradient all-reduce (comm.h): the elements [off, off + n) of the flat
// gradient buffer are final on the main stream -> reduce them (sum over the ranks) on the
// side stream while the backward pass goes on.
// MV_COMM_FAULT -- ONLY in -DMV_TEST_HOOKS builds (the test-hooks copy of the library that
// __graft_entry__.build() puts under tests/fake_rccl/; the shipped library compiles the waits
// unconditionally): the negative control of tests/test_gpu_parallel.py -- bit 1 drops the side
// stream's wait for the main stream's `ready` event, bit 2 the main stream's wait for the side
// stream's `done`: over an asynchronous collective either must give wrong parameters (and the
// test must see it).
#ifdef MV_TEST_HOOKS
static inline int comm_fault() {
  static const int f = getenv("MV_COMM_FAULT") ? atoi(getenv("MV_COMM_FAULT")) : 0;
  return f;
}
#else
static constexpr int comm_fault() { return 0; }
#endif
void comm_reduce_range(mv_engine* e, size_t off, size_t n) {
  mv::Comm* c = e->comm;
  if (!c || n == 0) return;
  TrainState& t = TS(e);
  // a SimAug attack pass (mv_attack_*) wants d loss / d scene_feat only: the parameter
  // gradients of that pass are never applied, so there is nothing to exchange
  if (t.want_dscene) return;
  HIP_CHECK(hipEventRecord(c->ready, e->stream));
  if (!(comm_fault() & 1)) HIP_CHECK(hipStreamWaitEvent(c->stream, c->ready, 0));
  float* p = t.grad.p + off;
  ncclResult_t rc = mv::rccl().AllReduce(p, p, n, ncclFloat, ncclSum, c->comm, c->stream);
  MV_REQUIRE(rc == ncclSuccess, "ncclAllReduce(%zu floats at %zu): %s", n, off,
             mv::rccl().GetErrorString(rc));
  c->buckets_last += 1;
  c->bytes_last += 4.0 * n;
}
// one bucket = a ConvLSTM kernel and its biases (adjacent in the parameter table)
void comm_reduce_cell(mv_engine* e, const ConvCell& cc) {
  if (!e->comm) return;
  TrainState& t = TS(e);
  const size_t ik = param_index(e, cc.kernel), ib = param_index(e, cc.biases);
  MV_REQUIRE(ib == ik + 1, "internal: kernel / biases not adjacent");
  const size_t end = ib + 1 < t.goff.size() ? t.goff[ib + 1] : t.total_elems;
  comm_reduce_range(e, t.goff[ik], end - t.goff[ik]);
}
// everything that is not a ConvLSTM bucket (scene convs, embeddings, hidden2grid), as
// one group once the weight-decay pass has touched the */W gradients; then the main
// stream waits for the side stream
void comm_reduce_rest_and_join(mv_engine* e) {
  mv::Comm* c = e->comm;
  if (!c) return;
  TrainState& t = TS(e);
  if (t.want_dscene) return;           // attack pass: see comm_reduce_range
  std::vector<char> taken(e->params.size(), 0);
  // parameters that exist without a gradient (--use_single_decoder: the regression
  // encoder) are in no bucket and must not ride along in the "rest" group either
  for (size_t i = 0; i < e->params.size(); ++i)
    if (e->params[i]->no_grad) taken[i] = 1;
  for (int s = 0; s < e->cfg.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    for (ConvCell* cc : active_cells(e, S)) {
      taken[param_index(e, cc->kernel)] = 1;
      taken[param_index(e, cc->biases)] = 1;
    }
  }
  HIP_CHECK(hipEventRecord(c->ready, e->stream));
  if (!(comm_fault() & 1)) HIP_CHECK(hipStreamWaitEvent(c->stream, c->ready, 0));
  mv::RcclApi& r = mv::rccl();
  ncclResult_t rc = r.GroupStart();
  MV_REQUIRE(rc == ncclSuccess, "ncclGroupStart: %s", r.GetErrorString(rc));
  for (size_t i = 0; i < e->params.size();) {
    if (taken[i]) { ++i; continue; }
    size_t j = i;
    while (j < e->params.size() && !taken[j]) ++j;      // a contiguous run of small tensors
    const size_t off = t.goff[i], end = j < t.goff.size() ? t.goff[j] : t.total_elems;
    float* p = t.grad.p + off;
    rc = r.AllReduce(p, p, end - off, ncclFloat, ncclSum, c->comm, c->stream);
    MV_REQUIRE(rc == ncclSuccess, "ncclAllReduce: %s", r.GetErrorString(rc));
    c->buckets_last += 1;
    c->bytes_last += 4.0 * (end - off);
    i = j;
  }
  rc = r.GroupEnd();
  MV_REQUIRE(rc == ncclSuccess, "ncclGroupEnd: %s", r.GetErrorString(rc));
  HIP_CHECK(hipEventRecord(c->done, c->stream));
  if (!(comm_fault() & 2)) HIP_CHECK(hipStreamWaitEvent(e->stream, c->done, 0));
}

// gslot: first gmax slot of the chain's steps (f16x3 mode), see train_backward
void run_wgrad(mv_engine* e, TrainChain& ch, const float* hin, int Tsteps, int H, int W,
               int gslot) {
  TrainState& t = TS(e);
  const int N = e->cfg.batch_size, C = e->cfg.hidden_size;
  mv::WgradArgs wa{};
  wa.x = ch.Cx ? ch.xs.p : nullptr; wa.h = hin; wa.g = ch.gates.p;
  wa.partial = t.partial.p;
  wa.R = Tsteps * N; wa.H = H; wa.W = W; wa.Cx = ch.Cx; wa.C = C;
  mv::wgrad_plan(wa, 3072);
  MV_REQUIRE(mv::wgrad_partial_elems(wa) <= t.partial.n, "internal: wgrad partial buffer");
  const double cells = (double)wa.R * H * W;
  if (e->cfg.convlstm_kernel != 3) {       // generic taps: one deterministic pass, no split
    const int k = e->cfg.convlstm_kernel;
    launch(e, "convlstm_wgrad", 2.0 * cells * k * k * (ch.Cx + C) * 4.0 * C,
           cells * (ch.Cx + 5.0 * C) * 4.0, [&] {
      mv::ConvGenericWgradArgs ga{};
      ga.x = wa.x; ga.h = wa.h; ga.g = wa.g; ga.dw = grad_of(e, ch.cell->kernel);
      ga.R = wa.R; ga.H = H; ga.W = W; ga.Cx = ch.Cx; ga.C = C; ga.ksize = k;
      mv::launch_convlstm_generic_wgrad(ga, e->stream);
    }, -1.0, 0.0);
    launch(e, "bias_colsum", 0, cells * 4.0 * C * 4.0, [&] {
      run_colsum(e, ch.gates.p, (size_t)cells, (size_t)4 * C, grad_of(e, ch.cell->biases),
                 t.partial.p);
    });
    return;
  }
  const size_t ncols = (size_t)9 * (ch.Cx + C) * 4 * C;
  const bool f16 = e->compute_mode != 0 && mv::wgrad16_ok(W, C);
  bool wino_form = false;                  // f16x3: the row-triple form (15 partial taps)
  int nsplit_x = 0;                        // ... and the split count of its x rows
  size_t bias_blocks = (size_t)(((long long)cells + 63) / 64);
  if (f16) {
    // both operands as cell-contiguous fp16 plane pairs, then the f16x3 GEMMs
    // (convlstm_wgrad_f16x3.h): h rows, x rows; bias partials fall out of the G pass
    const long long Mtot = (long long)wa.R * H * W;
    const long long Mrow = (Mtot + 63) / 64 * 64;
    const int Cx = ch.Cx;
    // compute mode 2: one fp16 plane per operand (see bf16_bwd_enabled); the lower planes are
    // neither written by the transposes nor read by the GEMMs
    const bool one = e->compute_mode == 2 && bf16_bwd_enabled(e);
    const int npl = one ? 1 : 2;
    MV_REQUIRE((size_t)Mrow <= t.mrow_max, "internal: wgrad plane scratch");
    // Winograd F(3,3) over row triples (convlstm_wgrad_f16x3.h, "the row-triple form"): 5/9 of
    // the MFMAs on operands of 5/3 the size; both planes only (fp32-class mode)
    const bool wino = mv::wgrad16_wino3_ok(H, one);
    const long long Mtot3 = Mtot / 3, Mrow3 = (Mtot3 + 63) / 64 * 64;
    const size_t mrow3_max = (t.mrow_max / 3 + 63) / 64 * 64 + 64;
    const size_t pl_cells = wino ? 5 * mrow3_max : t.mrow_max;   // plane pairs x cells per channel
    t.gt16.alloc((size_t)2 * 4 * C * pl_cells);
    t.bias_part.alloc(t.mrow_max / 64 * 4 * C);
    for (int d = 0; d < 3; ++d) t.at16[d].alloc((size_t)2 * C * pl_cells);
    if (Cx)
      for (int d = 0; d < 3; ++d) t.xt16[d].alloc((size_t)2 * std::max(64, Cx) * pl_cells);
    // x rows: any width up to 64 (narrow transpose), or whole 64-channel column groups
    // (--emb_size 128: the transpose of the h operand, one grid row per group)
    MV_REQUIRE(Cx <= 64 || Cx % 64 == 0, "internal: f16x3 wgrad x operand of %d channels", Cx);
    if (!t.wgrad16_attr) {
      HIP_CHECK(hipFuncSetAttribute(
          reinterpret_cast<const void*>(mv::convlstm_wgrad_f16x3_kernel<false, 3>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)mv::kWg16LdsBytes));
      HIP_CHECK(hipFuncSetAttribute(
          reinterpret_cast<const void*>(mv::convlstm_wgrad_f16x3_kernel<true, 3>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)mv::kWg16LdsBytes));
      t.wgrad16_attr = true;
    }

    if (wino)
      launch(e, "wgrad_transpose", 0,
             cells * (4.0 * C * (4.0 + 4.0 * 5 / 3) + (C + Cx) * (4.0 + 3.0 * 4.0 * 5 / 3)), [&] {
        hipLaunchKernelGGL(mv::chain_exp_kernel, dim3(1), dim3(64), 0, e->stream,
                           t.gmax.p + (size_t)gslot * 64, Tsteps, 64, t.chain_exp.p);
        hipLaunchKernelGGL(mv::wino3_transpose_g_kernel, dim3((unsigned)(Mrow3 / 64), 4 * C / 32),
                           dim3(256), 0, e->stream, ch.gates.p, t.gt16.p, Mtot3, 4 * C, Mrow3, W,
                           t.chain_exp.p, t.bias_part.p, (long long)2 * 4 * C * Mrow3, npl);
        hipLaunchKernelGGL(mv::wino3_transpose_a3_kernel, dim3((unsigned)(Mrow3 / 64), C / 32),
                           dim3(256), 0, e->stream, hin, t.at16[0].p, t.at16[1].p, t.at16[2].p,
                           Mtot3, C, Mrow3, H, W, (const int32_t*)nullptr, 8,
                           (long long)2 * C * Mrow3, npl);
        if (Cx) {   // x operand: exponent from max |x| of the chain
          HIP_CHECK(hipMemsetAsync(t.chain_exp.p + 64, 0, 64 * sizeof(int32_t), e->stream));
          hipLaunchKernelGGL(mv::absmax_bits_kernel, dim3(256), dim3(256), 0, e->stream,
                             ch.xs.p, (size_t)Mtot * Cx, t.chain_exp.p + 64);
          // |V| <= 6 max|x|: three more bits of headroom than the direct form's 2^13
          hipLaunchKernelGGL(mv::chain_exp_kernel, dim3(1), dim3(64), 0, e->stream,
                             t.chain_exp.p + 64, 1, 64, t.chain_exp.p + 2, 10);
          hipLaunchKernelGGL(mv::wino3_transpose_a3_kernel,
                             dim3((unsigned)(Mrow3 / 64), (unsigned)((Cx + 31) / 32)), dim3(256), 0,
                             e->stream, ch.xs.p, t.xt16[0].p, t.xt16[1].p, t.xt16[2].p, Mtot3, Cx,
                             Mrow3, H, W, t.chain_exp.p + 2, 0, (long long)2 * Cx * Mrow3, npl);
        }
      });
    else
    launch(e, "wgrad_transpose", 0, cells * (4.0 * C + (C + Cx) * 3.0) * (4.0 + 2.0 * npl), [&] {
      hipLaunchKernelGGL(mv::chain_exp_kernel, dim3(1), dim3(64), 0, e->stream,
                         t.gmax.p + (size_t)gslot * 64, Tsteps, 64, t.chain_exp.p);
      hipLaunchKernelGGL(mv::transpose_split_kernel, dim3((unsigned)(Mrow / 64), 4 * C / 64),
                         dim3(256), 0, e->stream, ch.gates.p, t.gt16.p, Mtot, 4 * C, Mrow, W,
                         0, t.chain_exp.p, 0, t.bias_part.p, npl);
      static const bool fused3 = !(getenv("MV_TRANSPOSE3") && atoi(getenv("MV_TRANSPOSE3")) == 0);
      if (fused3)     // the three column-shifted copies of h from one read
        hipLaunchKernelGGL(mv::transpose_split3_kernel, dim3((unsigned)(Mrow / 64), C / 64),
                           dim3(256), 0, e->stream, hin, t.at16[0].p, t.at16[1].p, t.at16[2].p,
                           Mtot, C, Mrow, W, (const int32_t*)nullptr, 8, npl);
      else
      for (int d = 0; d < 3; ++d)
        hipLaunchKernelGGL(mv::transpose_split_kernel, dim3((unsigned)(Mrow / 64), C / 64),
                           dim3(256), 0, e->stream, hin, t.at16[d].p, Mtot, C, Mrow, W,
                           d - 1, (const int32_t*)nullptr, 8, (float*)nullptr, npl);
      if (Cx) {   // x operand: exponent from max |x| of the chain
        HIP_CHECK(hipMemsetAsync(t.chain_exp.p + 64, 0, 64 * sizeof(int32_t), e->stream));
        hipLaunchKernelGGL(mv::absmax_bits_kernel, dim3(256), dim3(256), 0, e->stream,
                           ch.xs.p, (size_t)Mtot * Cx, t.chain_exp.p + 64);
        hipLaunchKernelGGL(mv::chain_exp_kernel, dim3(1), dim3(64), 0, e->stream,
                           t.chain_exp.p + 64, 1, 64, t.chain_exp.p + 2);
      }
      if (Cx && Cx % 64 == 0 && fused3)
        hipLaunchKernelGGL(mv::transpose_split3_kernel, dim3((unsigned)(Mrow / 64), Cx / 64),
                           dim3(256), 0, e->stream, ch.xs.p, t.xt16[0].p, t.xt16[1].p,
                           t.xt16[2].p, Mtot, Cx, Mrow, W, t.chain_exp.p + 2, 0, npl);
      else
      for (int d = 0; d < 3 && Cx; ++d) {
        if (Cx % 64 == 0)
          hipLaunchKernelGGL(mv::transpose_split_kernel, dim3((unsigned)(Mrow / 64), Cx / 64),
                             dim3(256), 0, e->stream, ch.xs.p, t.xt16[d].p, Mtot, Cx, Mrow, W,
                             d - 1, t.chain_exp.p + 2, 0, (float*)nullptr, npl);
        else
          hipLaunchKernelGGL(mv::transpose_split_narrow_kernel, dim3((unsigned)(Mrow / 64)),
                             dim3(256), 0, e->stream, ch.xs.p, t.xt16[d].p, Mtot, Cx, Mrow, W,
                             d - 1, t.chain_exp.p + 2, npl);
      }
    });
    mv::Wgrad16Args q{};
    for (int d = 0; d < 3; ++d) q.at[d] = t.at16[d].p;
    q.gt = t.gt16.p; q.partial = t.partial.p; q.g_exp = t.chain_exp.p;
    q.a_exp = t.chain_exp.p + 1;
    q.Mrow = Mrow; q.H = H; q.W = W; q.Cx = Cx; q.C = C; q.Ca = C;
    if (wino) {
      q.ntaps = 15; q.Mrow = Mrow3; q.H = H / 3;
      q.a_comp_stride = (int64_t)2 * C * Mrow3; q.g_comp_stride = (int64_t)2 * 4 * C * Mrow3;
    }
    const long long Mgemm = wino ? Mtot3 : Mtot;          // cells of the GEMMs' reduction
    wino_form = wino;
    if (wino) bias_blocks = (size_t)(Mrow3 / 64);
    // the wide tile (convlstm_wgrad_f16x3.h) balances on 7 / 14 splits; the x rows and the
    // reduction follow its count
    const bool wide = mv::wgrad16_wide_ok(W, C);
    if (wide) {
      q.map_mode = mv::wgrad16_wide_map_mode(C);
      wa.nsplit = mv::wgrad16_wide_splits(Mgemm, wa.nsplit, q.map_mode);
    }
    mv::wgrad16_plan(q, Mgemm, wa.nsplit);
    launch(e, "convlstm_wgrad", 2.0 * cells * 9 * C * 4.0 * C, cells * 5.0 * C * 4.0, [&] {
      if (wide && one)
        hipLaunchKernelGGL(mv::convlstm_wgrad_f16x3_wide_kernel<1>,
                           dim3(mv::wgrad16_wide_blocks(q)), dim3(256), 0, e->stream, q);
      else if (wide)
        hipLaunchKernelGGL(mv::convlstm_wgrad_f16x3_wide_kernel<3>,
                           dim3(mv::wgrad16_wide_blocks(q)), dim3(256), 0, e->stream, q);
      else if (one)
        hipLaunchKernelGGL((mv::convlstm_wgrad_f16x3_kernel<false, 1>),
                           dim3(mv::wgrad16_blocks(q, false)), dim3(256), mv::kWg16LdsBytes1,
                           e->stream, q);
      else
      hipLaunchKernelGGL((mv::convlstm_wgrad_f16x3_kernel<false, 3>),
                         dim3(mv::wgrad16_blocks(q, false)), dim3(256), mv::kWg16LdsBytes,
                         e->stream, q);
    }, -1.0, (one ? 1.0 : 3.0) * (wino ? 5.0 / 9.0 : 1.0));
    if (Cx > 0) {
      mv::Wgrad16Args qx = q;
      for (int d = 0; d < 3; ++d) qx.at[d] = t.xt16[d].p;
      qx.Ca = Cx; qx.a_exp = t.chain_exp.p + 2;
      if (wino) {
        qx.a_comp_stride = (int64_t)2 * Cx * Mrow3;
        nsplit_x = mv::wgrad16_x_splits(Mgemm, wa.nsplit);
        mv::wgrad16_plan(qx, Mgemm, nsplit_x);
      }
      launch(e, "convlstm_wgrad_x", 2.0 * cells * 9 * Cx * 4.0 * C,
             cells * (Cx + 4.0 * C) * 4.0, [&] {
        if (wide && mv::wgrad16_wide_x_enabled() && one)
          hipLaunchKernelGGL((mv::convlstm_wgrad_f16x3_wide_kernel<1, true>),
                             dim3(mv::wgrad16_wide_blocks(qx, true)), dim3(256), 0, e->stream, qx);
        else if (wide && mv::wgrad16_wide_x_enabled())
          hipLaunchKernelGGL((mv::convlstm_wgrad_f16x3_wide_kernel<3, true>),
                             dim3(mv::wgrad16_wide_blocks(qx, true)), dim3(256), 0, e->stream, qx);
        else if (one)
          hipLaunchKernelGGL((mv::convlstm_wgrad_f16x3_kernel<true, 1>),
                             dim3(mv::wgrad16_blocks(qx, true)), dim3(256), mv::kWg16LdsBytes1,
                             e->stream, qx);
        else
        hipLaunchKernelGGL((mv::convlstm_wgrad_f16x3_kernel<true, 3>),
                           dim3(mv::wgrad16_blocks(qx, true)), dim3(256), mv::kWg16LdsBytes,
                           e->stream, qx);
      }, -1.0, (one ? 1.0 : 3.0) * (wino ? 5.0 / 9.0 : 1.0));
    }
  } else {
  launch(e, "convlstm_wgrad", 2.0 * cells * 9 * (ch.Cx + C) * 4.0 * C,
         cells * (ch.Cx + 5.0 * C) * 4.0, [&] {
    mv::launch_convlstm_wgrad(wa, e->stream);
  }, -1.0, 1.0);
  }
  launch(e, "wgrad_reduce", 0, 4.0 * ncols * ((wino_form ? 15.0 / 9 : 1.0) * wa.nsplit + 1), [&] {
    if (wino_form)
      hipLaunchKernelGGL(mv::wgrad_wino3_reduce_kernel, dim3(cdiv(ncols / 3, 256)), dim3(256), 0,
                         e->stream, t.partial.p, wa.nsplit, nsplit_x > 0 ? nsplit_x : wa.nsplit,
                         (size_t)ch.Cx * 4 * C, ncols / 9, grad_of(e, ch.cell->kernel));
    else
    hipLaunchKernelGGL(mv::colsum_kernel, dim3(1, cdiv(ncols, 256)), dim3(256), 0,
                       e->stream, t.partial.p, grad_of(e, ch.cell->kernel),
                       (size_t)wa.nsplit, ncols, (size_t)wa.nsplit);
  });
  // biases: column sums of G
  launch(e, "bias_colsum", 0, cells * 4.0 * C * 4.0, [&] {
    if (f16)
      run_colsum(e, t.bias_part.p, bias_blocks, (size_t)4 * C,
                 grad_of(e, ch.cell->biases), t.partial.p);
    else
      run_colsum(e, ch.gates.p, (size_t)cells, (size_t)4 * C, grad_of(e, ch.cell->biases),
                 t.partial.p);
  });
}

void train_backward(mv_engine* e) {
  const mv_config& c = e->cfg;
  TrainState& t = TS(e);
  const int N = c.batch_size, To = c.obs_len, Tp = e->pred_len, C = c.hidden_size,
            D = c.scene_conv_dim, E = c.emb_size;
  float* dh_a[MV_MAX_SCALES][2];
  float* dh_b[MV_MAX_SCALES][2];
  for (int s = 0; s < c.num_scales; ++s) {
    if (!e->sc[s].use) continue;
    TrainScale& R = t.sc[s];
    const size_t NKC = (size_t)N * e->sc[s].K * C;
    for (int b = 0; b < 2; ++b) {
      dh_a[s][b] = R.dh_a[b].p; dh_b[s][b] = R.dh_b[b].p;
      HIP_CHECK(hipMemsetAsync(R.dh_a[b].p, 0, NKC * sizeof(float), e->stream));
      HIP_CHECK(hipMemsetAsync(R.dc[b].p, 0, NKC * sizeof(float), e->stream));
    }
    if (c.use_gnn)
      HIP_CHECK(hipMemsetAsync(R.dsmean.p, 0, (size_t)N * e->sc[s].K * D * sizeof(float),
                               e->stream));
  }
  const bool f16 = e->compute_mode != 0;
  const int nb = c.use_single_decoder ? 1 : 2;
  if (f16) {
    MV_REQUIRE(Tp <= 32 && To <= 32, "f16x3 training: at most 32 steps per chain");
    HIP_CHECK(hipMemsetAsync(t.gmax.p, 0, t.gmax.n * sizeof(int32_t), e->stream));
  }
  // gmax / gexp slot of (scale s, branch b, step): (2 s + b) * 64 + step (+32: encoder)
  // ---- decoders, t = Tp-1 .. 0
  for (int ts = Tp - 1; ts >= 0; --ts) {
    std::vector<ConvLstmArgs> probs;
    std::vector<double> flops;
    std::vector<TrainChain*> chains;
    std::vector<int> slots;
    for (int s = 0; s < c.num_scales; ++s) {
      ScaleState& S = e->sc[s];
      if (!S.use) continue;
      TrainScale& R = t.sc[s];
      const size_t NK = (size_t)N * S.K, NKC = NK * C;
      const int slot = To + ts;
      // d logits / d out_reg of this step -> d h'  (hidden2grid backward)
      run_small_dgrad(e, R.dlogits.p + (size_t)ts * NK, (size_t)S.K, S.out_cls_W->dev.p,
                      dh_a[s][0], (size_t)S.K * C, N, S.H, S.W, C, 1, true);
      run_small_dgrad(e, R.dreg.p + (size_t)ts * NK * 2, (size_t)S.K * 2,
                      S.out_reg_W->dev.p, dh_a[s][nb - 1], (size_t)S.K * C, N, S.H, S.W, C, 2,
                      true);
      for (int b = 0; b < nb; ++b) {
        float* G = R.dec[b].gates.p + (size_t)ts * 4 * NKC;
        const int gs = (2 * s + b) * 64 + ts;
        run_gate_bwd(e, G, R.cs[b].p + slot * NKC, R.cs[b].p + (slot + 1) * NKC,
                     dh_a[s][b], R.dc[b].p, NK, C, f16 ? t.gmax.p + (size_t)gs * 64 : nullptr,
                     (int)probs.size());
        chains.push_back(&R.dec[b]);
        slots.push_back(gs);
        ConvLstmArgs a;
        mv::convlstm_dgrad_args(a, G, R.dec[b].wdpack.p, dh_b[s][b],
                                R.dec[b].dxs.p + (size_t)ts * NK * E, N, S.H, S.W, E, C,
                                true, true);
        probs.push_back(a);
        flops.push_back(2.0 * NK * 9 * (E + C) * 4.0 * C);
      }
    }
    run_dgrad_group(e, probs, flops, chains, slots);
    for (int s = 0; s < c.num_scales; ++s) {
      ScaleState& S = e->sc[s];
      if (!S.use) continue;
      TrainScale& R = t.sc[s];
      const size_t NK = (size_t)N * S.K, NKC = NK * C;
      const int slot = To + ts;
      // class branch: d(h + GNN(h)) -> d h, d scene_mean
      if (c.use_gnn) {
        const float* hsrc = R.hs[0].p + slot * NKC;
        launch(e, "gnn_bwd", NK * (9.0 * 2 * 3 * (C + D) + 9.0 * 4 * C),
               4.0 * NK * (4.0 * C + 2.0 * D), [&] {
          if (C <= 256) {
          hipLaunchKernelGGL(mv::gnn_bwd_a_kernel<1>, dim3(cdiv(NK, 4)), dim3(256), 0,
                             e->stream, hsrc, S.scene_mean.p, dh_b[s][0], R.gnn_a.p,
                             R.gnn_de.p, R.gnn_n.p, N, S.H, S.W, C, gnn_scene_dim(e));
          hipLaunchKernelGGL(mv::gnn_bwd_b_kernel<1>, dim3(cdiv(NK, 4)), dim3(256), 0,
                             e->stream, hsrc, S.scene_mean.p, dh_b[s][0], R.gnn_a.p,
                             R.gnn_de.p, R.gnn_n.p, dh_a[s][0], R.dsmean.p, N, S.H, S.W,
                             C, gnn_scene_dim(e), 1);
          } else {
          hipLaunchKernelGGL(mv::gnn_bwd_a_kernel<2>, dim3(cdiv(NK, 4)), dim3(256), 0,
                             e->stream, hsrc, S.scene_mean.p, dh_b[s][0], R.gnn_a.p,
                             R.gnn_de.p, R.gnn_n.p, N, S.H, S.W, C, gnn_scene_dim(e));
          hipLaunchKernelGGL(mv::gnn_bwd_b_kernel<2>, dim3(cdiv(NK, 4)), dim3(256), 0,
                             e->stream, hsrc, S.scene_mean.p, dh_b[s][0], R.gnn_a.p,
                             R.gnn_de.p, R.gnn_n.p, dh_a[s][0], R.dsmean.p, N, S.H, S.W,
                             C, gnn_scene_dim(e), 1);
          }
        });
      } else {
        std::swap(dh_a[s][0], dh_b[s][0]);
      }
      std::swap(dh_a[s][1], dh_b[s][1]);
      // decoder input embeddings: d x -> d pre-activation (in place)
      for (int b = 0; b < nb; ++b) {
        float* dx = R.dec[b].dxs.p + (size_t)ts * NK * E;
        const size_t total = NK * E;
        run_dropout(e, dx, total, s, 2 + b, ts);       // backward of the input dropout
        launch(e, "tanh_bwd", 3.0 * total, 12.0 * total, [&] {
          // xs holds the DROPPED input; where the mask is zero d x is zero too, elsewhere
          // y = xs * keep_prob recovers the embedding output for 1 - y^2
          if (t.tc.keep_prob < 1.0f)
            hipLaunchKernelGGL(mv::tanh_bwd_scaled_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                               e->stream, dx, R.dec[b].xs.p + (size_t)ts * NK * E, dx, total,
                               t.tc.keep_prob, e->cfg.activation);
          else
            hipLaunchKernelGGL(mv::tanh_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                               e->stream, dx, R.dec[b].xs.p + (size_t)ts * NK * E, dx, total,
                               e->cfg.activation);
        });
      }
      // regression decoder: the step's input was grid_emb(out_reg[t-1]) -> d out_reg[t-1]
      // (teacher forcing feeds the ground truth instead: no gradient path)
      if (ts > 0 && !t.tc.reg_teacher_forcing && nb == 2)
        run_small_dgrad(e, R.dec[1].dxs.p + (size_t)ts * NK * E, (size_t)S.K * E,
                        S.emb_reg_W->dev.p, R.dreg.p + (size_t)(ts - 1) * NK * 2,
                        (size_t)S.K * 2, N, S.H, S.W, 2, E, true);
      // class decoder fed its own logits (training without --train_w_onehot): the same
      // path into d logits[t-1]
      if (ts > 0 && t.tc.class_feedback == 1)
        run_small_dgrad(e, R.dec[0].dxs.p + (size_t)ts * NK * E, (size_t)S.K * E,
                        S.emb_cls_W->dev.p, R.dlogits.p + (size_t)(ts - 1) * NK,
                        (size_t)S.K, N, S.H, S.W, 1, E, true);
    }
  }
  // ---- encoders, t = To-1 .. 0
  for (int ts = To - 1; ts >= 0; --ts) {
    std::vector<ConvLstmArgs> probs;
    std::vector<double> flops;
    std::vector<TrainChain*> chains;
    std::vector<int> slots;
    for (int s = 0; s < c.num_scales; ++s) {
      ScaleState& S = e->sc[s];
      if (!S.use) continue;
      TrainScale& R = t.sc[s];
      const size_t NK = (size_t)N * S.K, NKC = NK * C;
      for (int b = 0; b < nb; ++b) {
        float* G = R.enc[b].gates.p + (size_t)ts * 4 * NKC;
        const int gs = (2 * s + b) * 64 + 32 + ts;    // encoder steps: 32..
        const bool need_dh = ts > 0, need_dx = (b == 0);
        run_gate_bwd(e, G, R.cs[b].p + ts * NKC, R.cs[b].p + (ts + 1) * NKC, dh_a[s][b],
                     R.dc[b].p, NK, C,
                     f16 ? t.gmax.p + (size_t)gs * 64 : nullptr,
                     (need_dh || need_dx) ? (int)probs.size() : -1);
        if (!need_dh && !need_dx) continue;
        chains.push_back(&R.enc[b]);
        slots.push_back(gs);
        ConvLstmArgs a;
        mv::convlstm_dgrad_args(a, G, R.enc[b].wdpack.p, dh_b[s][b],
                                need_dx ? R.enc[b].dxs.p + (size_t)ts * NK * D : nullptr,
                                N, S.H, S.W, R.enc[b].Cx, C, need_dh, need_dx);
        probs.push_back(a);
        flops.push_back(2.0 * NK * 9 * ((need_dx ? R.enc[b].Cx : 0) + C) * 4.0 * C);
      }
    }
    run_dgrad_group(e, probs, flops, chains, slots);
    for (int s = 0; s < c.num_scales; ++s) {
      if (!e->sc[s].use) continue;
      for (int b = 0; b < 2; ++b) std::swap(dh_a[s][b], dh_b[s][b]);
      run_dropout(e, t.sc[s].enc[0].dxs.p + (size_t)ts * N * e->sc[s].K * D,
                  (size_t)N * e->sc[s].K * D, s, 0, ts);
    }
  }
  // ---- parameter gradients
  bool single_first_done = false;
  for (int s = 0; s < c.num_scales; ++s) {
    ScaleState& S = e->sc[s];
    if (!S.use) continue;
    TrainScale& R = t.sc[s];
    const size_t NK = (size_t)N * S.K, NKC = NK * C;
    for (int b = 0; b < nb; ++b) {
      run_wgrad(e, R.enc[b], R.hs[b].p, To, S.H, S.W, (2 * s + b) * 64 + 32);
      comm_reduce_cell(e, *R.enc[b].cell);       // bucket: overlaps the remaining wgrads
      const float* hin = (b == 0 && c.use_gnn) ? R.hg.p : R.hs[b].p + (size_t)To * NKC;
      run_wgrad(e, R.dec[b], hin, Tp, S.H, S.W, (2 * s + b) * 64);
      comm_reduce_cell(e, *R.dec[b].cell);
    }
    // class-decoder grid_emb: its input maps of all steps, [Tp][N][K] -- slot 0 the
    // one-hot of the last observed cell, slot t the one-hot argmax / the logits of step
    // t-1 / the ground-truth map of step t, by feedback mode
    if (!t.mix_on)     // (label mixup: slot 0 already holds the mixed map of the forward)
      hipLaunchKernelGGL(mv::onehot_map_kernel, dim3(cdiv(NK, 256)), dim3(256), 0, e->stream,
                         S.labels.p + (To - 1), To, R.onehot.p, N, S.K);
    if (Tp > 1) {
      const size_t rest = (size_t)(Tp - 1) * NK;
      if (t.tc.class_feedback == 0)
        hipLaunchKernelGGL(mv::onehot_map_kernel, dim3(cdiv(rest, 256)), dim3(256), 0,
                           e->stream, R.ids.p + N, 1, R.onehot.p + NK, (Tp - 1) * N, S.K);
      else
        HIP_CHECK(hipMemcpyAsync(R.onehot.p + NK,
                                 t.tc.class_feedback == 1 ? R.logits.p : R.gt_cls.p + NK,
                                 rest * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    }
    run_small_wgrad(e, R.onehot.p, R.dec[0].dxs.p, grad_of(e, S.emb_cls_W), Tp * N, S.H,
                    S.W, 1, E);
    run_colsum(e, R.dec[0].dxs.p, (size_t)Tp * NK, E, grad_of(e, S.emb_cls_b), t.partial.p);
    if (nb == 2) {
      if (t.tc.reg_teacher_forcing)     // slot 0 of reg_in: obs_grid_reg[:, -1], as regio's
        HIP_CHECK(hipMemcpyAsync(R.reg_in.p, R.regio.p, NK * 2 * sizeof(float),
                                 hipMemcpyDeviceToDevice, e->stream));
      run_small_wgrad(e, t.tc.reg_teacher_forcing ? R.reg_in.p : R.regio.p, R.dec[1].dxs.p,
                      grad_of(e, S.emb_reg_W), Tp * N, S.H,
                      S.W, 2, E);
      run_colsum(e, R.dec[1].dxs.p, (size_t)Tp * NK, E, grad_of(e, S.emb_reg_b), t.partial.p);
    }
    // hidden2grid
    run_small_wgrad(e, R.hs[0].p + (size_t)(To + 1) * NKC, R.dlogits.p,
                    grad_of(e, S.out_cls_W), Tp * N, S.H, S.W, C, 1);
    if (nb == 2) {
      run_small_wgrad(e, R.hs[1].p + (size_t)(To + 1) * NKC, R.dreg.p,
                      grad_of(e, S.out_reg_W), Tp * N, S.H, S.W, C, 2);
    } else {
      // single decoder: ONE offset kernel for all scales -- the scales' gradients add up
      // (in scale order, so the sum is deterministic)
      const size_t nw = (size_t)9 * C * 2;
      if (!single_first_done) {
        run_small_wgrad(e, R.hs[0].p + (size_t)(To + 1) * NKC, R.dreg.p,
                        grad_of(e, S.out_reg_W), Tp * N, S.H, S.W, C, 2);
        single_first_done = true;
      } else {
        run_small_wgrad(e, R.hs[0].p + (size_t)(To + 1) * NKC, R.dreg.p, t.single_w.p,
                        Tp * N, S.H, S.W, C, 2);
        hipLaunchKernelGGL(mv::add_scaled_kernel, dim3(cdiv(nw, 256)), dim3(256), 0, e->stream,
                           grad_of(e, S.out_reg_W), t.single_w.p, 1.0f, nw);
      }
    }
  }
  // ---- scene stack
  const int U = e->num_frames, L = c.num_scales, k = c.scene_conv_kernel;
  for (int i = L - 1; i >= 0; --i) {
    const int Ho = e->conv_h[i], Wo = e->conv_w[i];
    const int Hi = i == 0 ? c.scene_h : e->conv_h[i - 1];
    const int Wi = i == 0 ? c.scene_w : e->conv_w[i - 1];
    const int Ci = i == 0 ? c.scene_class : D;
    const size_t n = (size_t)U * Ho * Wo * D;
    if (e->sc[i].use) {
      TrainScale& R = t.sc[i];
      launch(e, "scene_grad_gather", 0, 4.0 * n * N, [&] {
        hipLaunchKernelGGL(mv::scene_grad_gather_kernel, dim3(cdiv(n, 256)), dim3(256), 0,
                           e->stream, (c.use_gnn && gnn_scene_dim(e) > 0) ? R.dsmean.p : (const float*)nullptr,
                           R.enc[0].dxs.p, e->obs_scene.p, e->sc[i].labels.p, t.dys[i].p,
                           U, N, To, Ho * Wo, D,
                           t.mix_on ? R.obs_labels2.p : (const int32_t*)nullptr, t.mix_w);
      });
    } else {
      HIP_CHECK(hipMemsetAsync(t.dys[i].p, 0, n * sizeof(float), e->stream));
    }
    if (i + 1 < L) {
      const int Ho2 = e->conv_h[i + 1], Wo2 = e->conv_w[i + 1];
      const int ph = std::max((Ho2 - 1) * 2 + k - Ho, 0), pw = std::max((Wo2 - 1) * 2 + k - Wo, 0);
      launch(e, "scene_conv_dgrad", 2.0 * n * k * k * D / 4, 8.0 * n, [&] {
        hipLaunchKernelGGL(mv::conv_s2_dgrad_kernel, dim3(cdiv(n, 256)), dim3(256), 0,
                           e->stream, t.dpre_sc[i + 1].p, e->scene_W[i + 1]->dev.p,
                           t.dys[i].p, U, Ho, Wo, D, Ho2, Wo2, D, k, ph / 2, pw / 2, 1);
      });
    }
    hipLaunchKernelGGL(mv::tanh_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, e->stream,
                       t.dys[i].p, e->scene_conv[i].p, t.dpre_sc[i].p, n, e->cfg.activation);
    const int ph = std::max((Ho - 1) * 2 + k - Hi, 0), pw = std::max((Wo - 1) * 2 + k - Wi, 0);
    const size_t nw = (size_t)k * k * Ci * D;
    const float* in = i == 0 ? e->scene_feat.p : e->scene_conv[i - 1].p;
    if (i == 0 && t.want_dscene) {     // down to the input: d loss / d scene_feat
      const size_t nin = (size_t)U * Hi * Wi * Ci;
      t.dscene.alloc((size_t)c.batch_size * c.obs_len * Hi * Wi * Ci);
      launch(e, "scene_input_dgrad", 2.0 * nin * k * k * D / 4, 4.0 * (nin + n), [&] {
        hipLaunchKernelGGL(mv::conv_s2_dgrad_kernel, dim3(cdiv(nin, 256)), dim3(256), 0,
                           e->stream, t.dpre_sc[0].p, e->scene_W[0]->dev.p, t.dscene.p, U, Hi,
                           Wi, Ci, Ho, Wo, D, k, ph / 2, pw / 2, 0);
      });
      t.have_dscene = true;
    }
    launch(e, "scene_conv_wgrad", 2.0 * n * k * k * Ci, 4.0 * nw, [&] {
      // output rows (u, oy) in <= 64 slabs, partials folded in slab order
      const int rows = U * Ho, rps = (rows + 63) / 64, nslab = (rows + rps - 1) / rps;
      MV_REQUIRE((size_t)nslab * nw <= t.partial.n, "internal: scene wgrad partial buffer");
      hipLaunchKernelGGL(mv::conv_s2_wgrad_kernel, dim3(cdiv(nw, 256), nslab), dim3(256), 0,
                         e->stream, in, t.dpre_sc[i].p, t.partial.p, U, Hi, Wi, Ci, Ho, Wo,
                         D, k, ph / 2, pw / 2, rps);
      hipLaunchKernelGGL(mv::colsum_kernel, dim3(1, cdiv(nw, 256)), dim3(256), 0, e->stream,
                         t.partial.p, grad_of(e, e->scene_W[i]), (size_t)nslab, nw,
                         (size_t)nslab);
    });
    run_colsum(e, t.dpre_sc[i].p, (size_t)U * Ho * Wo, D, grad_of(e, e->scene_b[i]),
               t.partial.p);
  }
  // ---- weight decay on every */W (wd_cost(".*/W"), code/pred_models.py:1033)
  {
    int nW = 0;
    const int base = 2 * MV_MAX_SCALES;
    for (auto& p : e->params) {
      const std::string& nm = p->name;
      if (nm.size() < 2 || nm.compare(nm.size() - 2, 2, "/W") != 0) continue;
      MV_REQUIRE(base + 1 + nW < 64, "too many W tensors");
      const size_t n = p->elems();
      const size_t nblk = cdiv(n, mv::kWdChunk);
      MV_REQUIRE(nblk <= t.partial.n, "internal: weight-decay partial sums");
      hipLaunchKernelGGL(mv::add_scaled_sumsq_kernel, dim3(nblk), dim3(256), 0, e->stream,
                         grad_of(e, p.get()), p->dev.p, t.tc.wd, n, t.partial.p);
      hipLaunchKernelGGL(mv::reduce_sum_kernel, dim3(1), dim3(256), 0, e->stream, t.partial.p,
                         nblk, t.losses.p + base + 1 + nW, 0.5f * t.tc.wd, 0);
      ++nW;
    }
    hipLaunchKernelGGL(mv::reduce_sum_kernel, dim3(1), dim3(256), 0, e->stream,
                       t.losses.p + base + 1, (size_t)nW, t.losses.p + base, 1.0f, 0);
  }
  comm_reduce_rest_and_join(e);
  HIP_CHECK(hipGetLastError());
}

float train_learning_rate(const TrainState& t) {
  const mv_train_config& c = t.tc;
  double lr = c.init_lr;
  if (c.use_cosine_lr) {
    const double gs = (double)std::min<int64_t>(t.global_step, c.max_steps);
    lr = c.init_lr * 0.5 * (1.0 + cos(M_PI * gs / (double)c.max_steps));
  } else if (c.has_decay) {
    const int64_t p = c.decay_steps > 0 ? t.global_step / c.decay_steps : 0;
    lr = c.init_lr * pow((double)c.learning_rate_decay, (double)p);
  }
  return (float)(lr * c.emb_lr);
}

void train_fwd_bwd(mv_engine* e, const mv_inputs* in, const mv_targets* tg, mv_losses* out) {
  MV_REQUIRE(e->train, "mv_train_init has not been called");
  MV_REQUIRE(e->cfg.beam_size == 1, "training needs a greedy (beam_size 1) engine");
  TrainState& t = TS(e);
  if (in) upload_inputs(e, in);
  if (tg) { upload_targets(e, tg); t.targets_ready = true; }
  MV_REQUIRE(e->inputs_ready && t.targets_ready,
             "no resident inputs / targets (mv_upload_inputs, mv_upload_targets)");
  ensure_params(e);
  if (!e->train_packs_valid) {   // first step / after mv_set_param
    train_pack_all(e);
    e->train_packs_valid = true;
  }
  if (e->comm) { e->comm->buckets_last = 0; e->comm->bytes_last = 0; }
  train_forward(e);
  train_losses(e);
  train_backward(e);
  t.have_grads = true;
  // losses to the host
  float hl[64];
  HIP_CHECK(hipMemcpyAsync(hl, t.losses.p, 64 * sizeof(float), hipMemcpyDeviceToHost,
                           e->stream));
  HIP_CHECK(hipStreamSynchronize(e->stream));
  drain_events(e);
  mv_losses L{};
  int li = 0;
  double total = 0;
  for (int s = 0; s < e->cfg.num_scales; ++s) {
    if (!e->sc[s].use) continue;
    L.pred_grid_loss[li] = hl[li]; L.pred_grid_loss[li + 1] = hl[li + 1];
    total += (double)hl[li] + (double)hl[li + 1];
    li += 2;
  }
  L.num_pred_grid_loss = li;
  L.wd_loss = hl[2 * MV_MAX_SCALES];
  // tf.add_n(losses): fp32 left-to-right
  float acc = 0.f;
  for (int i = 0; i < li; ++i) acc = i == 0 ? L.pred_grid_loss[0] : acc + L.pred_grid_loss[i];
  L.loss = acc + L.wd_loss;
  (void)total;
  t.last = L;
  if (out) *out = L;
}

void train_apply(mv_engine* e, float grad_scale) {
  MV_REQUIRE(e->train, "mv_train_init has not been called");
  TrainState& t = TS(e);
  MV_REQUIRE(t.have_grads, "mv_train_apply before mv_train_forward_backward");
  const float lr = train_learning_rate(t);
  const float clip = t.tc.clip_gradient_norm;
  const int do_clip = t.tc.do_clip;
  // Adam: alpha = lr sqrt(1 - beta2_power) / (1 - beta1_power), float32 like the TF kernel
  const float alpha = lr * sqrtf(1.0f - t.beta2_power) / (1.0f - t.beta1_power);
  for (size_t i = 0; i < e->params.size(); ++i) {
    Param* p = e->params[i].get();
    if (p->no_grad) continue;      // tf.gradients gave None: apply_gradients skips the variable
    const size_t n = p->elems(), o = t.goff[i];
    float *s0 = t.accum.p + o, *s1 = t.accum_update.p + o;
    const float* g = t.grad.p + o;
    const dim3 grid(cdiv(n, 256)), blk(256);
    switch (t.tc.optimizer) {
      case 0:
        hipLaunchKernelGGL(mv::adadelta_kernel, grid, blk, 0, e->stream, p->dev.p, s0, s1, g,
                           grad_scale, clip, do_clip, lr, 0.95f, 1e-8f, n);
        break;
      case 1:
        hipLaunchKernelGGL(mv::momentum_kernel, grid, blk, 0, e->stream, p->dev.p, s0, g,
                           grad_scale, clip, do_clip, lr, 0.9f, n);
        break;
      case 2:
        hipLaunchKernelGGL(mv::adam_kernel, grid, blk, 0, e->stream, p->dev.p, s0, s1, g,
                           grad_scale, clip, do_clip, alpha, 1.0f - 0.9f, 1.0f - 0.999f, 1e-8f,
                           n);
        break;
      default:
        hipLaunchKernelGGL(mv::rmsprop_kernel, grid, blk, 0, e->stream, p->dev.p, s0, s1, g,
                           grad_scale, clip, do_clip, lr, 1.0f - 0.9f, 0.0f, 1e-10f, n);
        break;
    }
  }
  if (t.tc.optimizer == 2) { t.beta1_power *= 0.9f; t.beta2_power *= 0.999f; }
  train_pack_all(e);
  for (int s = 0; s < e->cfg.num_scales; ++s) e->sc[s].wq_valid = e->sc[s].sx_valid = false;
  t.global_step += 1;
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(e->stream));
  if (e->compute_mode == 1 || (e->compute_mode == 2 && e->cfg.activation != 0)) {
    // the re-packed fp16 planes of the UPDATED weights (direct and Winograd forms, forward and
    // dgrad) stayed inside the scaled fp16 range?  (convlstm_f16x3.h g_pack_overflow)
    int ovf = 0;
    HIP_CHECK(hipMemcpyFromSymbol(&ovf, HIP_SYMBOL(mv::g_pack_overflow), sizeof(int)));
    if (ovf) {
      const int zero = 0;
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(mv::g_pack_overflow), &zero, sizeof(int)));
      MV_REQUIRE(false, "f16x3: after optimizer step %lld a ConvLSTM kernel (or a transformed "
                 "kernel plane of its Winograd forms) left the scaled fp16 range (|256 w| >= "
                 "60000): the next step would run on infinities; train this model in compute "
                 "mode f32", (long long)t.global_step);
    }
  }
}

}  // namespace
