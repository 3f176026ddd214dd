// engine set-up: parameter table, config validation, buffers, weight packs -- part of the ONE translation unit engine.hip (included from there, in order;
// not a stand-alone header).
#pragma once

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

// bf16 mode on the row-triple tile (convlstm_wino3.h BF16D): the pack of the CURRENT device
// weights, or none (tanh models only -- unbounded activations keep the three-pass x path of the
// 32-cell body; MV_BF16T=0: A/B runs)
void pack_bf16t(mv_engine* e, ConvCell& cc) {
  const int C = e->cfg.hidden_size;
  const bool small = cc.Cx > 0 && 9 * cc.Cx <= mv::kBK;
  const int Cx16 = small ? 0 : cc.Cx;
  if (!(mv::bf16t_enabled() && C_multiple_ok(e, cc) && e->cfg.activation == 0)) {
    cc.wpbt.release();
    return;
  }
  const size_t halves = mv::bf16t_wpack_elems(Cx16, C, mv::kW3Nrb);
  cc.wpbt.alloc(halves);
  hipLaunchKernelGGL(mv::pack_bf16t_kernel, dim3(cdiv(halves, 256)), dim3(256), 0, e->stream,
                     cc.kernel->dev.p, cc.wpbt.p, cc.Cx, Cx16, C, mv::kW3Nrb, halves);
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
  const bool xf16 = e->cfg.activation != 0 && !small;     // engine_state.h dyn_x
  std::vector<_Float16> pb(mv::bf16_wpack_elems(Cx16, C, xf16));
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
    if (xf16) {        // the x rows travel as fp16 of 256 w there: same range bound as f16x3
      const int Cin = cc.Cx + C, N4 = 4 * C;
      float mx = 0.f;
      for (int t = 0; t < 9; ++t)
        for (int ci = 0; ci < cc.Cx; ++ci)
          for (int n = 0; n < N4; ++n)
            mx = std::max(mx, std::fabs(cc.kernel->host[((size_t)t * Cin + ci) * N4 + n]));
      MV_REQUIRE(mx * mv::kF16Scale < 60000.f, "bf16 mode, unbounded activations: the x rows of "
                 "%s reach %g, outside the scaled fp16 range; use compute mode f32",
                 cc.kernel->name.c_str(), mx);
    }
    mv::pack_bf16_weights(cc.kernel->host.data(), cc.Cx, C, pb.data(), xf16);
  }
  cc.wpb.alloc(pb.size());
  HIP_CHECK(hipMemcpy(cc.wpb.p, pb.data(), pb.size() * sizeof(_Float16),
                      hipMemcpyHostToDevice));
  pack_bf16t(e, cc);
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

}  // namespace
