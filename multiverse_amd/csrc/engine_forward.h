// the forward pass: grouped gate launches, scene stack, encoders, graph attention, decode tail, greedy / beam decoders -- part of the ONE translation unit engine.hip (included from there, in order;
// not a stand-alone header).
#pragma once

namespace {

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
    // bf16 mode, unbounded activations: three x passes (convlstm_f16x3.h xpasses)
    const int xpass = (bf16 && e->dyn_x() && !a.x_small) ? 3 : 1;
    q.n_xk = a.x_small ? 0 : xpass * mv::f16x3_xksteps(a.Cx);
    q.n_hk = a.zero_state ? 0 : 9 * (a.C / 16);
    q.w_ksteps = xpass * mv::f16x3_xksteps(a.Cx) + 9 * (a.C / 16);
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
  // bf16 mode: the row-triple tile (convlstm_wino3.h BF16D) when every problem fits it
  bool bf16t = bf16 && mv::bf16t_enabled();
  for (size_t i = 0; i < p16.size() && bf16t; ++i) {
    ConvCell* cc = cell_of_pack(e, probs[i].wpack);
    if (!mv::bf16t_geometry_ok(p16[i].f, p16[i]) || !cc->wpbt.p) bf16t = false;
  }
  if (bf16t)
    for (size_t i = 0; i < p16.size(); ++i) {
      pw[i].b = p16[i];
      pw[i].wpw = cell_of_pack(e, probs[i].wpack)->wpbt.p;
      pw[i].v3x = pw[i].v3h = nullptr;
    }
  launch(e, "convlstm_step", flops, bytes, [&] {
    if (bf16t)
      mv::launch_convlstm_bf16t_steps(pw.data(), (int)pw.size(), e->stream);
    else if (e->compute_mode == 2)
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

}  // namespace
