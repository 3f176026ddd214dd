// host <-> device: input upload (dense / compact), output download, the pipelined feed / fetch -- part of the ONE translation unit engine.hip (included from there, in order;
// not a stand-alone header).
#pragma once

namespace {

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
