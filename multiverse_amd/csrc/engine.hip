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

#include "engine_state.h"
#include "engine_setup.h"
#include "engine_forward.h"
#include "engine_io.h"


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
          cc->wpb.release(); cc->wpbt.release(); cc->wx32u.release(); cc->wpw.release(); cc->wpw3.release();
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

}  // extern "C"

#include "engine_ops.h"
