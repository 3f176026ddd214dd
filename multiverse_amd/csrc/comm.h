// Gradient all-reduce of the data-parallel training step, inside the library
// (SURVEY.md 8b `mv_allreduce_init`, 8e): one RCCL communicator per engine, the flat
// gradient buffer reduced in BUCKETS on a side stream while the backward pass is still
// producing the later buckets -- one bucket per ConvLSTM kernel (+ its biases; 9.5-11.8 MB
// each, 8 of them), issued right after that kernel's wgrad reduction, then the small
// tensors (scene convs, embeddings, hidden2grid) as one group after the weight-decay
// pass.  The reference has no multi-GPU path at all (code/train.py:35); the only
// cross-sample coupling of the step is the batch mean in the loss
// (code/pred_models.py:995, 1016-1022), hence all-reduce(sum) then 1/world in
// mv_train_apply, with the element-wise clip AFTER the reduction.
//
// RCCL is resolved at run time (dlopen): a process that already carries an RCCL
// (PyTorch ships its own librccl.so) must not get a second copy, and a single-GPU user
// needs none at all.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and enums only; entry points come from dlsym

#include <stdlib.h>

#include <string>

namespace mv {

struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string where, error;
  bool ok = false;
};

inline RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    void* h = nullptr;
    // an RCCL already in the process first (NOLOAD matches by SONAME), then ROCm's
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                           "/opt/rocm/lib/librccl.so"};
#ifdef MV_TEST_HOOKS
    // MV_RCCL_LIB=<path> -- ONLY in -DMV_TEST_HOOKS builds (tests/fake_rccl/
    // libmultiverse_hip_testhooks.so): exactly that library and no other.  The single-GPU
    // tests put a shared-memory stand-in there (tests/fake_rccl) so that the bucketed
    // all-reduce can run with two ranks on one device -- real RCCL refuses duplicate devices.
    // The shipped library does not read the variable: no environment setting can redirect the
    // gradient all-reduce of a production job to another shared object.
    if (const char* forced = getenv("MV_RCCL_LIB")) {
      h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
      if (!h) { a.error = std::string("MV_RCCL_LIB: ") + dlerror(); return a; }
      a.where = std::string(forced) + " (MV_RCCL_LIB)";
      fprintf(stderr, "[multiverse_hip] TEST-HOOKS BUILD: collectives bound to %s instead of "
              "librccl\n", forced);
    }
#endif
    if (!h)
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) { a.where = std::string(n) + " (already loaded)"; break; }
    if (!h)
      for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) { a.where = n; break; }
    if (!h) { a.error = "librccl.so not found (dlopen)"; return a; }
#define MV_RCCL_SYM(field, name)                                              \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));            \
    if (!a.field) { a.error = std::string("RCCL symbol missing: ") + name; return a; }
    MV_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    MV_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    MV_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    MV_RCCL_SYM(AllReduce, "ncclAllReduce")
    MV_RCCL_SYM(GroupStart, "ncclGroupStart")
    MV_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    MV_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef MV_RCCL_SYM
    a.ok = true;
    return a;
  }();
  return api;
}

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr;       // the side stream the collectives run on
  hipEvent_t ready = nullptr;         // main stream -> side stream: a bucket is complete
  hipEvent_t done = nullptr;          // side stream -> main stream: all buckets reduced
  int buckets_last = 0;               // collectives of the last step (reporting)
  double bytes_last = 0;
};

}  // namespace mv
