// TEST INFRASTRUCTURE -- not part of the product, never loaded unless MV_RCCL_LIB names it.
//
// A stand-in for librccl that lets TWO (or more) ranks share ONE GPU, so that the engine's
// in-library bucketed gradient all-reduce (multiverse_amd/csrc/comm.h, engine_train.h
// comm_reduce_*) can execute with world > 1 on a single-GPU test box: real RCCL refuses two
// ranks on one device.  It exports the seven entry points comm.h resolves by dlsym
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclGroupStart,
// ncclGroupEnd, ncclGetErrorString) with RCCL's own signatures.
//
// Ranks are processes of one machine; the "fabric" is a POSIX shared-memory segment named
// after the unique id.  ncclAllReduce(float, sum): wait for the stream's earlier work,
// copy the send buffer to this rank's slot (chunks of kSlotFloats), barrier, add the slots
// in RANK ORDER on the host (every rank computes the same bits), barrier, copy the sum back
// on the caller's stream.  Blocking on the host is within the collective's contract; the
// stream / event protocol of the caller is honoured (the call synchronises the stream it is
// given, nothing else).  Every barrier has a timeout, so a protocol bug in the caller shows
// up as ncclSystemError, not as a hung test.  MV_FAKE_RCCL_LOG=<file>: one line per
// collective (rank, count, group depth) for the test to read.
//
// MV_FAKE_RCCL_ASYNC=1: the ASYNCHRONOUS mode -- ncclAllReduce returns at once, like RCCL's:
// it only ENQUEUES on the caller's stream (a) a device -> pinned-host copy of the send buffer,
// (b) a host function (hipLaunchHostFunc) that performs the exchange when the stream reaches
// it -- rendezvous with the other ranks, sum in rank order, an optional extra delay
// (MV_FAKE_RCCL_DELAY_MS) -- and (c) the pinned-host -> device copy of the sum.  Nothing is
// synchronised on the caller's behalf, so a missing event wait on either side of the
// collective (the main stream's `ready` in front of it, the side stream's `done` behind it)
// reads or applies unreduced gradients and the test FAILS; the synchronous mode above cannot
// see such bugs.  Staging buffers come from a ring of pinned allocations; ring slots are
// reused in stream order (one stream per communicator, as comm.h uses it).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr size_t kSlotFloats = (size_t)4 << 20;       // 16 MB per rank and chunk
constexpr int kMaxWorld = 8;
constexpr double kTimeoutS = 120.0;

struct Control {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> attached;
  std::atomic<int> failed;
};

struct FakeComm {
  int rank = 0, world = 1;
  std::string name;
  size_t bytes = 0;
  void* base = nullptr;
  Control* ctl = nullptr;
  float* slots = nullptr;
  std::vector<float> sum;
  int group_depth = 0;
  long collectives = 0;
  // asynchronous mode
  static constexpr int kRing = 4;
  float* stage[kRing] = {};          // pinned: send copy in, sum out
  size_t stage_cap[kRing] = {};
  hipStream_t stream_seen = nullptr;
  std::atomic<int> async_failed{0};
};

struct AsyncJob {
  FakeComm* c;
  float* buf;
  size_t count;
};

thread_local int g_group_depth = 0;

std::string shm_name(const ncclUniqueId& id) {
  char buf[64];
  unsigned long long a = 0, b = 0;
  memcpy(&a, id.internal, 8);
  memcpy(&b, id.internal + 8, 8);
  snprintf(buf, sizeof(buf), "/mvfakerccl_%016llx%016llx", a, b);
  return buf;
}

bool barrier(FakeComm* c) {
  Control* k = c->ctl;
  const int gen = k->generation.load(std::memory_order_acquire);
  if (k->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
    k->arrived.store(0, std::memory_order_relaxed);
    k->generation.fetch_add(1, std::memory_order_release);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (k->generation.load(std::memory_order_acquire) == gen) {
    if (k->failed.load(std::memory_order_relaxed)) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (dt > kTimeoutS) {
      k->failed.store(1);
      fprintf(stderr, "[fake rccl] rank %d: barrier timeout (a rank issued a different "
              "sequence of collectives?)\n", c->rank);
      return false;
    }
  }
  return true;
}

void log_call(const FakeComm* c, size_t count) {
  const char* path = getenv("MV_FAKE_RCCL_LOG");
  if (!path) return;
  if (FILE* f = fopen(path, "a")) {
    fprintf(f, "allreduce rank %d world %d count %zu group %d\n", c->rank, c->world, count,
            g_group_depth);
    fclose(f);
  }
}

bool async_mode() {
  static const bool on = getenv("MV_FAKE_RCCL_ASYNC") && atoi(getenv("MV_FAKE_RCCL_ASYNC")) == 1;
  return on;
}

// runs on a runtime thread when the caller's stream reaches it; no HIP calls in here
void exchange_on_host(void* user) {
  AsyncJob* job = static_cast<AsyncJob*>(user);
  FakeComm* c = job->c;
  static const int delay_ms = getenv("MV_FAKE_RCCL_DELAY_MS") ? atoi(getenv("MV_FAKE_RCCL_DELAY_MS")) : 0;
  for (size_t off = 0; off < job->count && !c->async_failed.load(); off += kSlotFloats) {
    const size_t n = std::min(kSlotFloats, job->count - off);
    float* mine = c->slots + (size_t)c->rank * kSlotFloats;
    memcpy(mine, job->buf + off, n * sizeof(float));
    if (!barrier(c)) { c->async_failed.store(1); break; }
    float* out = job->buf + off;
    for (size_t i = 0; i < n; ++i) out[i] = c->slots[i];
    for (int r = 1; r < c->world; ++r) {
      const float* s = c->slots + (size_t)r * kSlotFloats;
      for (size_t i = 0; i < n; ++i) out[i] += s[i];
    }
    if (!barrier(c)) { c->async_failed.store(1); break; }
  }
  if (delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
  delete job;
}

ncclResult_t allreduce_async(FakeComm* c, const float* src, float* dst, size_t count,
                             hipStream_t stream) {
  if (c->async_failed.load()) return ncclSystemError;
  if (c->stream_seen && c->stream_seen != stream) return ncclInvalidUsage;   // ring = stream order
  c->stream_seen = stream;
  const int slot = (int)((c->collectives - 1) % FakeComm::kRing);
  if (c->stage_cap[slot] < count) {
    // growing a slot: its previous user (kRing collectives ago, same stream) must be done
    if (c->stage[slot]) {
      if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
      (void)hipHostFree(c->stage[slot]);
    }
    // at least one 16 MB chunk: no bucket of the engine's model ever grows a slot again, so
    // the synchronisation above never hides a protocol bug after the first kRing collectives
    const size_t cap = std::max(count, kSlotFloats);
    if (hipHostMalloc(reinterpret_cast<void**>(&c->stage[slot]), cap * sizeof(float),
                      hipHostMallocDefault) != hipSuccess)
      return ncclUnhandledCudaError;
    c->stage_cap[slot] = cap;
  }
  float* buf = c->stage[slot];
  if (hipMemcpyAsync(buf, src, count * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess)
    return ncclUnhandledCudaError;
  // MV_FAKE_RCCL_POISON=1: between the copy-out and the copy-back the receive buffer holds
  // NaNs (0xFF bytes), so that a caller whose main stream does not wait for the collective's
  // completion reads poison WHATEVER the timing -- the deterministic form of the dropped-
  // `done`-wait negative control (the 20 ms delay alone made it a race)
  static const bool poison = getenv("MV_FAKE_RCCL_POISON") && atoi(getenv("MV_FAKE_RCCL_POISON")) == 1;
  if (poison && hipMemsetAsync(dst, 0xFF, count * sizeof(float), stream) != hipSuccess)
    return ncclUnhandledCudaError;
  AsyncJob* job = new AsyncJob{c, buf, count};
  if (hipLaunchHostFunc(stream, exchange_on_host, job) != hipSuccess) {
    delete job;
    return ncclUnhandledCudaError;
  }
  if (hipMemcpyAsync(dst, buf, count * sizeof(float), hipMemcpyHostToDevice, stream) != hipSuccess)
    return ncclUnhandledCudaError;
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof(*id));
  std::random_device rd;
  for (int i = 0; i < 4; ++i) {
    const unsigned v = rd();
    memcpy(id->internal + 4 * i, &v, 4);
  }
  const int pid = (int)getpid();
  memcpy(id->internal + 16, &pid, sizeof(pid));
  memcpy(id->internal + 24, "mvfake", 6);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > kMaxWorld || rank < 0 || rank >= nranks)
    return ncclInvalidArgument;
  if (memcmp(id.internal + 24, "mvfake", 6) != 0) return ncclInvalidArgument;
  FakeComm* c = new FakeComm();
  c->rank = rank; c->world = nranks;
  c->name = shm_name(id);
  c->bytes = 4096 + (size_t)nranks * kSlotFloats * sizeof(float);
  const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0) { delete c; return ncclSystemError; }
  if (ftruncate(fd, (off_t)c->bytes) != 0) { close(fd); delete c; return ncclSystemError; }
  c->base = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == MAP_FAILED) { delete c; return ncclSystemError; }
  c->ctl = reinterpret_cast<Control*>(c->base);        // a fresh segment is zero-filled
  c->slots = reinterpret_cast<float*>(static_cast<char*>(c->base) + 4096);
  c->sum.resize(kSlotFloats);
  c->ctl->attached.fetch_add(1);
  // rendezvous: every rank has mapped the segment before anyone uses it
  const auto t0 = std::chrono::steady_clock::now();
  while (c->ctl->attached.load() < nranks) {
    std::this_thread::sleep_for(std::chrono::microseconds(200));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) {
      fprintf(stderr, "[fake rccl] rank %d: %d of %d ranks attached\n", rank,
              c->ctl->attached.load(), nranks);
      return ncclSystemError;
    }
  }
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (!c) return ncclSuccess;
  if (c->stream_seen) (void)hipStreamSynchronize(c->stream_seen);   // pending host functions
  for (int i = 0; i < FakeComm::kRing; ++i)
    if (c->stage[i]) (void)hipHostFree(c->stage[i]);
  if (c->base) munmap(c->base, c->bytes);
  shm_unlink(c->name.c_str());          // the first rank to leave removes the name
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { ++g_group_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) {
  if (g_group_depth <= 0) return ncclInvalidUsage;
  --g_group_depth;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count,
                           ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  FakeComm* c = reinterpret_cast<FakeComm*>(comm);
  if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
  if (datatype != ncclFloat || op != ncclSum) return ncclInvalidArgument;
  log_call(c, count);
  c->collectives += 1;
  const float* src = static_cast<const float*>(sendbuff);
  float* dst = static_cast<float*>(recvbuff);
  if (async_mode()) return allreduce_async(c, src, dst, count, stream);
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  for (size_t off = 0; off < count; off += kSlotFloats) {
    const size_t n = std::min(kSlotFloats, count - off);
    float* mine = c->slots + (size_t)c->rank * kSlotFloats;
    if (hipMemcpy(mine, src + off, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
      return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    for (size_t i = 0; i < n; ++i) c->sum[i] = c->slots[i];
    for (int r = 1; r < c->world; ++r) {
      const float* s = c->slots + (size_t)r * kSlotFloats;
      for (size_t i = 0; i < n; ++i) c->sum[i] += s[i];
    }
    if (!barrier(c)) return ncclSystemError;       // nobody refills a slot still being read
    if (hipMemcpyAsync(dst + off, c->sum.data(), n * sizeof(float), hipMemcpyHostToDevice,
                       stream) != hipSuccess)
      return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t result) {
  switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake rccl: HIP call failed";
    case ncclSystemError: return "fake rccl: shared-memory rendezvous failed / timed out";
    case ncclInvalidArgument: return "fake rccl: invalid argument (float + sum only)";
    case ncclInvalidUsage: return "fake rccl: invalid usage";
    default: return "fake rccl: error";
  }
}

}  // extern "C"
