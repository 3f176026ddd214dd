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
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  const float* src = static_cast<const float*>(sendbuff);
  float* dst = static_cast<float*>(recvbuff);
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
