// rccl_stub.cpp — a stand-in for librccl that moves the collectives of the C ABI between PROCESSES THAT SHARE ONE GPU.
//
// TEST INFRASTRUCTURE. RCCL refuses two ranks on one device, and the build and round-end boxes of this project have one GPU, so
// the world > 1 branches of libykpred (ykpred_gather_bitmap / _compressed with its per-peer header exchange, the in-place
// all-reduces of ykpred_exchange_decisions and of the topology histograms inside ykpred_eval / ykpred_eval_nodes) had only ever
// run with world = 1. This library implements exactly the entry points libykpred resolves — ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclGetErrorString — with their RCCL semantics (device pointers, in-place
// forms, ordering after the work already queued on the stream) over a POSIX shared-memory segment: every rank copies its
// contribution into its slot, a barrier, every rank reads the slots. ykpred_comm_use_library() points the engine at it.
// It is slow on purpose (synchronous, through host memory): what it proves is argument marshalling, layouts and ordering.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace {
constexpr size_t kSlotBytes = (size_t)96 << 20;  // per rank (sparse: only what a test touches is ever backed by memory)
constexpr int kMaxRanks = 8;
struct Header {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> attached;
};
struct Comm {
  int rank = 0, world = 1;
  std::string name;
  char* base = nullptr;
  size_t bytes = 0;
  Header* hdr() const { return (Header*)base; }
  char* slot(int r) const { return base + 4096 + (size_t)r * kSlotBytes; }
};
void barrier(Comm* c) {
  Header* h = c->hdr();
  const int gen = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.fetch_add(1, std::memory_order_release);
  } else {
    while (h->generation.load(std::memory_order_acquire) == gen) usleep(50);
  }
}
size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}
template <class T>
void reduce_into(T* acc, const T* in, size_t n, ncclRedOp_t op) {
  for (size_t i = 0; i < n; ++i) {
    if (op == ncclSum) acc[i] = (T)(acc[i] + in[i]);
    else if (op == ncclMax) acc[i] = in[i] > acc[i] ? in[i] : acc[i];
    else if (op == ncclMin) acc[i] = in[i] < acc[i] ? in[i] : acc[i];
  }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/ykstub-%d-%ld", (int)getpid(), (long)random());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm* c = new Comm();
  c->rank = rank;
  c->world = nranks;
  c->name = std::string(id.internal, strnlen(id.internal, sizeof id.internal));
  c->bytes = 4096 + (size_t)nranks * kSlotBytes;
  const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
    delete c;
    return ncclSystemError;
  }
  c->base = (char*)mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == MAP_FAILED) {
    delete c;
    return ncclSystemError;
  }
  // every rank attaches before anybody uses the segment (a fresh segment is zero-filled: the counters start at 0)
  c->hdr()->attached.fetch_add(1, std::memory_order_acq_rel);
  while (c->hdr()->attached.load(std::memory_order_acquire) < nranks) usleep(50);
  *out = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclSuccess;
  barrier(c);
  if (c->rank == 0) shm_unlink(c->name.c_str());
  munmap(c->base, c->bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  const size_t bytes = count * type_size(type);
  if (!c || type_size(type) == 0 || bytes > kSlotBytes) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;  // ordered after the work queued on the stream
  if (bytes && hipMemcpy(c->slot(c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  barrier(c);
  for (int r = 0; r < c->world && bytes; ++r)
    if (hipMemcpy((char*)recv + (size_t)r * bytes, c->slot(r), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  barrier(c);  // nobody overwrites a slot before everybody has read it
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  const size_t bytes = count * type_size(type);
  if (!c || bytes > kSlotBytes || (op != ncclSum && op != ncclMax && op != ncclMin)) return ncclInvalidArgument;
  if (type != ncclInt32 && type != ncclInt64 && type != ncclUint64) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (bytes && hipMemcpy(c->slot(c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  barrier(c);
  if (bytes) {
    char* acc = new char[bytes];
    memcpy(acc, c->slot(0), bytes);
    for (int r = 1; r < c->world; ++r) {
      if (type == ncclInt32) reduce_into((int32_t*)acc, (const int32_t*)c->slot(r), count, op);
      else if (type == ncclInt64) reduce_into((int64_t*)acc, (const int64_t*)c->slot(r), count, op);
      else reduce_into((uint64_t*)acc, (const uint64_t*)c->slot(r), count, op);
    }
    const hipError_t s = hipMemcpy(recv, acc, bytes, hipMemcpyHostToDevice);
    delete[] acc;
    if (s != hipSuccess) return ncclUnhandledCudaError;
  }
  barrier(c);
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "invalid argument (rccl stub)";
    case ncclSystemError: return "system error (rccl stub: shared memory)";
    case ncclUnhandledCudaError: return "HIP call failed (rccl stub)";
    default: return "error (rccl stub)";
  }
}

}  // extern "C"
