// rccl_stub.cpp -- TEST INFRASTRUCTURE, not a product path.
//
// A stand-in for librccl that exports the nine nccl* symbols libpop_amd's in-library transport binds
// (pop2-cesm_amd/csrc/rccl_transport.hpp), so that pop_comm_init_rccl / rccl_exchange / rccl_allreduce and
// bench.py's "rccl-native" branch can run with SEVERAL ranks on ONE GPU (real RCCL refuses two ranks on one
// device).  Selected with POP_RCCL_LIB=<path to librccl_stub.so>.
//
// Semantics kept from RCCL: calls are stream-ordered (the stub synchronises the stream, then moves the bytes
// through a POSIX shared-memory segment named by the unique id), grouped send/recv pairs complete together,
// the all-reduce returns the same bits on every rank (slots added in rank order).  Every wait has a timeout,
// so a rank that died makes the others fail instead of hanging the GPU box.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kOk = 0, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4;
constexpr int kDouble = 8, kSum = 0;

struct Header {
  std::atomic<uint32_t> ready;       // set by rank 0 of CommInitRank once the segment has its full size
  std::atomic<uint32_t> arrived, generation, attached;
  uint32_t nranks;
  uint64_t slot_bytes, box_bytes, total_bytes;
};
struct Box {                         // one mailbox per (src, dst) pair, capacity box_bytes
  std::atomic<uint64_t> written, read;
  uint64_t bytes;
};

struct Comm {
  char name[64];
  int rank = 0, nranks = 0;
  size_t mapped = 0;
  char *base = nullptr;
  Header *hdr() const { return (Header *)base; }
  char *slot(int r) const { return base + 4096 + (size_t)r * hdr()->slot_bytes; }
  Box *box(int src, int dst) const {
    return (Box *)(base + 4096 + (size_t)nranks * hdr()->slot_bytes + ((size_t)src * nranks + dst) * (sizeof(Box) + hdr()->box_bytes));
  }
};

struct Op { bool send; void *dev; size_t count; int peer; Comm *c; hipStream_t st; bool done; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
double timeout_s() { const char *e = getenv("POP_RCCL_STUB_TIMEOUT"); return e ? atof(e) : 120.0; }
size_t env_mb(const char *n, size_t dflt) { const char *e = getenv(n); return (e ? (size_t)atol(e) : dflt) << 20; }

void id_to_name(const char *id, char *name) { snprintf(name, 64, "/%.60s", id); }

int barrier(Comm *c) {
  Header *h = c->hdr();
  const uint32_t gen = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.store(gen + 1, std::memory_order_release);
    return kOk;
  }
  const double t0 = now();
  while (h->generation.load(std::memory_order_acquire) == gen) {
    if (now() - t0 > timeout_s()) { fprintf(stderr, "rccl_stub: rank %d timed out in a collective\n", c->rank); return kSystemError; }
    usleep(20);
  }
  return kOk;
}

int run_ops(std::vector<Op> &ops) {
  for (Op &o : ops) if (hipStreamSynchronize(o.st) != hipSuccess) return kInternalError;
  const double t0 = now();
  size_t left = ops.size();
  while (left) {
    bool progress = false;
    for (Op &o : ops) {
      if (o.done) continue;
      Comm *c = o.c;
      const size_t bytes = o.count * sizeof(double);
      if (bytes > c->hdr()->box_bytes) { fprintf(stderr, "rccl_stub: message of %zu bytes exceeds the mailbox (POP_RCCL_STUB_BOX_MB)\n", bytes); return kInvalidArgument; }
      if (o.send) {
        Box *b = c->box(c->rank, o.peer);
        if (b->written.load(std::memory_order_acquire) != b->read.load(std::memory_order_acquire)) continue;   // previous message not consumed yet
        if (hipMemcpy((char *)(b + 1), o.dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return kInternalError;
        b->bytes = bytes;
        b->written.fetch_add(1, std::memory_order_release);
      } else {
        Box *b = c->box(o.peer, c->rank);
        if (b->written.load(std::memory_order_acquire) == b->read.load(std::memory_order_acquire)) continue;   // nothing there yet
        if (b->bytes != bytes) { fprintf(stderr, "rccl_stub: rank %d expected %zu bytes from %d, message has %llu\n", c->rank, bytes, o.peer, (unsigned long long)b->bytes); return kInvalidArgument; }
        if (hipMemcpy(o.dev, (char *)(b + 1), bytes, hipMemcpyHostToDevice) != hipSuccess) return kInternalError;
        b->read.fetch_add(1, std::memory_order_release);
      }
      o.done = true; --left; progress = true;
    }
    if (!progress) {
      if (now() - t0 > timeout_s()) { fprintf(stderr, "rccl_stub: send/recv group timed out\n"); return kSystemError; }
      usleep(20);
    }
  }
  return kOk;
}

}  // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId *id) {
  memset(id->internal, 0, 128);
  timespec t; clock_gettime(CLOCK_REALTIME, &t);
  snprintf(id->internal, 128, "pop_rccl_stub_%d_%lld", (int)getpid(), (long long)t.tv_sec * 1000000000LL + t.tv_nsec);
  char name[64]; id_to_name(id->internal, name);
  const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return kSystemError;
  if (ftruncate(fd, 4096) != 0) { close(fd); return kSystemError; }
  close(fd);
  return kOk;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return kInvalidArgument;
  Comm *c = new Comm();
  c->rank = rank; c->nranks = nranks;
  id_to_name(id.internal, c->name);
  const int fd = shm_open(c->name, O_RDWR, 0600);
  if (fd < 0) { delete c; return kSystemError; }
  Header *h0 = (Header *)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (h0 == MAP_FAILED) { close(fd); delete c; return kSystemError; }
  if (rank == 0) {
    h0->nranks = (uint32_t)nranks;
    h0->slot_bytes = env_mb("POP_RCCL_STUB_SLOT_MB", 8);
    h0->box_bytes = env_mb("POP_RCCL_STUB_BOX_MB", 8);
    h0->total_bytes = 4096 + (uint64_t)nranks * h0->slot_bytes + (uint64_t)nranks * nranks * (sizeof(Box) + h0->box_bytes);
    if (ftruncate(fd, (off_t)h0->total_bytes) != 0) { close(fd); delete c; return kSystemError; }
    h0->ready.store(1, std::memory_order_release);
  } else {
    const double t0 = now();
    while (!h0->ready.load(std::memory_order_acquire)) {
      if (now() - t0 > timeout_s()) { close(fd); delete c; return kSystemError; }
      usleep(50);
    }
    if ((int)h0->nranks != nranks) { close(fd); delete c; return kInvalidArgument; }
  }
  c->mapped = (size_t)h0->total_bytes;
  munmap(h0, 4096);
  c->base = (char *)mmap(nullptr, c->mapped, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == MAP_FAILED) { delete c; return kSystemError; }
  c->hdr()->attached.fetch_add(1, std::memory_order_acq_rel);
  *comm = c;
  return barrier(c);     // communicator creation is collective
}

int ncclCommCount(const void *comm, int *count) {
  if (!comm || !count) return kInvalidArgument;
  *count = ((const Comm *)comm)->nranks;
  return kOk;
}
int ncclCommUserRank(const void *comm, int *rank) {
  if (!comm || !rank) return kInvalidArgument;
  *rank = ((const Comm *)comm)->rank;
  return kOk;
}

int ncclCommDestroy(void *comm) {
  Comm *c = (Comm *)comm;
  if (!c) return kOk;
  const bool last = c->hdr()->attached.fetch_sub(1, std::memory_order_acq_rel) == 1;
  munmap(c->base, c->mapped);
  if (last) shm_unlink(c->name);
  delete c;
  return kOk;
}

int ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, int datatype, int op, void *comm, hipStream_t stream) {
  Comm *c = (Comm *)comm;
  if (datatype != kDouble || op != kSum) return kInvalidArgument;
  const size_t bytes = count * sizeof(double);
  if (bytes > c->hdr()->slot_bytes) { fprintf(stderr, "rccl_stub: all-reduce of %zu bytes exceeds the slot (POP_RCCL_STUB_SLOT_MB)\n", bytes); return kInvalidArgument; }
  if (hipStreamSynchronize(stream) != hipSuccess) return kInternalError;
  if (hipMemcpy(c->slot(c->rank), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return kInternalError;
  int rc = barrier(c);
  if (rc) return rc;
  std::vector<double> sum(count, 0.0);
  for (int r = 0; r < c->nranks; ++r) {      // rank order on every rank: identical bits everywhere
    const double *s = (const double *)c->slot(r);
    for (size_t i = 0; i < count; ++i) sum[i] = sum[i] + s[i];
  }
  if (hipMemcpy(recvbuff, sum.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return kInternalError;
  return barrier(c);                         // slots may be overwritten after this
}

int ncclGroupStart() { ++g_depth; return kOk; }
int ncclGroupEnd() {
  if (g_depth <= 0) return kInvalidArgument;
  if (--g_depth > 0) return kOk;
  std::vector<Op> ops; ops.swap(g_ops);
  return run_ops(ops);
}
static int p2p(bool send, void *buf, size_t count, int datatype, int peer, void *comm, hipStream_t stream) {
  Comm *c = (Comm *)comm;
  if (datatype != kDouble || peer < 0 || peer >= c->nranks) return kInvalidArgument;
  g_ops.push_back(Op{send, buf, count, peer, c, stream, false});
  if (g_depth > 0) return kOk;
  std::vector<Op> ops; ops.swap(g_ops);
  return run_ops(ops);
}
int ncclSend(const void *sendbuff, size_t count, int datatype, int peer, void *comm, hipStream_t stream) {
  return p2p(true, (void *)sendbuff, count, datatype, peer, comm, stream);
}
int ncclRecv(void *recvbuff, size_t count, int datatype, int peer, void *comm, hipStream_t stream) {
  return p2p(false, recvbuff, count, datatype, peer, comm, stream);
}
const char *ncclGetErrorString(int rc) {
  switch (rc) {
    case kOk: return "no error (rccl stub)";
    case kSystemError: return "system error / timeout (rccl stub)";
    case kInternalError: return "HIP call failed (rccl stub)";
    case kInvalidArgument: return "invalid argument (rccl stub)";
    default: return "unknown error (rccl stub)";
  }
}

}  // extern "C"
