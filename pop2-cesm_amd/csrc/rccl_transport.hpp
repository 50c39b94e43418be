// rccl_transport.hpp -- in-library transport for multi-GPU runs: halo messages and the block-sum
// all-reduce go straight onto RCCL (xGMI), stream-ordered on the library's launch stream, with no
// host round trip per message (the role of MPI_ISEND/IRECV/WAITALL in mpi/POP_HaloMod.F90:1865-1960
// and of MPI_ALLREDUCE in mpi/POP_ReductionsMod.F90:348-383).
//
// librccl is opened at run time (dlopen), so libpop_amd.so has no link-time dependency on it and a
// process that already holds an RCCL (e.g. the one bundled with a host framework) shares that
// instance.  The host side only has to broadcast the 128-byte unique id from rank 0
// (pop_rccl_unique_id -> pop_comm_init_rccl on every rank); INTEGRATION.md shows the Fortran call.
#pragma once
#include <dlfcn.h>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <string>

namespace pop {

struct RcclApi {
  // minimal restatement of the public RCCL/NCCL C API used here (rccl.h): opaque communicator,
  // 128-byte unique id, ncclDouble = 8, ncclSum = 0, result 0 = success
  struct UniqueId { char internal[128]; };
  typedef void *Comm;
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*CommCount)(const Comm, int *) = nullptr;       // optional (evidence for the bench line: how many ranks the communicator spans)
  int (*CommUserRank)(const Comm, int *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  void *handle = nullptr;
  std::string path;                                      // what dlopen bound
  static constexpr int kDouble = 8, kSum = 0;

  int load(std::string &err) {
    if (handle) return 0;
    // POP_RCCL_LIB names the library to bind instead of the system librccl (tests: tests/rccl_stub, which lets
    // several ranks share one GPU; a site build of RCCL).  An override that cannot be opened is an error.
    const char *over = getenv("POP_RCCL_LIB");
    if (over && *over) {
      handle = dlopen(over, RTLD_NOW | RTLD_LOCAL);
      if (!handle) { err = std::string("rccl transport: cannot open POP_RCCL_LIB=") + over + ": " + dlerror(); return 1; }
      path = over;
    } else {
      const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (const char *n : names) { handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (handle) { path = n; break; } }
      if (!handle) { err = std::string("rccl transport: cannot open librccl: ") + dlerror(); return 1; }
    }
    auto sym = [&](const char *n) { void *p = dlsym(handle, n); if (!p) err = std::string("rccl transport: missing symbol ") + n; return p; };
    GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
    Send = (decltype(Send))sym("ncclSend");
    Recv = (decltype(Recv))sym("ncclRecv");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    CommCount = (decltype(CommCount))dlsym(handle, "ncclCommCount");
    CommUserRank = (decltype(CommUserRank))dlsym(handle, "ncclCommUserRank");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !Send || !Recv || !GroupStart || !GroupEnd) return 1;
    return 0;
  }
  std::string what(int rc) const { return GetErrorString ? GetErrorString(rc) : "rccl error " + std::to_string(rc); }
};

inline RcclApi &rccl() { static RcclApi a; return a; }

// state of one context's native transport
struct RcclTransport {
  RcclApi::Comm comm = nullptr;
  RcclApi::Comm comm2 = nullptr;      // second communicator: exchanges on the side stream beside an all-reduce on `comm`
  hipStream_t *stream = nullptr;      // the context's launch stream (pointer: follows pop_set_stream)
  hipStream_t *side = nullptr;        // the context's side stream
  double *send = nullptr, *recv = nullptr, *red = nullptr;
  std::string err;
};

// pop_exchange_fn: offsets/counts are in doubles within the context's send / receive buffers
inline int rccl_exchange_on(RcclTransport *t, RcclApi::Comm comm, hipStream_t st, int nmsg, const int *peer, const long long *soff,
                            const long long *scnt, const long long *roff, const long long *rcnt) {
  RcclApi &a = rccl();
  int rc = a.GroupStart();
  for (int i = 0; i < nmsg && rc == 0; ++i) {
    if (rcnt[i]) rc = a.Recv(t->recv + roff[i], (size_t)rcnt[i], RcclApi::kDouble, peer[i], comm, st);
    if (rc == 0 && scnt[i]) rc = a.Send(t->send + soff[i], (size_t)scnt[i], RcclApi::kDouble, peer[i], comm, st);
  }
  const int rc2 = a.GroupEnd();
  if (rc == 0) rc = rc2;
  if (rc) { t->err = a.what(rc); return 1; }
  return 0;
}
inline int rccl_exchange(void *user, int nmsg, const int *peer, const long long *soff, const long long *scnt,
                         const long long *roff, const long long *rcnt) {
  RcclTransport *t = (RcclTransport *)user;
  return rccl_exchange_on(t, t->comm, *t->stream, nmsg, peer, soff, scnt, roff, rcnt);
}
// the same exchange on the side stream through the second communicator (operations on one communicator must not
// run concurrently; two communicators may)
inline int rccl_exchange_side(void *user, int nmsg, const int *peer, const long long *soff, const long long *scnt,
                              const long long *roff, const long long *rcnt) {
  RcclTransport *t = (RcclTransport *)user;
  return rccl_exchange_on(t, t->comm2, *t->side, nmsg, peer, soff, scnt, roff, rcnt);
}
// pop_allreduce_fn: in-place sum over ranks of red[off .. off+cnt)
inline int rccl_allreduce(void *user, long long off, long long cnt) {
  RcclTransport *t = (RcclTransport *)user;
  RcclApi &a = rccl();
  const int rc = a.AllReduce(t->red + off, t->red + off, (size_t)cnt, RcclApi::kDouble, RcclApi::kSum, t->comm, *t->stream);
  if (rc) { t->err = a.what(rc); return 1; }
  return 0;
}

}  // namespace pop
