// comm.hip -- RCCL panel exchange behind include/dbcsr_amd_comm.h (grouped ncclSend / ncclRecv over xGMI).
// librccl is opened lazily so that single-GPU use of the library has no dependency on it.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "../../include/dbcsr_amd_comm.h"
#include "common.h"

namespace {

// the part of rccl.h this file needs (C ABI of RCCL 2.x / ROCm 7: /opt/rocm/include/rccl/rccl.h)
typedef struct { char internal[128]; } nccl_unique_id;
typedef void* nccl_comm_t;
typedef int nccl_result_t;
constexpr int kNcclInt8 = 0;

struct Rccl {
  void* lib = nullptr;
  nccl_result_t (*GetUniqueId)(nccl_unique_id*) = nullptr;
  nccl_result_t (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
  nccl_result_t (*CommDestroy)(nccl_comm_t) = nullptr;
  nccl_result_t (*GroupStart)() = nullptr;
  nccl_result_t (*GroupEnd)() = nullptr;
  nccl_result_t (*Send)(const void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  nccl_result_t (*Recv)(void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  nccl_result_t (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(nccl_result_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      fprintf(stderr, "dbcsr_amd_comm: cannot load librccl (%s)\n", dlerror());
      return;
    }
#define DBCSR_SYM(field, sym)                                    \
  *reinterpret_cast<void**>(&r.field) = dlsym(r.lib, sym);       \
  if (!r.field) {                                                \
    fprintf(stderr, "dbcsr_amd_comm: librccl lacks %s\n", sym); \
    return;                                                      \
  }
    DBCSR_SYM(GetUniqueId, "ncclGetUniqueId")
    DBCSR_SYM(CommInitRank, "ncclCommInitRank")
    DBCSR_SYM(CommDestroy, "ncclCommDestroy")
    DBCSR_SYM(GroupStart, "ncclGroupStart")
    DBCSR_SYM(GroupEnd, "ncclGroupEnd")
    DBCSR_SYM(Send, "ncclSend")
    DBCSR_SYM(Recv, "ncclRecv")
    DBCSR_SYM(AllGather, "ncclAllGather")
    DBCSR_SYM(GetErrorString, "ncclGetErrorString")
#undef DBCSR_SYM
    r.ok = true;
  });
  return r;
}

struct Comm {
  nccl_comm_t c = nullptr;
  int rank = 0, nranks = 1;
};

int nccl_check(nccl_result_t e, const char* what) {
  if (e == 0) return 0;
  fprintf(stderr, "dbcsr_amd_comm: %s failed: %s\n", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "?");
  return -1;
}

}  // namespace

using dbcsr_amd::stream_of;

extern "C" {

int dbcsr_amd_comm_available(void) { return rccl().ok ? 1 : 0; }

int dbcsr_amd_comm_unique_id(char id[DBCSR_AMD_COMM_ID_BYTES]) {
  Rccl& r = rccl();
  if (!r.ok || !id) return -1;
  nccl_unique_id u;
  if (nccl_check(r.GetUniqueId(&u), "ncclGetUniqueId")) return -1;
  memcpy(id, u.internal, DBCSR_AMD_COMM_ID_BYTES);
  return 0;
}

int dbcsr_amd_comm_create(void** comm, const char id[DBCSR_AMD_COMM_ID_BYTES], int nranks, int rank) {
  Rccl& r = rccl();
  if (!comm) return -1;
  *comm = nullptr;
  if (!r.ok || !id || nranks < 1 || rank < 0 || rank >= nranks) return -1;
  Comm* c = new (std::nothrow) Comm();
  if (!c) return -1;
  nccl_unique_id u;
  memcpy(u.internal, id, DBCSR_AMD_COMM_ID_BYTES);
  if (nccl_check(r.CommInitRank(&c->c, nranks, u, rank), "ncclCommInitRank")) {
    delete c;
    return -1;
  }
  c->rank = rank;
  c->nranks = nranks;
  *comm = c;
  return 0;
}

int dbcsr_amd_comm_destroy(void* comm) {
  if (!comm) return 0;
  Comm* c = static_cast<Comm*>(comm);
  int rc = c->c ? nccl_check(rccl().CommDestroy(c->c), "ncclCommDestroy") : 0;
  delete c;
  return rc;
}

int dbcsr_amd_comm_rank(void* comm, int* rank, int* nranks) {
  if (!comm) return -1;
  Comm* c = static_cast<Comm*>(comm);
  if (rank) *rank = c->rank;
  if (nranks) *nranks = c->nranks;
  return 0;
}

int dbcsr_amd_comm_exchange(void* comm, const dbcsr_amd_comm_op* sends, int nsend, const dbcsr_amd_comm_op* recvs, int nrecv, void* stream) {
  if (!comm || nsend < 0 || nrecv < 0 || (nsend && !sends) || (nrecv && !recvs)) return -1;
  Comm* c = static_cast<Comm*>(comm);
  Rccl& r = rccl();
  for (int i = 0; i < nsend; ++i)
    if (sends[i].peer < 0 || sends[i].peer >= c->nranks || sends[i].bytes < 0 || (sends[i].bytes && !sends[i].buf)) return -1;
  for (int i = 0; i < nrecv; ++i)
    if (recvs[i].peer < 0 || recvs[i].peer >= c->nranks || recvs[i].bytes < 0 || (recvs[i].bytes && !recvs[i].buf)) return -1;
  hipStream_t st = stream_of(stream);
  int rc = nccl_check(r.GroupStart(), "ncclGroupStart");
  for (int i = 0; i < nsend && rc == 0; ++i)
    if (sends[i].bytes) rc = nccl_check(r.Send(sends[i].buf, (size_t)sends[i].bytes, kNcclInt8, sends[i].peer, c->c, st), "ncclSend");
  for (int i = 0; i < nrecv && rc == 0; ++i)
    if (recvs[i].bytes) rc = nccl_check(r.Recv(recvs[i].buf, (size_t)recvs[i].bytes, kNcclInt8, recvs[i].peer, c->c, st), "ncclRecv");
  const int rc_end = nccl_check(r.GroupEnd(), "ncclGroupEnd");  // always close the group
  return rc ? rc : rc_end;
}

int dbcsr_amd_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
  if (!comm || bytes_per_rank < 0 || (bytes_per_rank && (!send || !recv))) return -1;
  if (bytes_per_rank == 0) return 0;
  Comm* c = static_cast<Comm*>(comm);
  return nccl_check(rccl().AllGather(send, recv, (size_t)bytes_per_rank, kNcclInt8, c->c, stream_of(stream)), "ncclAllGather");
}

}  // extern "C"
