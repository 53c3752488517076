#include "comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "model.h"

namespace bnmtf {

namespace {
struct Api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
Api g_api;

template <typename F>
bool sym(F& f, const char* name) {
  void* p = dlsym(RTLD_DEFAULT, name);            // a copy already in the process (torch's) wins
  if (!p && g_api.lib) p = dlsym(g_api.lib, name);
  f = reinterpret_cast<F>(p);
  return p != nullptr;
}

int load_api() {
  if (g_api.ok) return BNMTF_OK;
  if (!dlsym(RTLD_DEFAULT, "ncclCommInitRank")) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      g_api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (g_api.lib) break;
    }
    if (!g_api.lib) { set_error("cannot load librccl: %s", dlerror()); return BNMTF_ECOMM; }
  }
  bool ok = sym(g_api.GetUniqueId, "ncclGetUniqueId") && sym(g_api.CommInitRank, "ncclCommInitRank") &&
            sym(g_api.CommDestroy, "ncclCommDestroy") && sym(g_api.AllGather, "ncclAllGather") &&
            sym(g_api.AllReduce, "ncclAllReduce") && sym(g_api.Broadcast, "ncclBroadcast") &&
            sym(g_api.GroupStart, "ncclGroupStart") && sym(g_api.GroupEnd, "ncclGroupEnd") &&
            sym(g_api.GetErrorString, "ncclGetErrorString");
  if (!ok) { set_error("librccl is missing a required symbol"); return BNMTF_ECOMM; }
  g_api.ok = true;
  return BNMTF_OK;
}
}  // namespace

#define NCHK(expr)                                                                 \
  do {                                                                             \
    ncclResult_t r_ = (expr);                                                      \
    if (r_ != ncclSuccess) {                                                       \
      set_error("%s failed: %s", #expr, g_api.GetErrorString(r_));                 \
      return BNMTF_ECOMM;                                                          \
    }                                                                              \
  } while (0)

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

int comm_unique_id(uint8_t out[128]) {
  CHK(load_api());
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  NCHK(g_api.GetUniqueId(&id));
  memcpy(out, &id, 128);
  return BNMTF_OK;
}

int comm_create(Comm** out, const uint8_t idb[128], int rank, int world, hipStream_t) {
  CHK(load_api());
  ncclUniqueId id;
  memcpy(&id, idb, 128);
  Comm* c = new Comm();
  c->rank = rank; c->world = world;
  ncclResult_t r = g_api.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank failed: %s", g_api.GetErrorString(r));
    delete c;
    return BNMTF_ECOMM;
  }
  *out = c;
  return BNMTF_OK;
}

void comm_destroy(Comm* c) {
  if (!c) return;
  if (c->comm) g_api.CommDestroy(c->comm);
  delete c;
}

int comm_allgather_factor(Comm* c, float* X, int KP, int nglob, int world, hipStream_t st) {
  auto first = [&](int r) { return (int)(((int64_t)nglob * r) / world); };
  if (nglob % world == 0) {
    const size_t cnt = (size_t)(nglob / world) * KP;
    NCHK(g_api.AllGather(X + (size_t)first(c->rank) * KP, X, cnt, ncclFloat, c->comm, st));
  } else {                                           // ragged split: one grouped broadcast per owner
    NCHK(g_api.GroupStart());
    for (int r = 0; r < world; ++r) {
      float* blk = X + (size_t)first(r) * KP;
      const size_t cnt = (size_t)(first(r + 1) - first(r)) * KP;
      NCHK(g_api.Broadcast(blk, blk, cnt, ncclFloat, r, c->comm, st));
    }
    NCHK(g_api.GroupEnd());
  }
  return BNMTF_OK;
}

int comm_allreduce_sum(Comm* c, double* buf, int count, hipStream_t st) {
  NCHK(g_api.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, c->comm, st));
  return BNMTF_OK;
}

}  // namespace bnmtf
