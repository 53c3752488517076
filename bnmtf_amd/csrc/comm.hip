#include "comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

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
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;     // optional
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
  (void)sym(g_api.CommCount, "ncclCommCount");
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

// In-process transport (test hook): a communicator id that starts with "BNMTFLOC" joins the ranks of ONE process (one
// host thread per rank, any devices) through a rendezvous in host memory and device-to-device copies.  It lets a single
// GPU exercise everything of the sharded path except RCCL itself: shard ranges, the kernels' row offsets, the placement
// of the gathered blocks, the order of the reduced sums.
struct LocalGroup {
  std::mutex m;
  std::condition_variable cv;
  int world = 0, arrived = 0;
  uint64_t generation = 0;
  std::vector<void*> ptrs;
  std::vector<double> vals;
  bool barrier() {                               // false on timeout (a peer died): the caller reports an error instead of hanging
    std::unique_lock<std::mutex> lk(m);
    const uint64_t gen = generation;
    if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return true; }
    return cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != gen; });
  }
};
namespace {
std::mutex g_groups_m;
std::map<std::string, std::shared_ptr<LocalGroup>> g_groups;
}

struct Comm {
  ncclComm_t comm = nullptr;
  std::shared_ptr<LocalGroup> local;
  std::string local_key;
  int rank = 0, world = 1;
};

// what the communicator itself says: 0 none, 1 RCCL, 2 the in-process transport; its rank count as RCCL (ncclCommCount) reports it
int comm_info(const Comm* c, int* kind, int* ranks) {
  *kind = 0; *ranks = 1;
  if (!c) return BNMTF_OK;
  if (c->local) { *kind = 2; *ranks = c->world; return BNMTF_OK; }
  *kind = 1; *ranks = c->world;
  if (c->comm && g_api.CommCount) { int n = 0; if (g_api.CommCount(c->comm, &n) == ncclSuccess) *ranks = n; }
  return BNMTF_OK;
}

int comm_unique_id(uint8_t out[128]) {
  CHK(load_api());
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  NCHK(g_api.GetUniqueId(&id));
  memcpy(out, &id, 128);
  return BNMTF_OK;
}

int comm_create(Comm** out, const uint8_t idb[128], int rank, int world, hipStream_t) {
  if (!memcmp(idb, "BNMTFLOC", 8)) {
    Comm* c = new Comm();
    c->rank = rank; c->world = world;
    c->local_key.assign(reinterpret_cast<const char*>(idb), 128);
    std::lock_guard<std::mutex> lk(g_groups_m);
    auto& g = g_groups[c->local_key];
    if (!g) { g = std::make_shared<LocalGroup>(); g->world = world; g->ptrs.assign(world, nullptr); }
    if (g->world != world) { set_error("local communicator: world size mismatch"); delete c; return BNMTF_ECOMM; }
    c->local = g;
    *out = c;
    return BNMTF_OK;
  }
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
  if (c->local) {
    std::lock_guard<std::mutex> lk(g_groups_m);
    c->local.reset();
    auto it = g_groups.find(c->local_key);
    if (it != g_groups.end() && it->second.use_count() == 1) g_groups.erase(it);
    delete c;
    return;
  }
  if (c->comm) g_api.CommDestroy(c->comm);
  delete c;
}

int comm_allgather_factor(Comm* c, float* X, int KP, int nglob, int world, hipStream_t st) {
  auto first = [&](int r) { return (int)(((int64_t)nglob * r) / world); };
  if (c->local) {
    LocalGroup& g = *c->local;
    HIPCHK(hipStreamSynchronize(st));                     // own block written
    g.ptrs[c->rank] = X;
    if (!g.barrier()) { set_error("local communicator: peer did not arrive"); return BNMTF_ECOMM; }
    for (int r = 0; r < world; ++r) {
      if (r == c->rank) continue;
      const size_t off = (size_t)first(r) * KP, cnt = (size_t)(first(r + 1) - first(r)) * KP;
      HIPCHK(hipMemcpyAsync(X + off, static_cast<const float*>(g.ptrs[r]) + off, cnt * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    if (!g.barrier()) { set_error("local communicator: peer did not arrive"); return BNMTF_ECOMM; }   // peers have copied my block
    return BNMTF_OK;
  }
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

// op: 0 sum, 1 max.  In-process transport: every rank forms the result in rank order, so all ranks hold the same bits.
template <typename T>
static int allreduce_impl(Comm* c, T* buf, int count, int op, ncclDataType_t dt, hipStream_t st) {
  if (c->local) {
    LocalGroup& g = *c->local;
    const size_t words = ((size_t)c->world * count * sizeof(T) + sizeof(double) - 1) / sizeof(double);
    {
      std::lock_guard<std::mutex> lk(g.m);
      if (g.vals.size() < words) g.vals.assign(words, 0.0);
    }
    if (!g.barrier()) { set_error("local communicator: peer did not arrive"); return BNMTF_ECOMM; }
    T* all = reinterpret_cast<T*>(g.vals.data());
    HIPCHK(hipMemcpyAsync(all + (size_t)c->rank * count, buf, count * sizeof(T), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (!g.barrier()) { set_error("local communicator: peer did not arrive"); return BNMTF_ECOMM; }
    std::vector<T> res(count);
    for (int t = 0; t < count; ++t) {
      T v = all[t];
      for (int r = 1; r < c->world; ++r) { const T x = all[(size_t)r * count + t]; v = op == 1 ? std::max(v, x) : v + x; }
      res[t] = v;
    }
    HIPCHK(hipMemcpyAsync(buf, res.data(), count * sizeof(T), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    if (!g.barrier()) { set_error("local communicator: peer did not arrive"); return BNMTF_ECOMM; }
    return BNMTF_OK;
  }
  NCHK(g_api.AllReduce(buf, buf, (size_t)count, dt, op == 1 ? ncclMax : ncclSum, c->comm, st));
  return BNMTF_OK;
}
int comm_allreduce_sum(Comm* c, double* buf, int count, hipStream_t st) { return allreduce_impl(c, buf, count, 0, ncclDouble, st); }
int comm_allreduce_max(Comm* c, double* buf, int count, hipStream_t st) { return allreduce_impl(c, buf, count, 1, ncclDouble, st); }
int comm_allreduce_sum_f32(Comm* c, float* buf, int count, hipStream_t st) { return allreduce_impl(c, buf, count, 0, ncclFloat, st); }

}  // namespace bnmtf
