// Many models in one launch (round 6): the model-selection jobs of the reference (a cross-validation's folds x ranks: dozens of
// models of GDSC's size, 622 x 138) run a model per process slot, and a kernel of ONE such model occupies 18-78 blocks of a
// 256-CU chip.  Every kernel of the multi-launch path that a variational iteration uses has a LIST form: blockIdx.z = model, the
// arguments of model z read from a device array through the scalar cache (constant address space: as uniform as kernel
// arguments, no vector registers).  Host side: the launchers, called while a Recorder is installed (thread-local), append
// (list-form kernel, grid, block, LDS, argument bytes) instead of launching; api_many.inc runs the models' host code in lock-step
// and turns the records of a launch site into one launch.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

namespace bnmtf {

template <class P>
__device__ __forceinline__ P load_pack(const P* list, int idx) {
  static_assert(sizeof(P) % 4 == 0, "argument packs are whole dwords");
  typedef __attribute__((address_space(4))) const uint32_t cu32;
  cu32* src = (cu32*)(uintptr_t)(list + idx);
  P out;
  uint32_t* dst = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(P) / 4); ++i) dst[i] = src[i];
  return out;
}

struct LaunchRec {
  const void* fn = nullptr;            // the list form: (const Pack* list, int it)
  dim3 grid, block;
  size_t lds = 0;
  bool flex = false;                   // grid.x may differ between the models of one launch (the list form leaves with blockIdx.x >= its own count)
  uint32_t off = 0, size = 0;          // the argument pack's bytes in the recorder's arena
};
struct Recorder {                      // (cleared per iteration: the vectors keep their capacity, a record costs a memcpy)
  std::vector<LaunchRec> recs;
  std::vector<unsigned char> arena;
  void clear() { recs.clear(); arena.clear(); }
  const unsigned char* args(const LaunchRec& r) const { return arena.data() + r.off; }
};
extern thread_local Recorder* g_recorder;

template <class P>
inline bool record_launch(const void* many_fn, dim3 grid, dim3 block, size_t lds, const P& p, bool flex = false) {
  Recorder* rc = g_recorder;
  if (!rc) return false;
  LaunchRec r;
  r.fn = many_fn; r.grid = grid; r.block = block; r.lds = lds; r.flex = flex;
  r.off = (uint32_t)rc->arena.size(); r.size = (uint32_t)sizeof(P);
  rc->arena.resize(rc->arena.size() + sizeof(P));
  memcpy(rc->arena.data() + r.off, &p, sizeof(P));
  rc->recs.push_back(r);
  return true;
}

}  // namespace bnmtf
