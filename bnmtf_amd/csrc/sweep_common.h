// Shared device helpers of the register-resident sweep kernels (sweep_chip.inc, kernel_sweep_vb.hip, kernel_bnmtf.hip):
// half-wave reductions and broadcasts, the one-instruction TN candidate arithmetic, LDS-DMA panel staging.
#pragma once
#include <atomic>

#include "kernels.h"
#include "device_rng.h"

namespace bnmtf {

__device__ __forceinline__ float dpp_xor_row_sum(float v) {   // all-reduce inside each 16-lane row
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}
__device__ __forceinline__ float half_sum(float v) {          // all-reduce inside each 32-lane half
  v = dpp_xor_row_sum(v);
  return v + __shfl_xor(v, 16, 64);
}
// Sum over each 32-lane half, valid in the UPPER 16 lanes of the half only (lanes 16-31 and 48-63): four DPP steps
// inside the rows, then row_bcast:15 adds row 0's total into row 1 (and row 2's into row 3).  No LDS round trip.
__device__ __forceinline__ float half_sum_upper(float v) {
  v = dpp_xor_row_sum(v);
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));   // row_bcast:15, rows 1 and 3
}
// The same with the last step as ONE instruction: rows 1 and 3 add lane 15 of the row before them, rows 0 and 2 keep their value
// (through the builtin the compiler makes a zero, a DPP move and an add of it).  s_nop 1: the two wait states a DPP read needs
// behind the vector instruction that wrote its source.  For the Gibbs sweep's column loop (sweep_chip.inc), whose phases are
// fenced anyway: in the VB sweep's loop the volatile statement upset the register allocation (750 spilled registers, 4 x slower).
__device__ __forceinline__ float half_sum_upper_fused(float v) {
  v = dpp_xor_row_sum(v);
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
  return v;
}
// v[lane & 31] + v[(lane & 31) + 32] in every lane, the lower half's term first in both halves (gfx950's v_permlane32_swap:
// the upper half of the first operand and the lower half of the second change places; written as asm because the builtin of
// this compiler folds the two results of a swap of a register with its own copy into one).  s_nop 1: the wait states
// behind the vector instruction that wrote the operands.
__device__ __forceinline__ float half_swap_sum(float v) {
  float lo = v, hi = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
  return lo + hi;
}
__device__ __forceinline__ double half_sum_d(double v) {
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
// value held by lane (half*32 + src) for every lane of that half
__device__ __forceinline__ float half_bcast(float v, int src, int half) {
  const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
  const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src + 32));
  return half ? a1 : a0;
}
__device__ __forceinline__ uint32_t half_bcast_u(uint32_t v, int src, int half) {
  const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)v, src);
  const uint32_t a1 = (uint32_t)__builtin_amdgcn_readlane((int)v, src + 32);
  return half ? a1 : a0;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// reciprocal-based parameters (v_rcp / v_rsq, ~1 ulp): the sampler needs no correctly rounded division
struct TnFast { float mu, irt, a, d, ilam; bool live, tail; };
// the part that needs tau_p only (a kernel that knows tau_p ahead of its sequential chain takes it off the chain) ...
struct TnPre { float irt, rcp, tpirt; bool live; };
__device__ __forceinline__ TnPre tn_fast_pre(float tau_p) {
#pragma clang fp contract(off)      // explicit fmaf only: every kernel that inlines this rounds identically (same chain on 1 and N GPUs)
  TnPre q;
  q.live = tau_p > 0.0f;
  const float tp = q.live ? tau_p : 1.0f;
  q.irt = __builtin_amdgcn_rsqf(tp);             // sigma   (v_rsq_f32 / v_rcp_f32 / v_sqrt_f32: ~1 ulp, one instruction each)
  q.rcp = __builtin_amdgcn_rcpf(tp);
  q.tpirt = tp * q.irt;                          // sqrt(tau)
  return q;
}
// ... and the part that needs the numerator
__device__ __forceinline__ TnFast tn_fast_post(const TnPre& q, float numer) {
#pragma clang fp contract(off)
  TnFast p;
  p.irt = q.irt;
  p.mu = numer * q.rcp;
  p.a = -p.mu * q.tpirt;                         // -mu * sqrt(tau)
  p.live = q.live && isfinite(p.a);
  p.d = 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(p.a, p.a, 4.0f)) + p.a);
  p.ilam = p.d;                                  // Robert's rate lam = a + d satisfies lam d = 1: no second reciprocal on the samplers' chains (round 5)
  p.tail = p.a >= kTnA0;
  return p;
}
__device__ __forceinline__ TnFast tn_fast_params(float numer, float tau_p) { return tn_fast_post(tn_fast_pre(tau_p), numer); }
// One candidate, in two parts: what depends on the random words only (can be issued ahead of the sequential chain) ...
// (sw = sqrt(-2 ln u2): the translated-exponential candidate e = nl / lam is accepted iff u2 <= exp(-(e - d)^2 / 2), i.e. iff
// |e - d| <= sw -- the same event stated without an exponential on the chain that waits for it; the logarithm and the root depend on
// the random word only and are made with the candidate's other word-only parts, ahead of the chain: round 5)
struct TnCand { float nl, z, sw; };
__device__ __forceinline__ TnCand tn_cand_pre(uint32_t r0, uint32_t r1) {
#pragma clang fp contract(off)
  TnCand c;
  const float u1 = u23(r0), u2 = u23(r1);
  c.nl = -0.69314718f * __builtin_amdgcn_logf(u1);                       // v_log_f32 is log2
  c.z = __builtin_amdgcn_sqrtf(2.0f * c.nl) * __builtin_amdgcn_cosf(u2);     // v_cos_f32 takes revolutions
  c.sw = __builtin_amdgcn_sqrtf(-1.38629436f * __builtin_amdgcn_logf(u2));
  return c;
}
// ... and the acceptance test + value given the conditional's parameters
__device__ __forceinline__ bool tn_cand_post(const TnFast& p, const TnCand& c, float* x) {
#pragma clang fp contract(off)
  const float e = c.nl * p.ilam;
  const float t = e - p.d;
  const bool acc_t = fabsf(t) <= c.sw;           // u2 <= exp(-t^2 / 2)
  const bool acc_n = c.z >= p.a;
  *x = p.tail ? e * p.irt : fmaf(c.z, p.irt, p.mu);
  return p.tail ? acc_t : acc_n;
}
__device__ __forceinline__ bool tn_eval_fast(const TnFast& p, uint32_t r0, uint32_t r1, float* x) { return tn_cand_post(p, tn_cand_pre(r0, r1), x); }

// LDS-direct staging of one panel: `chunks` pieces of 1 KiB (64 lanes x 16 B), wave w takes
// chunks w, w+8, ...  No VGPRs, no ds_write; completion: the issuing wave's vmcnt(0), which
// sync_with_dma() (below) waits for ahead of the barrier -- __syncthreads() alone does not.
template <int NW>
__device__ __forceinline__ void stage_panel(const float* src, float* dst, int chunks, int wave, int lane) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  for (int c = wave; c < chunks; c += NW) {
    __builtin_amdgcn_global_load_lds(src + (size_t)c * 256 + lane * 4, (lds_ptr)(dst + (size_t)c * 256), 16, 0, 0);
  }
}

// The same staging through a buffer descriptor: panel base and chunk offset travel in SGPRs (descriptor + soffset), the
// only VGPR is the lane's byte offset, so one piece costs s_mov m0 + buffer_load ... lds instead of 64-bit VGPR address
// arithmetic per piece.  `lane16` = 16 * lane.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t panel_rsrc(const float* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
template <int NW>
__device__ __forceinline__ void stage_panel_buf(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, float* dst, int chunks, int wave, int lane16) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  for (int c = wave; c < chunks; c += NW)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(dst + (size_t)c * 256), 16, lane16, (int)(byte_off + (uint32_t)c * 1024u), 0, 0);
}

// Workgroup barrier for kernels that stage through LDS-DMA.  __syncthreads() is a workgroup-scope fence + s_barrier; on
// gfx950 that fence waits for the LDS counter only, and the compiler adds a vector-memory wait just where ITS alias
// analysis sees one of this wave's LDS reads meet an LDS-DMA still in flight -- which says nothing about the OTHER
// waves that gather from the piece this wave staged.  (Seen as wrong rows from some column on when three ranks shared one
// GPU and the DMA took longer than a column: tools/stress_sharded.py.)  Every barrier that is meant to publish staged
// pieces waits for vmcnt(0) explicitly.
__device__ __forceinline__ void sync_with_dma() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- q hand-over between the half sweeps (FastArgs::ho_*, model.h Dir::ho_*, kernel_handover.hip): the pieces the Gibbs
// kernel (sweep_chip.inc) and the VB kernel (kernel_sweep_vb.hip) share
// the block's region -- q of its missing entries as the other direction's sweep left them -- on its way into LDS at `dst`
template <int NW>
__device__ __forceinline__ void ho_issue_region(const FastArgs& f, float* dst, int wave, int lane) {
  const uint32_t r0 = f.ho_region_ofs[blockIdx.x], r1 = f.ho_region_ofs[blockIdx.x + 1];       // entries, multiples of 256
  stage_panel_buf<NW>(panel_rsrc(f.ho_region + r0, (size_t)(r1 - r0) * 4), 0u, dst, (int)((r1 - r0) / 256), wave, lane * 16);
}
// the staged runs go out, one contiguous packet per destination block.  A packet per half wave and pass (32 lanes x 4 entries
// cover most runs at once); the descriptors of 16 packets -- (start in the staging area, entries, start in the other
// direction's regions), multiples of 4 -- are fetched by one load per wave.
template <int NW>
__device__ __forceinline__ void ho_send_packets(const FastArgs& f, const float* stage, int wave, int lane) {
  const int half = lane >> 5, l5 = lane & 31;
  const uint32_t* pk = f.ho_pk + (size_t)blockIdx.x * f.ho_nb_other * 3;
  for (int p0 = 16 * wave; p0 < f.ho_nb_other; p0 += 16 * NW) {
    const int np = min(16, f.ho_nb_other - p0);
    uint32_t d0 = 0;
    if (lane < 3 * np) d0 = pk[3 * p0 + lane];
    auto word = [&](int wd) { return (uint32_t)__shfl((int)d0, wd & 63); };
    for (int i = 0; i < np; i += 4) {
      // half 0 takes packets i and i + 2, half 1 packets i + 1 and i + 3 (words 3 p .. 3 p + 2 of the descriptor list)
      // (every lane takes part in every shuffle: a lane switched off by a condition would hand its neighbours nothing)
      const int pa = i + half, pb = i + 2 + half;
      const uint32_t sa = word(3 * pa), ca_all = word(3 * pa + 1), ga = word(3 * pa + 2);
      const uint32_t sb = word(3 * pb), cb_all = word(3 * pb + 1), gb = word(3 * pb + 2);
      const uint32_t ca = pa < np ? ca_all : 0u, cb = pb < np ? cb_all : 0u;
      const uint32_t l0 = (uint32_t)l5 * 4u;
      float4 va = {}, va2 = {}, vb = {}, vb2 = {};
      if (l0 < ca) va = *reinterpret_cast<const float4*>(stage + sa + l0);
      if (l0 + 128u < ca) va2 = *reinterpret_cast<const float4*>(stage + sa + l0 + 128u);
      if (l0 < cb) vb = *reinterpret_cast<const float4*>(stage + sb + l0);
      if (l0 + 128u < cb) vb2 = *reinterpret_cast<const float4*>(stage + sb + l0 + 128u);
      if (l0 < ca) *reinterpret_cast<float4*>(f.ho_dst + ga + l0) = va;
      if (l0 + 128u < ca) *reinterpret_cast<float4*>(f.ho_dst + ga + l0 + 128u) = va2;
      if (l0 < cb) *reinterpret_cast<float4*>(f.ho_dst + gb + l0) = vb;
      if (l0 + 128u < cb) *reinterpret_cast<float4*>(f.ho_dst + gb + l0 + 128u) = vb2;
      for (uint32_t l = l0 + 256u; l < ca; l += 128u) *reinterpret_cast<float4*>(f.ho_dst + ga + l) = *reinterpret_cast<const float4*>(stage + sa + l);   // (long runs: rare)
      for (uint32_t l = l0 + 256u; l < cb; l += 128u) *reinterpret_cast<float4*>(f.ho_dst + gb + l) = *reinterpret_cast<const float4*>(stage + sb + l);
    }
  }
}

#ifdef BNMTF_PHASE_TIMING
// debug build only (make timing): shader-clock stamps at the phase boundaries; a few blocks print their sums
__device__ __forceinline__ unsigned long long tick(float dep) {
  unsigned long long t;
  asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
  return t;
}
#define TICK(i, dep) do { const unsigned long long t_ = tick(dep); ph[i] += t_ - tlast; tlast = t_; } while (0)
#else
// Phase boundary: nothing is scheduled across it.  Left to itself the compiler interleaves the latency-bound reduction
// and sampler chain with the slot work of the same wave; with several waves per SIMD that costs ~25 % (measured),
// because a wave then holds issue slots in its slot phase that the other waves' chains could have used.
#define TICK(i, dep) __builtin_amdgcn_sched_barrier(0)
#endif

// Sum of the contraction's partial slabs for one (unit, column): P = sum_s slabs[s][u][kk], added in slab order (so every
// kernel rounds alike), with the loads of eight slabs issued together -- a shard of a multi-GPU run has 32 or more slabs
// and a load-wait-add loop serialises their L2 latency (measured: 19 us of a 100 us sweep).
__device__ __forceinline__ float slab_sum_ordered(const float* slabs, int split, size_t slab_stride, size_t elem) {
  float p = 0.f;
  const float* q = slabs + elem;
  for (int s0 = 0; s0 < split; s0 += 8) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = s0 + j < split ? q[(size_t)(s0 + j) * slab_stride] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (s0 + j < split) p += t[j];
  }
  return p;
}

// Dynamic-LDS opt-in of a kernel (more than 64 KiB): a per-DEVICE attribute, so it is set once per (kernel, device) --
// a process may hold handles on several devices -- and its result is checked (a launch that asks for more LDS than the
// attribute allows fails late and anonymously otherwise).  `mask` is the calling instantiation's own flag word.
inline bool allow_full_lds(const void* kernel, std::atomic<uint64_t>& mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;     // devices >= 63 share a flag: set every time
  const uint64_t bit = 1ull << dev;
  if (dev != 63 && (mask.load(std::memory_order_acquire) & bit)) return true;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
  mask.fetch_or(bit, std::memory_order_release);
  return true;
}

typedef __attribute__((address_space(3))) const float lds_cf;
typedef __attribute__((address_space(3))) const f32x2 lds_cf2;
typedef __attribute__((address_space(3))) float* lds_fp;

}  // namespace bnmtf
