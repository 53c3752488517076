// K1/K2: the masked contractions  P = R~ . V  and  Pv = R~^T . U  ("the U^T.R step").
//
// Both are the same skinny product  out[n][KP] = sum_r big[r][n] * X[r][KP]  with n, r ~ 8192 and KP = 32/64: every
// element of `big` (the 256 MiB operand) is used by exactly one wave, so it is streamed HBM -> VGPR directly in MFMA
// B-fragment shape (no LDS round trip; see cdna_hip_programming.md "GEMV / operand streamed once"), while the small
// factor X is the A operand.  A wave owns 128 output columns x KP and a private slice of the inner dimension; the four
// waves of a block reduce through LDS and the block writes one partial slab, which the sweep kernel sums in its prologue.
//
// Two kernels:
//  * gemm_bf16x3_kernel (default, further down): fp32-exact products on the bf16 matrix cores from three-term operand
//    splits -- the contraction becomes a stream of R~ from HBM;
//  * gemm_kernel (BNMTF_GEMM=f32, kept for comparison): v_mfma_f32_32x32x2_f32, bound by the f32 matrix-core rate:
//      D[i][j] += sum_{k<2} A[i][k] B[k][j]
//      A: lane l holds A[i = l&31][k = l>>5]   -> X[r + (l>>5)][mt*32 + (l&31)]      (coalesced 128 B)
//      B: lane l holds B[k = l>>5][j = l&31]   -> big[r + (l>>5)][col0 + 4*(l&31) + t] (one dwordx4 = 4 tiles)
//      D: reg g, lane l -> i = (g&3) + 8*(g>>2) + 4*(l>>5), j = l&31
//    One dwordx4 per lane per r-pair gives the B fragments of four 32-column tiles whose columns interleave with
//    stride 4, so each wave-instruction reads two fully used 512 B row segments.
#include <cstdlib>
#include <cstring>

#include "kernels.h"
#include "many.h"

namespace bnmtf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT, int U, int WPS>
__global__ __launch_bounds__(256, WPS) void gemm_kernel(GemmArgs a) {
  constexpr int KP = MT * 32;
  constexpr int NACC = MT * 4 * 16;            // accumulator floats per lane
  __shared__ float red[2][NACC * 64];          // 2 x 32 KiB (MT=2)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int col0 = blockIdx.x * 128;
  const int s = blockIdx.y + a.split0;
  const int ipw = a.inner_per_wave;
  const size_t r0 = (size_t)(s * 4 + wave) * ipw + h;

  const float* bp = a.big + r0 * (size_t)a.ld + col0 + 4 * c;
  const float* xp = a.X + r0 * KP + c;
  const size_t bstep = 2 * (size_t)a.ld;

  f32x16 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[mt][t][g] = 0.0f;

  // Software pipeline in registers: while the MFMAs of group g run, the loads of group g+1 are in flight
  // (U r-pairs = U KiB of `big` per wave; one wave per SIMD, so the wave has to cover its own HBM latency).
  f32x4 b0[U], b1[U];
  float a0[U][MT], a1[U][MT];
  auto load_group = [&](f32x4 (&b)[U], float (&av)[U][MT]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      b[u] = *reinterpret_cast<const f32x4*>(bp + u * bstep);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[u][mt] = xp[u * 2 * KP + mt * 32];
    }
    bp += U * bstep;
    xp += U * 2 * KP;
  };
  auto mfma_group = [&](const f32x4 (&b)[U], const float (&av)[U][MT]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][mt], b[u][t], acc[mt][t], 0, 0, 0);
  };
  const int ngroups = ipw / (2 * U);              // ipw is a multiple of 2*U (host pads)
  if (WPS == 1) {                                 // three register sets: two groups in flight behind the one being multiplied
    f32x4 b2[U]; float a2[U][MT];
    load_group(b0, a0);
    if (ngroups > 1) load_group(b1, a1);
    for (int g = 0; g < ngroups; g += 3) {
      if (g + 2 < ngroups) load_group(b2, a2);
      mfma_group(b0, a0);
      if (g + 1 < ngroups) {
        if (g + 3 < ngroups) load_group(b0, a0);
        mfma_group(b1, a1);
      }
      if (g + 2 < ngroups) {
        if (g + 4 < ngroups) load_group(b1, a1);
        mfma_group(b2, a2);
      }
    }
  } else {
    load_group(b0, a0);
    for (int g = 0; g < ngroups; g += 2) {
      if (g + 1 < ngroups) load_group(b1, a1);
      mfma_group(b0, a0);
      if (g + 1 < ngroups) {
        if (g + 2 < ngroups) load_group(b0, a0);
        mfma_group(b1, a1);
      }
    }
  }

  // cross-wave tree reduction through LDS: (2,3) -> (0,1), then 1 -> 0
  auto put = [&](float* dst) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) dst[((mt * 4 + t) * 16 + g) * 64 + lane] = acc[mt][t][g];
  };
  auto add = [&](const float* src) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[mt][t][g] += src[((mt * 4 + t) * 16 + g) * 64 + lane];
  };
  if (wave >= 2) put(red[wave - 2]);
  __syncthreads();
  if (wave < 2) add(red[wave]);
  __syncthreads();
  if (wave == 1) put(red[0]);
  __syncthreads();
  if (wave == 0) {
    add(red[0]);
    float* out = a.slabs + ((size_t)s * a.n_pad + col0) * KP;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 v = {acc[mt][t][4 * g4], acc[mt][t][4 * g4 + 1], acc[mt][t][4 * g4 + 2], acc[mt][t][4 * g4 + 3]};
          *reinterpret_cast<f32x4*>(out + (size_t)(4 * c + t) * KP + mt * 32 + 8 * g4 + 4 * h) = v;
        }
  }
}

// ---------------------------------------------------------------------------
// The same product on the bf16 matrix cores, exact to fp32: every fp32 operand is split into three bf16 terms
// x = hi + mid + lo (8 significant bits each, round-to-nearest residuals) and the product is taken as
//   hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid        (the dropped terms are below 2^-24 of the product)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs (32 cycles each) replace eight f32 MFMAs (64 cycles
// each) per 16 inner rows: the contraction stops being bound by the matrix cores (f32 MFMA peak = 157 TFLOP/s) and
// becomes a stream of R~ from HBM.
//   A (factor):  lane l holds A[i = l&31][k = 8*(l>>5) .. +7]  -> X[r0 + 8*(l>>5) + 0..7][mt*32 + (l&31)]
//   B (R~):      lane l holds B[k = 8*(l>>5) .. +7][j = l&31]  -> big[r0 + 8*(l>>5) + 0..7][col0 + 4*(l&31) + t]
//   D: as the f32 32x32 tile (reg g, lane l -> i = (g&3) + 8*(g>>2) + 4*(l>>5), j = l&31)
// so the loads (one dwordx4 per lane per row: four interleaved column tiles), the per-wave inner slices, the LDS tree
// reduction and the slab layout are those of gemm_kernel.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2g __attribute__((ext_vector_type(2)));

// 8 fp32 -> three packed bf16x8 planes (element j of a plane = bf16 term of v[j]).  Round-to-nearest at every level
// (v_cvt_pk_bf16_f32): the residuals are signed, so the three dropped cross terms have no systematic sign -- with
// truncation they are all positive for positive operands and bias the sums by ~2^-25, which the SSE identity sees.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2c __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_rne(float a0, float a1) {
  f32x2c v; v.x = a0; v.y = a1;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3(const float (&v)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a0 = v[2 * p], a1 = v[2 * p + 1];
    const uint32_t h = pack_rne(a0, a1);
    hi[p] = h;
    const float b0 = a0 - __builtin_bit_cast(float, h << 16);                 // exact
    const float b1 = a1 - __builtin_bit_cast(float, h & 0xffff0000u);
    const uint32_t m = pack_rne(b0, b1);
    mid[p] = m;
    const float c0 = b0 - __builtin_bit_cast(float, m << 16);                 // exact
    const float c1 = b1 - __builtin_bit_cast(float, m & 0xffff0000u);
    lo[p] = pack_rne(c0, c1);
  }
}

template <int MT, int NSET, int TW, int RB = 1>      // RB = 0 (BNMTF_GEMM_RING=old, A/B only): the ring with its loads behind conditions
__device__ __forceinline__ void gemm_bf16x3_body(const GemmArgs& a) {
  typedef float f32xT __attribute__((ext_vector_type(TW)));   // TW column tiles per wave: one TW-dword load per lane per row
  constexpr int KP = MT * 32;
  constexpr int NACC = MT * TW * 16;            // accumulator floats per lane
  __shared__ float red[2][NACC * 64];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int col0 = blockIdx.x * (32 * TW);
  const int s = blockIdx.y + a.split0;
  const int ipw = a.inner_per_wave;
  const size_t r0 = (size_t)(s * 4 + wave) * ipw + 8 * h;

  const float* bp = a.big + r0 * (size_t)a.ld + col0 + TW * c;
  const float* xp = a.X + r0 * KP + c;

  f32x16 acc[MT][TW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[mt][t][g] = 0.0f;

  f32xT braw[NSET][8];
  float araw[NSET][MT][8];
  auto load_step = [&](int gi, f32xT (&b)[8], float (&av)[MT][8]) {
    const float* bpg = bp + (size_t)gi * 16 * (size_t)a.ld;
    const float* xpg = xp + (size_t)gi * 16 * KP;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      b[rr] = __builtin_nontemporal_load(reinterpret_cast<const f32xT*>(bpg + (size_t)rr * a.ld));   // streamed once: keep it out of the way of X and the slabs in L2
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[mt][rr] = xpg[rr * KP + mt * 32];
    }
  };
  auto mul_step = [&](const f32xT (&b)[8], const float (&av)[MT][8]) {
    u32x4 ah[MT], am[MT], al[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) split3(av[mt], ah[mt], am[mt], al[mt]);
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      float bv[8];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) bv[rr] = b[rr][t];
      u32x4 bh, bm, bl;
      split3(bv, bh, bm, bl);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x16 d = acc[mt][t];
        // small terms first
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[mt]), __builtin_bit_cast(bf16x8, bh), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[mt]), __builtin_bit_cast(bf16x8, bl), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, am[mt]), __builtin_bit_cast(bf16x8, bm), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, am[mt]), __builtin_bit_cast(bf16x8, bh), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[mt]), __builtin_bit_cast(bf16x8, bm), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[mt]), __builtin_bit_cast(bf16x8, bh), d, 0, 0, 0);
        acc[mt][t] = d;
      }
#ifdef BNMTF_GEMM_SLEEP          // (tools/variant.sh experiment: a pause behind every tile's products -- does a cooler contraction buy the sweep a higher clock?)
      __builtin_amdgcn_s_sleep(BNMTF_GEMM_SLEEP);
#endif
    }
  };
#ifdef BNMTF_GEMM_STAGGER        // (tools/variant.sh experiment: the block's four waves -- one per SIMD -- a fraction of a step apart, so that their product bursts do not coincide)
  for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(BNMTF_GEMM_STAGGER);
#endif
  const int nsteps = ipw / 16;                    // ipw is a multiple of 32 (host pads)
  // ring of NSET raw-operand register sets: NSET-1 steps (8 KiB of R~ each) in flight behind the one being multiplied.
  // No branch in the steady state (round 4): with the loads behind `if (more steps)` the compiler's wait counts at the join
  // points came out as vmcnt(0) at the top of every NSET steps -- the ring was drained once per trip, one step in flight where
  // two were meant.  A load that would run past the slice re-reads the slice's last step instead; the scheduling fences keep a
  // later step's splits from being hoisted above the loads they would then wait for.  Same products in the same order.
  const int last = nsteps - 1;
  if constexpr (RB == 0) {
    int nl = 0;
#pragma unroll
    for (int j = 0; j < NSET - 1; ++j)
      if (j < nsteps) load_step(nl++, braw[j], araw[j]);
    for (int g = 0; g < nsteps; g += NSET) {
#pragma unroll
      for (int j = 0; j < NSET; ++j) {
        if (g + j < nsteps) {
          if (g + j + NSET - 1 < nsteps) load_step(nl++, braw[(j + NSET - 1) % NSET], araw[(j + NSET - 1) % NSET]);
          mul_step(braw[j], araw[j]);
        }
      }
    }
  } else {
#pragma unroll
  for (int j = 0; j < NSET - 1; ++j) load_step(j < last ? j : last, braw[j], araw[j]);
  int g = 0;
  for (; g + NSET <= nsteps; g += NSET) {
#pragma unroll
    for (int j = 0; j < NSET; ++j) {
      const int gl = g + j + NSET - 1;
      load_step(gl < last ? gl : last, braw[(j + NSET - 1) % NSET], araw[(j + NSET - 1) % NSET]);
      __builtin_amdgcn_sched_barrier(0);
      mul_step(braw[j], araw[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int j = 0; j < NSET - 1; ++j)
    if (g + j < nsteps) mul_step(braw[j], araw[j]);
  }

  // cross-wave tree reduction through LDS: (2,3) -> (0,1), then 1 -> 0
  auto put = [&](float* dst) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) dst[((mt * TW + t) * 16 + g) * 64 + lane] = acc[mt][t][g];
  };
  auto add = [&](const float* src) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[mt][t][g] += src[((mt * TW + t) * 16 + g) * 64 + lane];
  };
  if (wave >= 2) put(red[wave - 2]);
  __syncthreads();
  if (wave < 2) add(red[wave]);
  __syncthreads();
  if (wave == 1) put(red[0]);
  __syncthreads();
  if (wave == 0) {
    add(red[0]);
    float* out = a.slabs + ((size_t)s * a.n_pad + col0) * KP;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 v = {acc[mt][t][4 * g4], acc[mt][t][4 * g4 + 1], acc[mt][t][4 * g4 + 2], acc[mt][t][4 * g4 + 3]};
          *reinterpret_cast<f32x4*>(out + (size_t)(TW * c + t) * KP + mt * 32 + 8 * g4 + 4 * h) = v;
        }
  }
}

template <int MT, int NSET, int TW, int RB = 1>
__global__ __launch_bounds__(256, (TW == 4 ? 1 : 2)) void gemm_bf16x3_kernel(GemmArgs a) { gemm_bf16x3_body<MT, NSET, TW, RB>(a); }
// list form (many.h): blockIdx.z = model
template <int MT, int NSET, int TW>
__global__ __launch_bounds__(256, (TW == 4 ? 1 : 2)) void gemm_bf16x3_many(const GemmArgs* list, int) { gemm_bf16x3_body<MT, NSET, TW, 1>(load_pack(list, blockIdx.z)); }

#ifndef BNMTF_GEMM_NSET
#define BNMTF_GEMM_NSET 3        // raw-operand register sets of the ring (tools/variant.sh builds try 4)
#endif
void launch_gemm(const GemmArgs& a, int KP, hipStream_t st) {
  // (a launch covers the inner slices [split0, split0 + nsplit): all of them, or -- several GPUs -- first the ones over the rank's
  // own rows of the factor, the rest once the other ranks' blocks have arrived; every slice writes its own slab either way)
  const int ns = a.nsplit > 0 ? a.nsplit : a.split;
  if (ns <= 0) return;
  dim3 grid(a.n_pad / 128, ns), block(256);
#ifdef BNMTF_EXPERIMENTS
  const char* mode = getenv("BNMTF_GEMM");                        // "f32": the f32-MFMA kernel (the cross-check of the bf16x3 products: tests/test_contraction_gpu.py); read per launch
  const bool f32 = mode && !strcmp(mode, "f32");
#else
  constexpr bool f32 = false;
#endif
  if (!f32) {
    // three raw-operand register sets (two 8 KiB steps in flight per wave); deeper rings measured no faster
    // TW = 4 column tiles (128 columns) per wave, one wave per SIMD.  TW = 2 with two waves per SIMD (grid n_pad/64) was
    // measured slower: 68-70 us against 60-62 us (the factor operand is split twice as often per MFMA).
#ifdef BNMTF_EXPERIMENTS
    const char* ring = getenv("BNMTF_GEMM_RING");      // "old": the ring with its loads behind conditions (tools/ab_gemm_ring.sh; make EXPERIMENTS=1)
    if (ring && !strcmp(ring, "old")) {
      if (KP == 32) hipLaunchKernelGGL((gemm_bf16x3_kernel<1, 3, 4, 0>), grid, block, 0, st, a);
      else if (a.tw == 2) hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 3, 2, 0>), dim3(a.n_pad / 64, ns), block, 0, st, a);
      else          hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 3, 4, 0>), grid, block, 0, st, a);
      return;
    }
#endif
    if (g_recorder) {
      if (KP == 32) record_launch((const void*)gemm_bf16x3_many<1, BNMTF_GEMM_NSET, 4>, grid, block, 0, a);
      else if (a.tw == 2) record_launch((const void*)gemm_bf16x3_many<2, BNMTF_GEMM_NSET, 2>, dim3(a.n_pad / 64, ns), block, 0, a);
      else record_launch((const void*)gemm_bf16x3_many<2, BNMTF_GEMM_NSET, 4>, grid, block, 0, a);
      return;
    }
    if (KP == 32) hipLaunchKernelGGL((gemm_bf16x3_kernel<1, BNMTF_GEMM_NSET, 4>), grid, block, 0, st, a);
    else if (a.tw == 2) hipLaunchKernelGGL((gemm_bf16x3_kernel<2, BNMTF_GEMM_NSET, 2>), dim3(a.n_pad / 64, ns), block, 0, st, a);
    else          hipLaunchKernelGGL((gemm_bf16x3_kernel<2, BNMTF_GEMM_NSET, 4>), grid, block, 0, st, a);
    return;
  }
#ifdef BNMTF_EXPERIMENTS
  if (KP == 32) hipLaunchKernelGGL((gemm_kernel<1, 16, 2>), grid, block, 0, st, a);
  else          hipLaunchKernelGGL((gemm_kernel<2, 16, 1>), grid, block, 0, st, a);
#endif
}

}  // namespace bnmtf
