// VB: the two sums of a half sweep that run over a unit's MISSING entries but do not depend on the sequential chain,
//   asq[u][k] = sum_{r in miss(u)} S2o[r][k]      (tau_uk = exptau * (colsum2_k - asq_uk):   bnmf_vb_optimised.py:189-199)
//   vsq[u][k] = sum_{r in miss(u)} Eo[r][k]^2     (the unit's own term in the numerator, and the exp_square_diff pieces)
// as ONE dense product on the matrix cores: out[u][0:2KP] = sum_r bit[u][r] * [S2o | Eo^2][r][:], the mask as one BIT per
// entry.  The variational sweep gathered (E, S2) pairs and formed both sums per column inside its sequential loop -- 4.5 vector
// instructions and 8 LDS bytes per (entry, column) against 1.5 and 4 for the Gibbs sweep; with the sums taken here the loop
// gathers E only and keeps one running sum (sweep_chip.inc, MODE = kSweepVB).
//
// Integer arithmetic (v_mfma_i32_32x32x32_i8: twice the bf16 rate, and EXACT):
//  * the moments are non-negative, so column c is put on a fixed-point grid of 22 bits below 2^e_c > max_r x[r][c]
//    (vb_colmax_kernel): n = rint(x 2^(22 - e_c)), written as three balanced base-256 digits n = d0 2^16 + d1 2^8 + d2,
//    d1, d2 in [-128, 127] (vb_planes_kernel).  An element is off by at most 2^(e_c - 23), whatever its size; a sum over ~800
//    missing entries by ~1e-8 of itself -- below the rounding of the fp32 result;
//  * the mask bits become bytes 0 / 1 in registers (B operand); the three digit planes (A operand) accumulate in three
//    i32 tiles -- integer sums: no rounding, no dependence on the order of the additions or on how the inner range is cut;
//  * after the cross-wave reduction (LDS, integers) wave 0 forms d0 2^16 + d1 2^8 + d2 (exact in fp64), scales by
//    2^(e_c - 22) and writes ONE fp32 slab per inner slice; the consumers (sweep prologue, vb_pieces_slabs_kernel) add
//    the slabs in slab order as they add the contraction's.
//  * B operand: bits[r / 32][u] -- one u32 per unit and 32 inner indices, the unit index contiguous (8 MB at 8192^2): the 32
//    lanes of a tile read 128 contiguous bytes; lane l takes the 16 bits of its half (l >> 5) of the step's 32 inner rows
//  * A operand: XB[plane][r / 16][col][r % 16] bytes: a lane reads the 16 inner rows of its half as one 16-byte load
//    (which k index of the instruction a given (half, byte) lands on is irrelevant: both operands use the same places)
//  * launch shape as K1/K2 (kernel_gemm.hip): a wave owns 128 units x 64 output columns and a private inner slice, the four
//    waves of a block reduce through LDS.  History: three bf16 planes (fp32-exact as well, 6.7e-4 drift on the reference's toy
//    trajectory against 1.5e-3 of the pair-panel sweep) took 48 us at cfg5; two planes (16 bits) drifted 2.0e-3.
#include <cstdlib>

#include "kernels.h"

namespace bnmtf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2c __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mg_pack_rne(float a0, float a1) {
  f32x2c v; v.x = a0; v.y = a1;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// per column c of [S2 | E^2]: the largest element's fp32 bits (non-negative floats order like unsigned integers).  256 blocks
// (with 64 the kernel is the latency of 64 dependent trips per thread: 23 us), the sub-rows of a block combined through LDS,
// one atomic per block and column (an atomic per thread: 15 us).  The fall-back when the relayout has not left the maxima
// (PostArgs::umax): it costs a memset and this launch, ~12 us.
__global__ __launch_bounds__(256) void vb_colmax_kernel(const float* S2, const float* E, int rows, int KP, unsigned* umax) {
  __shared__ float part[256];
  const int ncol = 2 * KP;
  const int col = threadIdx.x % ncol, sub = threadIdx.x / ncol, nsub = 256 / ncol;       // ncol = 64 or 128
  float m = 0.f;
  const float* src = col < KP ? S2 + col : E + col - KP;
  const int r0 = blockIdx.x * nsub + sub, step = gridDim.x * nsub;
  for (int r = r0; r < rows; r += 4 * step) {           // four rows in flight per trip
    float x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = r + q * step < rows ? src[(size_t)(r + q * step) * KP] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) m = fmaxf(m, col < KP ? x[q] : x[q] * x[q]);
  }
  part[threadIdx.x] = m;
  __syncthreads();
  if (sub == 0) {
    for (int q = 1; q < nsub; ++q) m = fmaxf(m, part[q * ncol + col]);
    atomicMax(umax + col, __builtin_bit_cast(unsigned, m));
  }
}

// XB[plane (d0, d1, d2)][r16][col][16 bytes] from S2 (columns [0, KP)) and E^2 (columns [KP, 2 KP)); rows >= `rows` are zero.
// One thread per (r16, col); cexp[col] = e_c is written by the threads of r16 == 0.
__global__ __launch_bounds__(256) void vb_planes_kernel(const float* S2, const float* E, int rows, int rows_pad, int KP, const unsigned* umax,
                                                         int* cexp, uint32_t* XB) {
  const int ncol = 2 * KP;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int r16 = t / ncol, col = t % ncol;
  if (r16 * 16 >= rows_pad) return;
  // 2^e > max: exponent field of the largest element + 1 (a zero column: e = 0)
  const unsigned mb = umax[col];
  const int e = mb ? (int)((mb >> 23) & 255u) - 126 : 0;
  if (r16 == 0) cexp[col] = e;
  u32x4 d0 = {0u, 0u, 0u, 0u}, d1 = d0, d2 = d0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int r = r16 * 16 + j;
    float x = 0.f;
    if (r < rows) {
      if (col < KP) x = S2[(size_t)r * KP + col];
      else { const float y = E[(size_t)r * KP + col - KP]; x = y * y; }
    }
    const int n = __float2int_rn(ldexpf(x, 22 - e));              // 0 .. 2^22
    const int b2 = ((n + 128) & 255) - 128;
    const int n1 = (n - b2) >> 8;
    const int b1 = ((n1 + 128) & 255) - 128;
    const int b0 = (n1 - b1) >> 8;                                  // 0 .. 64
    const int sh = 8 * (j & 3);
    d0[j >> 2] |= (uint32_t)(b0 & 255) << sh;
    d1[j >> 2] |= (uint32_t)(b1 & 255) << sh;
    d2[j >> 2] |= (uint32_t)(b2 & 255) << sh;
  }
  const size_t plane = (size_t)(rows_pad / 16) * ncol * 4;      // u32 words per plane
  const size_t o = ((size_t)r16 * ncol + col) * 4;
  *reinterpret_cast<u32x4*>(XB + o) = d0;
  *reinterpret_cast<u32x4*>(XB + plane + o) = d1;
  *reinterpret_cast<u32x4*>(XB + 2 * plane + o) = d2;
}
void launch_vb_planes(const float* S2, const float* E, int rows, int rows_pad, int KP, unsigned* umax, bool have_max, int* cexp, uint32_t* XB, hipStream_t st) {
  const long total = (long)(rows_pad / 16) * 2 * KP;
  if (total <= 0) return;
  if (!have_max) {
    (void)hipMemsetAsync(umax, 0, sizeof(unsigned) * 2 * KP, st);
    hipLaunchKernelGGL(vb_colmax_kernel, dim3(256), dim3(256), 0, st, S2, E, rows, KP, umax);
  }
  hipLaunchKernelGGL(vb_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, S2, E, rows, rows_pad, KP, (const unsigned*)umax, cexp, XB);
}

// bits[w][ul] for the local units ul < n_pad (rows of padding: zero), w < ldw: bit b = entry (unit0 + ul, 32 w + b) is missing
__global__ __launch_bounds__(256) void mask_bits_kernel(const uint8_t* M, int I, int J, int by_rows, int unit0, int n, int m, int n_pad, int ldw, uint32_t* bits) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)n_pad * ldw) return;
  const int w = (int)(t / n_pad), ul = (int)(t % n_pad);
  uint32_t word = 0;
  if (ul < n) {
    const int u = unit0 + ul;
#pragma unroll 4
    for (int b = 0; b < 32; ++b) {
      const int r = 32 * w + b;
      if (r < m) {
        const uint8_t mv = by_rows ? M[(size_t)u * J + r] : M[(size_t)r * J + u];
        word |= (mv == 0 ? 1u : 0u) << b;
      }
    }
  }
  bits[t] = word;
}
void launch_mask_bits(const uint8_t* M, int I, int J, int by_rows, int unit0, int n, int m, int n_pad, int ldw, uint32_t* bits, hipStream_t st) {
  const long total = (long)n_pad * ldw;
  if (total <= 0) return;
  hipLaunchKernelGGL(mask_bits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, M, I, J, by_rows, unit0, n, m, n_pad, ldw, bits);
}

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// four mask bits (b .. b + 3 of w) as four bytes 0 / 1: x * (1 + 2^7 + 2^14 + 2^21) puts bit i at 8 i (and elsewhere)
__device__ __forceinline__ int mg_bits4(uint32_t w, int b) {
  return (int)(__umul24(__builtin_amdgcn_ubfe(w, b, 4), 0x204081u) & 0x01010101u);
}

// A wave owns 128 units (TW = 4 tiles) x 32 output columns (one "column group" of the 2 KP) x three digit planes -- 192
// accumulator registers: with 64 columns (384) the compiler shuttles tiles between the two register files inside the loop --
// and a private inner slice; grid = unit groups x column groups x inner slices.
__global__ __launch_bounds__(256, 1) void maskgemm_kernel(MaskGemmArgs a) {
  constexpr int MT = 1, TW = 4, NSET = 3;
  constexpr int NRED = MT * TW * 16;                     // one plane per pass of the LDS reduction (64 KiB of LDS)
  __shared__ int red[2][NRED * 64];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int ncol = a.ncol;                               // 2 KP
  // block id -> (unit group bx, column group cg, inner slice s).  Blocks b and b + 8 share an XCD and its 4 MiB L2: the blocks
  // of an XCD take the same inner slice(s), so its L2 holds a 1 / split part of the digit planes
  const int nxg = a.n_pad / (32 * TW), ncg = ncol / (32 * MT), nx = nxg * ncg, nb = nx * a.split;
  int bq = (int)blockIdx.x % nx, s = (int)blockIdx.x / nx;
  if (nb % 8 == 0 && a.split <= 8 && 8 % a.split == 0) {
    const int g = 8 / a.split, xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3;
    s = xcd / g; bq = li * g + xcd % g;
  }
  const int bx = bq / ncg, cg = bq % ncg;
  const int col0 = bx * (32 * TW);
  const int ipw = a.inner_per_wave;                      // multiple of 32
  const int r0 = (s * 4 + wave) * ipw;

  // A fragments: plane p, tile mt of step g: 16 bytes at XB + p * plane + (((r0 >> 4) + 2 g + h) * ncol + cg * 64 + mt * 32 + c) * 4 words
  const size_t plane = (size_t)(a.rows_pad / 16) * ncol * 4;
  const uint32_t* ap = a.XB + ((size_t)((r0 >> 4) + h) * ncol + cg * (32 * MT) + c) * 4;
  const uint32_t* bp = a.bits + (size_t)(r0 >> 5) * a.n_pad + col0 + c;

  i32x16 acc[3][MT][TW];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[p][mt][t][g] = 0;

  i32x4 araw[NSET][3][MT];
  uint32_t wraw[NSET][TW];
  auto load_step = [&](int g, i32x4 (&av)[3][MT], uint32_t (&wv)[TW]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[p][mt] = *reinterpret_cast<const i32x4*>(ap + p * plane + ((size_t)(2 * g) * ncol + mt * 32) * 4);
#pragma unroll
    for (int t = 0; t < TW; ++t) wv[t] = bp[(size_t)g * a.n_pad + 32 * t];      // (raw: a shift here would wait for the load it follows)
  };
  auto expand = [&](uint32_t w16) {
    i32x4 b;
#pragma unroll
    for (int p = 0; p < 4; ++p) b[p] = mg_bits4(w16, 4 * p);
    return b;
  };
  auto mul_step = [&](const i32x4 (&av)[3][MT], const uint32_t (&wv)[TW]) {
    // tile t's MFMAs with the expansion of tile t + 1's mask bits in their gaps (an MFMA holds the vector issue for 8 of its 32
    // cycles: three 4-cycle instructions per gap are free)
    i32x4 b[2];
    b[0] = expand(wv[0] >> (16 * h));
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      if (t + 1 < TW) b[(t + 1) & 1] = expand(wv[t + 1] >> (16 * h));
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[p][mt][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[p][mt], b[t & 1], acc[p][mt][t], 0, 0, 0);
      if (t + 1 < TW) {
#pragma unroll
        for (int i = 0; i < 3 * MT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // four vector instructions
        }
      }
    }
  };
  const int nsteps = ipw / 32;
  // ring of NSET operand sets, NSET - 1 steps in flight behind the one being multiplied.  No branch inside the steady state: a
  // load that would run past the slice re-reads its last step instead (with the loads behind conditions the compiler's wait
  // counts at the join points drained the ring every step)
  const int last = nsteps - 1;
#pragma unroll
  for (int j = 0; j < NSET - 1; ++j) load_step(j < last ? j : last, araw[j], wraw[j]);
  int g = 0;
  for (; g + NSET <= nsteps; g += NSET) {
#pragma unroll
    for (int j = 0; j < NSET; ++j) {
      const int gl = g + j + NSET - 1;
      load_step(gl < last ? gl : last, araw[(j + NSET - 1) % NSET], wraw[(j + NSET - 1) % NSET]);
      __builtin_amdgcn_sched_barrier(0);      // (nothing of a later step -- the cheap shifts of its mask words, say -- is scheduled
      mul_step(araw[j], wraw[j]);             //  up here, where it would wait for loads that have only just been issued)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int j = 0; j < NSET - 1; ++j)
    if (g + j < nsteps) mul_step(araw[j], wraw[j]);

  // cross-wave tree reduction through LDS (integers: exact), one digit plane at a time: (2,3) -> (0,1), then 1 -> 0
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    auto put = [&](int* dst) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int g = 0; g < 16; ++g) dst[((mt * TW + t) * 16 + g) * 64 + lane] = acc[p][mt][t][g];
    };
    auto add = [&](const int* src) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int g = 0; g < 16; ++g) acc[p][mt][t][g] += src[((mt * TW + t) * 16 + g) * 64 + lane];
    };
    if (p > 0) __syncthreads();
    if (wave >= 2) put(red[wave - 2]);
    __syncthreads();
    if (wave < 2) add(red[wave]);
    __syncthreads();
    if (wave == 1) put(red[0]);
    __syncthreads();
    if (wave == 0) add(red[0]);
  }
  if (wave == 0) {
    // D: reg g, lane l -> column i = (g & 3) + 8 (g >> 2) + 4 h of the tile, unit j = c
    float* out = a.slabs + (size_t)s * a.n_pad * ncol + cg * (32 * MT);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int colb = cg * (32 * MT) + mt * 32 + 8 * g4 + 4 * h;
        double sc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sc[q] = __builtin_ldexp(1.0, a.cexp[colb + q] - 22);
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int gg = 4 * g4 + q;
            const double n = (double)acc[0][mt][t][gg] * 65536.0 + (double)acc[1][mt][t][gg] * 256.0 + (double)acc[2][mt][t][gg];
            v[q] = (float)(n * sc[q]);
          }
          *reinterpret_cast<f32x4*>(out + (size_t)(col0 + 32 * t + c) * ncol + mt * 32 + 8 * g4 + 4 * h) = v;
        }
      }
  }
}

void launch_maskgemm(const MaskGemmArgs& a, int KP, hipStream_t st) {
  if (a.split <= 0 || a.n_pad <= 0) return;
  MaskGemmArgs b = a;
  b.ncol = 2 * KP;
  dim3 grid((a.n_pad / 128) * (b.ncol / 32) * a.split), block(256);
  hipLaunchKernelGGL(maskgemm_kernel, grid, block, 0, st, b);
}

}  // namespace bnmtf
