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
//  * at the end a wave forms d0 2^16 + d1 2^8 + d2 (exact in fp64), scales by 2^(e_c - 22) and writes its columns of the fp32
//    slab of the block's inner slice; the consumers (sweep prologue, vb_pieces_slabs_kernel) add the slabs in slab order as they
//    add the contraction's.
//  * B operand: bits[r / 32][u] -- one u32 per unit and 32 inner indices, the unit index contiguous (8 MB at 8192^2): the 32
//    lanes of a tile read 128 contiguous bytes; lane l takes the 16 bits of its half (l >> 5) of the step's 32 inner rows
//  * A operand: XB[plane][r / 16][col][r % 16] bytes: a lane reads the 16 inner rows of its half as one 16-byte load
//    (which k index of the instruction a given (half, byte) lands on is irrelevant: both operands use the same places)
//  * launch shape and the software pipeline of a step: see maskgemm_kernel.  History (cfg5): three bf16 planes
//    (fp32-exact as well, 6.7e-4 drift on the reference's toy trajectory against 1.5e-3 of the pair-panel sweep) 48 us; two planes
//    (16 bits) drifted 2.0e-3; int8 digit planes with every wave expanding its own mask bits 31 us; the expansion shared
//    through LDS, a deeper operand ring, an fp32 instead of an fp64 ending: 33-35 us each (none of them was the limit); every
//    non-product instruction of a step placed in the gaps between its products: 24 us.  (`-DBNMTF_MG_TIMING`: cycle and
//    wall-clock stamps around the loop; tools/micro/mfma_i8_rate.hip: the instruction's issue rate, 32 cycles like the bf16 form.)
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

// A block owns one (or, at 2 KP = 64, two) group(s) of 128 units (TW = 4 tiles), ALL 2 KP output columns and an inner slice: its
// four waves take the 32-column groups -- 128 units x 32 columns x three digit planes = 192 accumulator registers each (with 64
// columns per wave, 384, the compiler shuttles tiles between the two register files inside the loop).  The waves of a unit
// group need the SAME mask bytes: each expands its share of the four tiles (12 vector instructions per tile), puts it in LDS and
// all read the four fragments back.  One barrier per step (two fragment buffers).  Each wave writes its own columns of the slab of
// the block's inner slice: no cross-wave reduction.
template <int NCG>      // column groups of 32 = waves per unit group: 2 KP / 32 = 2 or 4
__global__ __launch_bounds__(256, 1) void maskgemm_kernel(MaskGemmArgs a) {
  constexpr int TW = 4, NSET = 6;                        // five steps of operands in flight (14 registers each): with two, a step waited for its loads
  __shared__ i32x4 bfrag[2][2][TW][64];                 // [buffer][unit group of the block][tile][lane]: 16 KiB

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int ncol = a.ncol;                               // 2 KP: 64 or 128
  constexpr int ncg = NCG;
  constexpr int ugpb = 4 / ncg;                          // unit groups per block: 2 or 1
  constexpr int tpw = TW / ncg;                          // tiles a wave expands per step: 2 or 1
  const int ug = wave / ncg, cg = wave % ncg;
  // block id -> (unit-group slot bx, inner slice s).  Blocks b and b + 8 share an XCD and its 4 MiB L2: the blocks of an XCD take
  // the same inner slice(s), so its L2 holds a 1 / split part of the digit planes
  const int nug = a.n_pad / (32 * TW), nx = (nug + ugpb - 1) / ugpb, nb = nx * a.split;
  int bx = (int)blockIdx.x % nx, s = (int)blockIdx.x / nx;
  if (nb % 8 == 0 && a.split <= 8 && 8 % a.split == 0) {
    const int g = 8 / a.split, xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3;
    s = xcd / g; bx = li * g + xcd % g;
  }
  const int ugi = bx * ugpb + ug;
  const bool live = ugi < nug;                           // (2 KP = 64 and an odd number of unit groups: the last block's second pair of waves goes through the motions)
  const int col0 = (live ? ugi : nug - 1) * (32 * TW);
  const int ipb = a.inner_per_wave;                      // inner rows of the BLOCK's slice, a multiple of 32
  const int r0 = s * ipb;

  // A fragments: plane p of step g: 16 bytes at XB + p * plane + (((r0 >> 4) + 2 g + h) * ncol + cg * 32 + c) * 4 words
  const size_t plane = (size_t)(a.rows_pad / 16) * ncol * 4;
  const uint32_t* ap = a.XB + ((size_t)((r0 >> 4) + h) * ncol + cg * 32 + c) * 4;
  const uint32_t* bp = a.bits + (size_t)(r0 >> 5) * a.n_pad + col0 + 32 * (cg * tpw) + c;      // the wave's own tile(s)

  i32x16 acc[3][TW];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[p][t][g] = 0;

  i32x4 araw[NSET][3];
  uint32_t wraw[NSET][2];
  auto load_step = [&](int g, i32x4 (&av)[3], uint32_t (&wv)[2]) {
#pragma unroll
    for (int p = 0; p < 3; ++p) av[p] = *reinterpret_cast<const i32x4*>(ap + p * plane + (size_t)(2 * g) * ncol * 4);
    wv[0] = bp[(size_t)g * a.n_pad];                       // (raw: a shift here would wait for the load it follows)
    wv[1] = bp[(size_t)g * a.n_pad + 32 * (tpw - 1)];      // (the wave's second tile, or the first again: no branch in the ring)
  };
  // this wave's share of the unit group's mask bytes of one step -> LDS buffer `buf`
  auto put = [&](int buf, const uint32_t (&wv)[2]) {
    auto one = [&](uint32_t w, int t) {
      const uint32_t w16 = w >> (16 * h);
      i32x4 b;
#pragma unroll
      for (int p = 0; p < 4; ++p) b[p] = mg_bits4(w16, 4 * p);
      bfrag[buf][ug][t][lane] = b;
    };
    one(wv[0], cg * tpw);
    if (tpw == 2) one(wv[1], cg * tpw + 1);
  };
  auto products = [&](const i32x4 (&av)[3], const i32x4 (&b)[TW], int t0, int t1) {
#pragma unroll
    for (int t = t0; t < t1; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) acc[p][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[p], b[t], acc[p][t], 0, 0, 0);
  };
  const int nsteps = ipb / 32;
  // the blocks of an XCD walk the SAME slice of the digit planes: every block starts somewhere else in its slice and wraps round
  // (integer sums do not care about the order)
  const int g0 = (int)(((unsigned)bx * 11u) % (unsigned)nsteps);
  auto at = [&](int i) { const int g = g0 + i; return g >= nsteps ? g - nsteps : g; };
  // Software pipeline (one wave per SIMD issues in order and blocks on the matrix pipe: whatever is not placed in the gaps
  // between a step's twelve products adds to it -- measured 900 cycles per step with the loads, the expansion and the exchange
  // ahead of the products, against 384 of matrix pipe):
  //   step i = barrier ; { products of step i } with, in their gaps: fragments of step i + 1 <- LDS (written during step i - 1,
  //   published by the barrier), operand loads of step i + NSET - 1, bytes of step i + 2 -> LDS.
  // A fragment buffer is rewritten one barrier after the reads of its previous contents were issued.  No branch in the steady
  // state: loads and expansions past the slice repeat its last step (their results are not used).
  const int last = nsteps - 1;
#ifdef BNMTF_MG_TIMING
  const unsigned long long tk0 = __builtin_readcyclecounter(), tw0 = wall_clock64();
#endif
#pragma unroll
  for (int j = 0; j < NSET - 1; ++j) load_step(at(j < last ? j : last), araw[j], wraw[j]);
  i32x4 bcur[TW], bnxt[TW];
  put(0, wraw[0]);
  put(1, wraw[1]);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < TW; ++t) bcur[t] = bfrag[0][ug][t][lane];
  auto step = [&](int i, const i32x4 (&av)[3], i32x4 (&anew)[3], uint32_t (&wnew)[2], const uint32_t (&w2)[2]) {
    __syncthreads();                                       // (LDS counter only: the operand loads of the steps ahead stay in flight)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TW; ++t) bnxt[t] = bfrag[(i + 1) & 1][ug][t][lane];
    const int gl = i + NSET - 1;
    load_step(at(gl < last ? gl : last), anew, wnew);
    put(i & 1, w2);                                        // step i + 2's bytes into the buffer step i's came from
    products(av, bcur, 0, TW);
#pragma unroll
    for (int q = 0; q < 3 * TW; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one product
      __builtin_amdgcn_sched_group_barrier(0x326, 5, 0);   // five of the others (vector / scalar ALU, loads, LDS reads and writes)
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TW; ++t) bcur[t] = bnxt[t];
  };
  // (set j holds step i = g + j; step i + 2's mask words are in set (j + 2) % NSET, loaded NSET - 3 steps ago; the new loads go
  // to set (j + NSET - 1) % NSET, which held step i - 1)
  static_assert(NSET >= 4, "the words of step i + 2 must not be the set being reloaded");
  int g = 0;
  for (; g + NSET <= nsteps; g += NSET) {
#pragma unroll
    for (int j = 0; j < NSET; ++j) step(g + j, araw[j], araw[(j + NSET - 1) % NSET], wraw[(j + NSET - 1) % NSET], wraw[(j + 2) % NSET]);
  }
#pragma unroll
  for (int j = 0; j < NSET - 1; ++j)            // (every wave of the block runs the same number of steps: the barriers match)
    if (g + j < nsteps) step(g + j, araw[j], araw[(j + NSET - 1) % NSET], wraw[(j + NSET - 1) % NSET], wraw[(j + 2) % NSET]);

#ifdef BNMTF_MG_TIMING
  const unsigned long long tk1 = __builtin_readcyclecounter(), tw1 = wall_clock64();
#endif
  if (live) {
    // D: reg g, lane l -> column i = (g & 3) + 8 (g >> 2) + 4 h of the tile, unit j = c.  d0 2^16 + (d1 2^8 + d2), each digit sum
    // exact as an fp32 (< 2^24), the inner sum off by < 2^-32 of the total, one rounding in the outer FMA; the scale is a power of
    // two.  (In fp64 -- three quarter-rate conversions per value -- this end was ~8 us of the kernel's 35.)
    float* out = a.slabs + (size_t)s * a.n_pad * ncol + cg * 32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int colb = cg * 32 + 8 * g4 + 4 * h;
      int sc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) sc[q] = a.cexp[colb + q] - 22;
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gg = 4 * g4 + q;
          const float n = fmaf((float)acc[0][t][gg], 65536.0f, fmaf((float)acc[1][t][gg], 256.0f, (float)acc[2][t][gg]));
          v[q] = __builtin_ldexpf(n, sc[q]);               // (v_ldexp_f32: exact, whatever the column's size)
        }
        *reinterpret_cast<f32x4*>(out + (size_t)(col0 + 32 * t + c) * ncol + 8 * g4 + 4 * h) = v;
      }
    }
  }
#ifdef BNMTF_MG_TIMING
  const unsigned long long tk2 = __builtin_readcyclecounter(), tw2 = wall_clock64();
  if ((blockIdx.x % 37) == 0 && threadIdx.x == 0)
    printf("maskgemm block %d: loop %llu cycles / %llu wall ticks (100 MHz), end %llu cycles / %llu ticks, %d steps\n", (int)blockIdx.x, tk1 - tk0, tw1 - tw0, tk2 - tk1, tw2 - tw1, nsteps);
#endif
}

void launch_maskgemm(const MaskGemmArgs& a, int KP, hipStream_t st) {
  if (a.split <= 0 || a.n_pad <= 0) return;
  MaskGemmArgs b = a;
  b.ncol = 2 * KP;
  const int ugpb = 4 / (b.ncol / 32), nug = a.n_pad / 128;
  dim3 grid(((nug + ugpb - 1) / ugpb) * a.split), block(256);
  if (b.ncol == 64) hipLaunchKernelGGL(maskgemm_kernel<2>, grid, block, 0, st, b);
  else hipLaunchKernelGGL(maskgemm_kernel<4>, grid, block, 0, st, b);
}

}  // namespace bnmtf
