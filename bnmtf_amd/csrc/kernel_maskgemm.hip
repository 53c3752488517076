// VB: the two sums of a half sweep that run over a unit's MISSING entries but do not depend on the sequential chain,
//   asq[u][k] = sum_{r in miss(u)} S2o[r][k]      (tau_uk = exptau * (colsum2_k - asq_uk):   bnmf_vb_optimised.py:189-199)
//   vsq[u][k] = sum_{r in miss(u)} Eo[r][k]^2     (the unit's own term in the numerator, and the exp_square_diff pieces)
// as ONE dense product on the bf16 matrix cores: out[u][0:2KP] = sum_r bit[u][r] * [S2o | Eo^2][r][:], the mask as one BIT per
// entry.  The variational sweep gathered (E, S2) pairs and formed both sums per column inside its sequential loop -- 4.5 vector
// instructions and 8 LDS bytes per (entry, column) against 1.5 and 4 for the Gibbs sweep; with the sums taken here the loop
// gathers E only and keeps one running sum (sweep_chip.inc, MODE = kSweepVB).
//
//  * B operand (the mask): bits[r / 32][u] -- one u32 per unit and 32 inner indices, the unit index contiguous (8 MB at
//    8192^2): the 32 lanes of a tile read 128 contiguous bytes (unit-major rows cost 32 cache lines per load: 51 us against
//    the time below), each lane the word of its own unit, expanded in registers to bf16 0 / 1:
//      lane l holds B[k = 8*(l>>5) .. +7][j = l&31] = bit (r0 + 8*(l>>5) + e) of unit col0 + 32*t + (l&31)
//  * A operand (the moments): every fp32 value as its THREE bf16 terms hi + mid + lo (round-to-nearest residuals as in
//    kernel_gemm.hip: 24 significant bits; the products with 0 / 1 are exact and the accumulation is fp32, so the sums are what
//    an fp32 loop over the missing entries gives, up to the order of the additions.  Two planes -- 16 bits -- measured 1.5 x
//    the drift of the old sweep on the reference's 20-iteration toy trajectory), pre-split once per half sweep by
//    vb_planes_kernel into the fragment layout
//      XB[plane][r / 8][col][r % 8]   (lane l reads its 8 inner rows of column mt*32 + (l&31) as one 16-byte load)
//  * the launch shape, the per-wave inner slices, the LDS tree reduction and the slab layout are those of K1/K2
//    (kernel_gemm.hip); the consumer (sweep prologue, vb_pieces_kernel) adds the slabs in slab order.
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

// XB[plane (hi, mid, lo)][r8][col][8] from S2 (columns [0, KP)) and E^2 (columns [KP, 2 KP)); rows >= `rows` are zero.
// One thread per (r8, col); a wave reads 64 consecutive columns of 8 rows (coalesced) and writes 64 x 16 bytes per plane.
__global__ __launch_bounds__(256) void vb_planes_kernel(const float* S2, const float* E, int rows, int rows_pad, int KP, uint32_t* XB) {
  const int ncol = 2 * KP;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int r8 = t / ncol, col = t % ncol;
  if (r8 * 8 >= rows_pad) return;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int r = r8 * 8 + e;
    float x = 0.f;
    if (r < rows) {
      if (col < KP) x = S2[(size_t)r * KP + col];
      else { const float y = E[(size_t)r * KP + col - KP]; x = y * y; }
    }
    v[e] = x;
  }
  u32x4 hi, mid, lo;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a0 = v[2 * p], a1 = v[2 * p + 1];
    const uint32_t h = mg_pack_rne(a0, a1);
    hi[p] = h;
    const float b0 = a0 - __builtin_bit_cast(float, h << 16), b1 = a1 - __builtin_bit_cast(float, h & 0xffff0000u);     // exact
    const uint32_t m = mg_pack_rne(b0, b1);
    mid[p] = m;
    lo[p] = mg_pack_rne(b0 - __builtin_bit_cast(float, m << 16), b1 - __builtin_bit_cast(float, m & 0xffff0000u));
  }
  const size_t plane = (size_t)(rows_pad / 8) * ncol * 4;       // u32 words per plane
  const size_t o = ((size_t)r8 * ncol + col) * 4;
  *reinterpret_cast<u32x4*>(XB + o) = hi;
  *reinterpret_cast<u32x4*>(XB + plane + o) = mid;
  *reinterpret_cast<u32x4*>(XB + 2 * plane + o) = lo;
}
void launch_vb_planes(const float* S2, const float* E, int rows, int rows_pad, int KP, uint32_t* XB, hipStream_t st) {
  const long total = (long)(rows_pad / 8) * 2 * KP;
  if (total <= 0) return;
  hipLaunchKernelGGL(vb_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, S2, E, rows, rows_pad, KP, XB);
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

// two elements (bits b, b + 1 of w) as packed bf16 0 / 1
__device__ __forceinline__ uint32_t mg_bits2(uint32_t w, int b) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)w, b, 1) & 0x3F80u;            // v_bfe_i32: 0 or -1
  return ((uint32_t)__builtin_amdgcn_sbfe((int)w, b + 1, 1) & 0x3F800000u) | lo;
}

// A wave owns 256 units (TW = 8 tiles) x 64 output columns (MT = 2 tiles: one "column group" of the 2 KP) and a private inner
// slice; grid = unit groups x column groups x inner slices.  (128 units x 128 columns per wave -- the same 256 accumulator
// registers -- read every byte of the moments once per 128 units: 403 MB from L2 per launch and 47 us; this shape reads half.)
template <int TW, int DBG = 0>     // DBG (tools only, BNMTF_MG_DBG): 1 = the loads without the products, 2 = the products without the loads
__global__ __launch_bounds__(256, 1) void maskgemm_kernel(MaskGemmArgs a) {
  constexpr int MT = 2, NSET = 3;
  constexpr int TH = TW / 2;                             // unit tiles per pass of the LDS reduction (64 KiB of LDS)
  constexpr int NRED = MT * TH * 16;
  __shared__ float red[2][NRED * 64];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int ncol = a.ncol;                               // 2 KP
  // block id -> (unit group bx, column group cg, inner slice s).  Blocks b and b + 8 share an XCD and its 4 MiB L2: the blocks
  // of an XCD take the same inner slice(s), so its L2 holds a 1 / split part of the moments
  const int nxg = (a.n_pad + 32 * TW - 1) / (32 * TW), ncg = ncol / 64, nx = nxg * ncg, nb = nx * a.split;
  int bq = (int)blockIdx.x % nx, s = (int)blockIdx.x / nx;
  if (nb % 8 == 0 && a.split <= 8 && 8 % a.split == 0) {
    const int g = 8 / a.split, xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3;
    s = xcd / g; bq = li * g + xcd % g;
  }
  const int bx = bq / ncg, cg = bq % ncg;
  const int col0 = bx * (32 * TW);
  const int ipw = a.inner_per_wave;                      // multiple of 32
  const int r0 = (s * 4 + wave) * ipw;

  // A fragments: plane p, tile mt of step g: 16 bytes at XB + p * plane + (((r0 >> 3) + 2 g + h) * ncol + cg * 64 + mt * 32 + c) * 4 words
  const size_t plane = (size_t)(a.rows_pad / 8) * ncol * 4;
  const uint32_t* ap = a.XB + ((size_t)((r0 >> 3) + h) * ncol + cg * 64 + c) * 4;
  // the mask words of this wave's inner slice, one per tile and 32 inner rows (a tile beyond n_pad re-reads the last one: its
  // results are not stored)
  const uint32_t* bp = a.bits + (size_t)(r0 >> 5) * a.n_pad;
  int ucol[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t) ucol[t] = min(col0 + 32 * t, a.n_pad - 32) + c;

  f32x16 acc[MT][TW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[mt][t][g] = 0.0f;

  u32x4 araw[NSET][3][MT];
  uint32_t wraw[NSET][TW];
  auto load_step = [&](int g, u32x4 (&av)[3][MT], uint32_t (&wv)[TW]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[p][mt] = *reinterpret_cast<const u32x4*>(ap + p * plane + ((size_t)(2 * g) * ncol + mt * 32) * 4);
#pragma unroll
    for (int t = 0; t < TW; ++t) wv[t] = bp[(size_t)(g >> 1) * a.n_pad + ucol[t]];      // (raw: a shift here would wait for the load it follows)
  };
  auto expand = [&](uint32_t w8) {
    u32x4 b;
#pragma unroll
    for (int p = 0; p < 4; ++p) b[p] = mg_bits2(w8, 2 * p);
    return b;
  };
  auto mul_step = [&](int g, const u32x4 (&av)[3][MT], const uint32_t (&wv)[TW]) {
    const int sh = 8 * h + 16 * (g & 1);                  // the lane's byte of this step
    // tile t's MFMAs with the expansion of tile t + 1's mask bits in their gaps (an MFMA holds the vector issue for 8 of its 32
    // cycles: three 4-cycle instructions per gap are free)
    u32x4 b[2];
    b[0] = expand(wv[0] >> sh);
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      if (t + 1 < TW) b[(t + 1) & 1] = expand(wv[t + 1] >> sh);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x16 d = acc[mt][t];
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[2][mt]), __builtin_bit_cast(bf16x8, b[t & 1]), d, 0, 0, 0);   // small terms first
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[1][mt]), __builtin_bit_cast(bf16x8, b[t & 1]), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[0][mt]), __builtin_bit_cast(bf16x8, b[t & 1]), d, 0, 0, 0);
        acc[mt][t] = d;
      }
      if (t + 1 < TW) {
#pragma unroll
        for (int i = 0; i < 3 * MT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // three vector instructions
        }
      }
    }
  };
  const int nsteps = ipw / 16;
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
      if (DBG != 2) load_step(gl < last ? gl : last, araw[(j + NSET - 1) % NSET], wraw[(j + NSET - 1) % NSET]);
      __builtin_amdgcn_sched_barrier(0);      // (nothing of a later step -- the cheap shifts of its mask words, say -- is scheduled
      if (DBG != 1) mul_step(g + j, araw[j], wraw[j]);      //  up here, where it would wait for loads that have only just been issued)
      else {
        uint32_t x = 0;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) x ^= araw[j][p][mt][0] ^ araw[j][p][mt][3];
#pragma unroll
        for (int t = 0; t < TW; ++t) x ^= wraw[j][t];
        acc[0][0][0] += (float)x;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int j = 0; j < NSET - 1; ++j)
    if (g + j < nsteps) mul_step(g + j, araw[j], wraw[j]);

  // cross-wave tree reduction through LDS, TH unit tiles at a time: (2,3) -> (0,1), then 1 -> 0; wave 0 writes the slab
  float* out = a.slabs + (size_t)s * a.n_pad * ncol + cg * 64;
#pragma unroll
  for (int t0 = 0; t0 < TW; t0 += TH) {
    auto put = [&](float* dst) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
          for (int g = 0; g < 16; ++g) dst[((mt * TH + t) * 16 + g) * 64 + lane] = acc[mt][t0 + t][g];
    };
    auto add = [&](const float* src) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
          for (int g = 0; g < 16; ++g) acc[mt][t0 + t][g] += src[((mt * TH + t) * 16 + g) * 64 + lane];
    };
    if (t0 > 0) __syncthreads();
    if (wave >= 2) put(red[wave - 2]);
    __syncthreads();
    if (wave < 2) add(red[wave]);
    __syncthreads();
    if (wave == 1) put(red[0]);
    __syncthreads();
    if (wave == 0) {
      add(red[0]);
      // D: reg g, lane l -> column i = (g & 3) + 8 (g >> 2) + 4 h of the tile, unit j = c
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < TH; ++t) {
          const int un = col0 + 32 * (t0 + t) + c;
          if (col0 + 32 * (t0 + t) < a.n_pad) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              f32x4 v = {acc[mt][t0 + t][4 * g4], acc[mt][t0 + t][4 * g4 + 1], acc[mt][t0 + t][4 * g4 + 2], acc[mt][t0 + t][4 * g4 + 3]};
              *reinterpret_cast<f32x4*>(out + (size_t)un * ncol + mt * 32 + 8 * g4 + 4 * h) = v;
            }
          }
        }
    }
  }
}

void launch_maskgemm(const MaskGemmArgs& a, int KP, hipStream_t st) {
  if (a.split <= 0 || a.n_pad <= 0) return;
  MaskGemmArgs b = a;
  b.ncol = 2 * KP;
  dim3 grid(((a.n_pad + 255) / 256) * (b.ncol / 64) * a.split), block(256);
  const char* e = getenv("BNMTF_MG_DBG");
  const int dbg = e ? atoi(e) : 0;
  if (dbg == 1) hipLaunchKernelGGL((maskgemm_kernel<8, 1>), grid, block, 0, st, b);
  else if (dbg == 2) hipLaunchKernelGGL((maskgemm_kernel<8, 2>), grid, block, 0, st, b);
  else hipLaunchKernelGGL((maskgemm_kernel<8>), grid, block, 0, st, b);
}

}  // namespace bnmtf
