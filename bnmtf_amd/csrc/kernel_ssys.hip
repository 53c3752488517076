// The S step of the tri-factorisation as one dense K.L x K.L system (bnmtf_gibbs_optimised.py:157-160, 201-205;
// bnmtf_vb_optimised.py:172-176, 252-262).
//
// Given F and G, the K.L entries of S have a jointly Gaussian (truncated) conditional with
//   precision  tau * A ,   A[(k,l),(k',l')] = sum_ij M_ij F_ik G_jl F_ik' G_jl'
//   linear term        b[(k,l)] = sum_ij M_ij R_ij F_ik G_jl = sum_j Pv_jk G_jl        (Pv = R~^T F, the contraction)
// and the reference's scalar updates are coordinate steps on it:
//   tauS_kl = tau A_aa ,   muS_kl = (-lambdaS_kl + tau (b_a - sum_{a' != a} A_aa' S_a')) / tauS_kl ,   a = k L + l.
// A factorises over the columns of R:  A = sum_j W_j (x) (G_j G_j^T),  W_j[k][k'] = sum_i M_ij F_ik F_ik'
//                                         = (F^T F)[k][k'] - sum_{i in miss(j)} F_ik F_ik'   (a K x K matrix per column).
// W_j and G_j G_j^T are symmetric: with their upper triangles packed (p = (k <= k'), r = (l <= l'); tri_index, kernels.h)
// the whole system is one plain GEMM, A[p][r] = sum_j Wc[j][p] Gc[j][r], written to its four symmetric places.
// So an iteration costs one masked Gram per column (scol_gram_kernel: f32 MFMA over the ~10 % missing entries), the
// packed second moments of G's rows (gamma_pack_kernel), one K(K+1)/2 x J x L(L+1)/2 GEMM (ssys_gemm_kernel: f32 MFMA,
// partial slabs over column ranges, summed in a fixed order by ssys_reduce_kernel), and then the
// K.L sequential conditionals touch nothing but A: ssys_chain_kernel keeps the residual r = b - A S on chip and walks the
// entries row by row (a row of S = the lanes of one wave, the running correction of a lane grows by one FMA per step).
// The variational version (second moments) is the same system with E[F_ik F_ik'] = F_ik F_ik' + [k = k'] varF_ik and
// E[G_jl G_jl'] likewise: pass varF / varG.  Sharded over GPUs, W_j and the GEMM cover a rank's own columns and (A, b) is
// summed with ONE all-reduce -- the "K x L Gram" exchange -- after which every rank walks the same chain.
// K, L <= 32 (one 32 x 32 MFMA tile per column / per row of A); larger ranks keep the per-row path of kernel_bnmtf.hip.
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "sweep_common.h"

namespace bnmtf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// W~_j = C~f - sum_{i in miss(j)} (F_i F_i^T + diag(varF_i)) for every local column j: one wave per column, the 32 x 32
// tile in MFMA accumulators.  v_mfma_f32_32x32x2_f32 takes A[i][k] and B[k][j] from lane (i or j) + 32 k: for the Gram of
// two entries both operands are the same register, F[entry of this half][lane & 31] -- and since the Gram is a sum over
// the entries, which entry goes to which (step, half) is free: each half takes four CONSECUTIVE slots per 16-byte index
// load.  What the loop costs besides its MFMAs is what matters (a vector instruction between MFMAs adds its 4 cycles to
// their 64, the texture path takes ~9 cycles per wave-level load): per step one shift-add (row offset), one row load.
// Everything else sits in buffer descriptors, scalar offsets and immediates.  Rows are loaded a batch (8 steps) ahead,
// indices two.
// (the three-term bf16 split of the round-6 matrix-core kernels below: ssys_gemm_bf16_kernel's comment)
namespace {
typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 g_bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t g_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t g_pack_rne(float a0, float a1) {
  f32x2 v; v.x = a0; v.y = a1;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, g_bf16x2));
}
__device__ __forceinline__ void g_split3(const float (&v)[8], g_u32x4& hi, g_u32x4& mid, g_u32x4& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a0 = v[2 * p], a1 = v[2 * p + 1];
    const uint32_t h = g_pack_rne(a0, a1);
    hi[p] = h;
    const float b0 = a0 - __builtin_bit_cast(float, h << 16);                 // exact
    const float b1 = a1 - __builtin_bit_cast(float, h & 0xffff0000u);
    const uint32_t m = g_pack_rne(b0, b1);
    mid[p] = m;
    const float c0 = b0 - __builtin_bit_cast(float, m << 16);                 // exact
    const float c1 = b1 - __builtin_bit_cast(float, m & 0xffff0000u);
    lo[p] = g_pack_rne(c0, c1);
  }
}
}  // namespace
__device__ __forceinline__ void gamma_pack_body(const GammaPackArgs& a, int block);
// (blocks behind the column Grams' pack the second moments of G's rows -- gamma_pack_body, below: the two do not depend on each
// other, and one launch less is ~5 us of the S step)
__device__ __forceinline__ void ssys_b_body(const SSysBArgs& a, int block);
// BF = 1 (round 6): the outer products on the bf16 matrix cores, fp32-exact -- a half's eight rows of a 16-slot step ARE the A (and
// B) operand of v_mfma_f32_32x32x16_bf16 (lane (c, g) holds k = 8 g .. 8 g + 7), so the loads stay as they are and a step is one
// three-term split of eight registers (g_split3) and six products (6 x 32 cycles) instead of eight f32 products (8 x 64).
template <int VB, int BF = 0>
__global__ __launch_bounds__(256) void scol_gram_kernel(SColGramArgs a, GammaPackArgs gp, SSysBArgs sb) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int gram_blocks = (a.n + 3) / 4, pack_blocks = (gp.n + 7) / 8;
  // (... and behind those, the blocks of b = sum_j Pv_j (x) G_j: a third independent piece of the S system's build in this launch)
  if ((int)blockIdx.x >= gram_blocks + pack_blocks) { ssys_b_body(sb, (int)blockIdx.x - gram_blocks - pack_blocks); return; }
  if ((int)blockIdx.x >= gram_blocks) { gamma_pack_body(gp, (int)blockIdx.x - gram_blocks); return; }
  const int lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31;
  const int u = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (u >= a.n) return;
  const uint32_t s0 = __builtin_amdgcn_readfirstlane(a.slot_ptr[u]), s1 = __builtin_amdgcn_readfirstlane(a.slot_ptr[u + 1]);   // 64-wide slots, padded with the zero row
  const __amdgpu_buffer_rsrc_t rsI = panel_rsrc(reinterpret_cast<const float*>(a.idx), 0x7fffffffu);
  const __amdgpu_buffer_rsrc_t rsF = panel_rsrc(a.F, 0x7fffffffu), rsV = panel_rsrc(VB ? a.varF : a.F, 0x7fffffffu);
  const int hoff = half * 16, c4 = c * 4;
  f32x16 acc, acc2;                                                 // two accumulators: an MFMA never waits for the one before it
#pragma unroll
  for (int t = 0; t < 16; ++t) { acc[t] = 0.f; acc2[t] = 0.f; }
  float dv = 0.f;                                                   // sum_miss varF_i[c] (diagonal, VB)
  struct Idx { u32x4 q[2]; };
  struct Rows { float f[8], v[8]; };
  auto load_idx = [&](uint32_t e, Idx& ix) {                        // slots e .. e + 15: half h takes e + 8 q + 4 h .. + 3
    const int so = (int)((e < s1 ? e : s0) * 4u);                   // past the column's end: any valid slots (never multiplied)
    ix.q[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsI, hoff, so, 0));
    ix.q[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsI, hoff + 32, so, 0));
  };
  auto load_rows = [&](const Idx& ix, Rows& r) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int off = (int)(ix.q[t >> 2][t & 3] * 128u) + c4;
      r.f[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsF, off, 0, 0));
      if (VB) r.v[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsV, off, 0, 0));
    }
  };
  auto mfmas = [&](const Rows& r) {
    if constexpr (BF != 0) {
      g_u32x4 hi, mid, lo;
      g_split3(r.f, hi, mid, lo);
      const g_bf16x8 h8 = __builtin_bit_cast(g_bf16x8, hi), m8 = __builtin_bit_cast(g_bf16x8, mid), l8 = __builtin_bit_cast(g_bf16x8, lo);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l8, h8, acc, 0, 0, 0);          // small terms first, the two accumulators in turn
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h8, l8, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m8, m8, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m8, h8, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h8, m8, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h8, h8, acc2, 0, 0, 0);
      if (VB) {
#pragma unroll
        for (int t = 0; t < 8; ++t) dv += r.v[t];
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; t += 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(r.f[t], r.f[t], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(r.f[t + 1], r.f[t + 1], acc2, 0, 0, 0);
        if (VB) dv += r.v[t] + r.v[t + 1];
      }
    }
  };
  // a column with nothing missing has no slots at all (s0 == s1, possibly the very end of idx): nothing may be
  // prefetched for it -- the words behind its range are another column's, or not there at all
  if (s0 < s1) {
    Idx ia, ib; Rows ra, rb;
    load_idx(s0, ia); load_idx(s0 + 16, ib);
    load_rows(ia, ra);
    for (uint32_t e0 = s0; e0 < s1; e0 += 32) {                     // two batches per trip (the slot count is a multiple of 64): register sets alternate without copies
      load_rows(ib, rb); load_idx(e0 + 32, ia);
      mfmas(ra);
      load_rows(ia, ra); load_idx(e0 + 48, ib);
      mfmas(rb);
    }
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] += acc2[t];
  dv += __shfl_xor(dv, 32, 64);
  if (VB && a.var_obs_out && half == 0)      // sum_{i in Omega_j} varF_ik = total - missing: what the G sweep's covariance term reads (masked_colsum_kernel's output)
    a.var_obs_out[(size_t)u * 32 + c] = c < a.K ? (float)(a.cf_diag_extra[c] - a.Cf64[(size_t)c * 32 + c]) - dv : 0.f;
  float* w = a.Wc + (size_t)u * tri_padded(a.K);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int row = (t & 3) + 8 * (t >> 2) + 4 * half;             // C/D layout of the 32 x 32 tile: column on the lane
    float cf = (float)a.Cf64[(size_t)row * 32 + c];
    float m = acc[t];
    if (row == c) {
      // VB: C~f_kk = sum_i (E[F_ik]^2 + varF_ik) = the column sum of the second moments; the missing rows' variances gathered here
      // (dv), or -- round 6 -- taken from the masked variance sums the G step has formed for this very q(F): total - observed
      if (a.cf_diag_extra) cf = (float)a.cf_diag_extra[c];
      m += a.var_obs ? (float)(a.cf_diag_extra[c] - a.Cf64[(size_t)c * 32 + c]) - a.var_obs[(size_t)u * 32 + c] : dv;
    }
    if (row <= c && c < a.K) w[tri_pos(tri_index(row, c, a.K))] = cf - m;    // the upper triangle, packed: what the S-system GEMM reads
  }
}
// gp: the packing of G's second moments rides along (gp.n == 0: none); sb (may be null): and the blocks of b
void launch_scol_gram(const SColGramArgs& a, const GammaPackArgs& gp, hipStream_t st, const SSysBArgs* sb) {
  if (a.n <= 0) return;
  SSysBArgs b0 = {};
  const int blocks = (a.n + 3) / 4 + (gp.n + 7) / 8 + (sb ? ssys_b_blocks(sb->n) : 0);
  // BNMTF_SCOL_GRAM=f32: the f32 matrix-core form (A/B switch)
  static const bool f32 = [] { const char* e = getenv("BNMTF_SCOL_GRAM"); return e && !strcmp(e, "f32"); }();
  if (f32) {
    if (a.varF) hipLaunchKernelGGL((scol_gram_kernel<1, 0>), dim3(blocks), dim3(256), 0, st, a, gp, sb ? *sb : b0);
    else        hipLaunchKernelGGL((scol_gram_kernel<0, 0>), dim3(blocks), dim3(256), 0, st, a, gp, sb ? *sb : b0);
  } else {
    if (a.varF) hipLaunchKernelGGL((scol_gram_kernel<1, 1>), dim3(blocks), dim3(256), 0, st, a, gp, sb ? *sb : b0);
    else        hipLaunchKernelGGL((scol_gram_kernel<0, 1>), dim3(blocks), dim3(256), 0, st, a, gp, sb ? *sb : b0);
  }
}

// Gc[j][r(l, l')] = G_jl G_jl' (l <= l'), the packed second-moment matrix of column j's row of G.  Block = 8 columns (512 blocks at 4096 columns: two per CU hide each other's load -> store latency).
__device__ __forceinline__ void gamma_pack_body(const GammaPackArgs& a, int block) {
  constexpr int NC = 8;
  __shared__ float g[NC][32], v[NC][32];
  const int j0 = block * NC, PL = tri_count(a.L), PLp = tri_padded(a.L);
  for (int e = threadIdx.x; e < NC * 32; e += 256) {
    const int j = j0 + (e >> 5);
    g[e >> 5][e & 31] = j < a.n ? a.G[(size_t)(a.n0 + j) * 32 + (e & 31)] : 0.f;
    v[e >> 5][e & 31] = (j < a.n && a.varG) ? a.varG[(size_t)(a.n0 + j) * 32 + (e & 31)] : 0.f;
  }
  __syncthreads();
  for (int pos = threadIdx.x; pos < PLp; pos += 256) {             // by position in the row: contiguous stores
    const int r = tri_unpos(pos);
    if (r >= PL) continue;                                          // pads stay zero
    int l, lp; tri_unindex(r, a.L, &l, &lp);
#pragma unroll
    for (int t = 0; t < NC; ++t)
      if (j0 + t < a.n) a.Gc[(size_t)(j0 + t) * PLp + pos] = fmaf(g[t][l], g[t][lp], l == lp ? v[t][l] : 0.f);
  }
}
__global__ __launch_bounds__(256) void gamma_pack_kernel(GammaPackArgs a) { gamma_pack_body(a, (int)blockIdx.x); }
void launch_gamma_pack(const GammaPackArgs& a, hipStream_t st) {
  if (a.n > 0) hipLaunchKernelGGL(gamma_pack_kernel, dim3((a.n + 7) / 8), dim3(256), 0, st, a);
}

// A on the packed pairs: slab[s][p][r] = sum_{j in range s} Wc[j][p] Gc[j][r], p = (k <= k'), r = (l <= l') -- with both
// operands materialised the S system is a plain GEMM (M = N = K(K+1)/2 padded, reduction over the columns j), a quarter
// of the K L x K L products by the two symmetries.  fp32 MFMA 32x32x2, a 2 x 2 block of tiles per wave: four loads for
// four MFMAs, and nothing else in the loop -- on this chip a vector instruction between two MFMAs does not hide behind
// them (tools/micro/mfma_rate.hip: +4 cycles each on the 64 of the MFMA), and the texture path takes ~9 cycles per
// wave-level load whatever it fetches.  (The version that scaled G by W~_j[k][k'] on the fly ran at 2x its MFMA time.)
// Base addresses advance in scalar registers; a range past its end reads the zero rows behind the arrays.
__global__ __launch_bounds__(256) void ssys_gemm_kernel(SSysGemmArgs a) {
  const int lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int PKp = tri_padded(a.K), PLp = tri_padded(a.L), TR = PLp / 64;
  const int wt = blockIdx.x * 4 + wave, sp = blockIdx.y;
  if (wt >= (PKp / 64) * TR) return;
  const int tp = wt / TR, tr = wt % TR;
  const int per = ((a.n + a.nsplit - 1) / a.nsplit + 1) & ~1;      // columns per range (even: two per MFMA step)
  const int jbeg = sp * per, jend = min(a.n, jbeg + per);
  // operands through buffer descriptors: the lane's byte offset in a VGPR, the column's in an SGPR, the second tile in the
  // instruction's immediate -- no vector address arithmetic at all
    const int offA = 4 * (half * PKp + tp * 64 + 2 * c), offB = 4 * (half * PLp + tr * 64 + 2 * c);     // (tile 0, tile 1) of the lane's element: tri_pos
  const __amdgpu_buffer_rsrc_t rsA = panel_rsrc(a.Wc, (size_t)(a.n + 2) * PKp * 4), rsB = panel_rsrc(a.Gc, (size_t)(a.n + 2) * PLp * 4);
  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[x][y][t] = 0.f;
  constexpr int NS = 4;                                             // steps of two columns per batch; the next batch's operands load while this one's MFMAs run
  struct Batch { float a0[NS], a1[NS], b0[NS], b1[NS]; };
  auto fetch = [&](int j0, Batch& b) {
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      const int j = j0 + 2 * t < jend ? j0 + 2 * t : a.n;          // wave-uniform; rows n, n + 1 are zero
      const int sa = j * PKp * 4, sb = j * PLp * 4;
      const f32x2 av = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsA, offA, sa, 0));
      const f32x2 bv = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsB, offB, sb, 0));
      b.a0[t] = av.x; b.a1[t] = av.y; b.b0[t] = bv.x; b.b1[t] = bv.y;
    }
  };
  auto mfmas = [&](const Batch& b) {
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.a0[t], b.b0[t], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.a0[t], b.b1[t], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.a1[t], b.b0[t], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.a1[t], b.b1[t], acc[1][1], 0, 0, 0);
    }
  };
  Batch b0, b1;
  fetch(jbeg, b0);
  for (int j0 = jbeg; j0 < jend; j0 += 4 * NS) {                    // two batches per trip: the register sets alternate without copies
    fetch(j0 + 2 * NS, b1);                                         // (a batch past the range multiplies zeros: no branch around MFMAs,
    mfmas(b0);                                                      //  or the accumulators get copied between register files every trip)
    fetch(j0 + 4 * NS, b0);
    mfmas(b1);
  }
  float* slab = a.slabs + (size_t)sp * PKp * PLp;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = tp * 64 + 32 * x + (t & 3) + 8 * (t >> 2) + 4 * half;
        slab[(size_t)row * PLp + tr * 64 + 32 * y + c] = acc[x][y][t];
      }
}
// Round 6: the same product on the bf16 matrix cores, fp32-exact.  Every fp32 operand is split on the fly into three bf16 terms
// (round-to-nearest residuals; kernel_gemm.hip's split, restated here: the two kernels share no header) and a product is six
// v_mfma_f32_32x32x16_bf16 (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid; what is dropped is below 2^-24 of the product and has
// no systematic sign): 6 x 32 cycles per 16 columns and tile against 8 x 64 of the f32 form.  A step is 16 columns j: lane (c, g)
// takes rows j0 + 8 g + 0..7 -- eight 8-byte loads per operand, the same count as before (a load still gives the lane its element
// of BOTH interleaved tiles) -- splits its four fragments (a0, a1, b0, b1) and issues the 24 products; the next step's rows are
// on their way in a second register set meanwhile.  Ranges are cut at multiples of 16 columns; past a range's end the zero rows
// behind the arrays are read.  Same tiles, same slab layout, same epilogue as ssys_gemm_kernel.
__host__ __device__ inline int ssys_gemm_range(int n, int nsplit) { return ((n + nsplit - 1) / nsplit + 15) & ~15; }     // columns per range: a multiple of a step
__global__ __launch_bounds__(256) void ssys_gemm_bf16_kernel(SSysGemmArgs a) {
  const int lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int PKp = tri_padded(a.K), PLp = tri_padded(a.L), TR = PLp / 64;
  const int wt = blockIdx.x * 4 + wave, sp = blockIdx.y;
  if (wt >= (PKp / 64) * TR) return;
  const int tp = wt / TR, tr = wt % TR;
  const int per = ssys_gemm_range(a.n, a.nsplit);
  const int jbeg = sp * per, jend = min(a.n, jbeg + per);
  const int offA = 4 * (tp * 64 + 2 * c), offB = 4 * (tr * 64 + 2 * c);
  const __amdgpu_buffer_rsrc_t rsA = panel_rsrc(a.Wc, (size_t)(a.n + 2) * PKp * 4), rsB = panel_rsrc(a.Gc, (size_t)(a.n + 2) * PLp * 4);
  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[x][y][t] = 0.f;
  struct Raw { float a0[8], a1[8], b0[8], b1[8]; };
  auto fetch = [&](int j0, Raw& r) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int jj = j0 + 8 * half + t;
      const int j = jj < jend ? jj : a.n;                            // (rows n, n + 1 are zero)
      const f32x2 av = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsA, offA + j * PKp * 4, 0, 0));
      const f32x2 bv = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsB, offB + j * PLp * 4, 0, 0));
      r.a0[t] = av.x; r.a1[t] = av.y; r.b0[t] = bv.x; r.b1[t] = bv.y;
    }
  };
  auto products = [&](const Raw& r) {
    g_u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
    g_split3(r.a0, ah[0], am[0], al[0]); g_split3(r.a1, ah[1], am[1], al[1]);
    g_split3(r.b0, bh[0], bm[0], bl[0]); g_split3(r.b1, bh[1], bm[1], bl[1]);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        f32x16 d = acc[x][y];
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8, al[x]), __builtin_bit_cast(g_bf16x8, bh[y]), d, 0, 0, 0);     // small terms first
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8, ah[x]), __builtin_bit_cast(g_bf16x8, bl[y]), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8, am[x]), __builtin_bit_cast(g_bf16x8, bm[y]), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8, am[x]), __builtin_bit_cast(g_bf16x8, bh[y]), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8, ah[x]), __builtin_bit_cast(g_bf16x8, bm[y]), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8, ah[x]), __builtin_bit_cast(g_bf16x8, bh[y]), d, 0, 0, 0);
        acc[x][y] = d;
      }
  };
  Raw r0, r1;
  fetch(jbeg, r0);
  for (int j0 = jbeg; j0 < jend; j0 += 32) {                        // two steps per trip: the register sets alternate without copies
    fetch(j0 + 16, r1);                                             // (a step past the range multiplies zeros: no branch around the products)
    __builtin_amdgcn_sched_barrier(0);
    products(r0);
    __builtin_amdgcn_sched_barrier(0);
    fetch(j0 + 32, r0);
    __builtin_amdgcn_sched_barrier(0);
    products(r1);
    __builtin_amdgcn_sched_barrier(0);
  }
  float* slab = a.slabs + (size_t)sp * PKp * PLp;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = tp * 64 + 32 * x + (t & 3) + 8 * (t >> 2) + 4 * half;
        slab[(size_t)row * PLp + tr * 64 + 32 * y + c] = acc[x][y][t];
      }
}
void launch_ssys_gemm(const SSysGemmArgs& a, hipStream_t st) {
  const dim3 grid((ssys_gemm_wave_tiles(a.K, a.L) + 3) / 4, a.nsplit);
  const char* e = getenv("BNMTF_SSYS_GEMM");        // A/B switch: "f32" = the f32-MFMA form of rounds 2-5
  if (e && !strcmp(e, "f32")) hipLaunchKernelGGL(ssys_gemm_kernel, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(ssys_gemm_bf16_kernel, grid, dim3(256), 0, st, a);
}

// A[(k,l)][(k',l')] = sum of the column-range slabs (in range order) at (p(k,k'), r(l,l')).  One block per packed pair p:
// its row of the slabs is summed 16 bytes per thread, the two halves of the ranges by two threads (combined in a fixed
// order), goes through LDS and comes out as the full L x L blocks (k,k') and (k',k), rows contiguous.
__global__ __launch_bounds__(320) void ssys_reduce_kernel(const float* slabs, int nsplit, int K, int L, float* A) {
  __shared__ __align__(16) float v[2][640];
  int k, kp; tri_unindex(blockIdx.x, K, &k, &kp);
  const int PLp = tri_padded(L), n2 = K * L, nq = PLp / 4;         // nq <= 144 float4 per row
  const size_t slab = (size_t)tri_padded(K) * PLp;
  const int q = threadIdx.x % 160, hs = threadIdx.x / 160;         // 2 x 160 threads
  if (q < nq) {
    const int t0 = hs ? (nsplit + 1) / 2 : 0, t1 = hs ? nsplit : (nsplit + 1) / 2;
    const float4* sp = reinterpret_cast<const float4*>(slabs + (size_t)blockIdx.x * PLp) + q;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = t0; t < t1; ++t) { const float4 x = sp[(size_t)t * (slab / 4)]; s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w; }
    *reinterpret_cast<float4*>(&v[hs][4 * q]) = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < L * L; e += 320) {
    const int l = e / L, lp = e % L, pos = tri_index(min(l, lp), max(l, lp), L);      // (the slabs are indexed by the packed pair itself; only the GEMM's operands are tile-interleaved)
    const float x = v[0][pos] + v[1][pos];
    A[(size_t)(k * L + l) * n2 + kp * L + lp] = x;
    if (k != kp) A[(size_t)(kp * L + l) * n2 + k * L + lp] = x;
  }
}
void launch_ssys_reduce(const float* slabs, int nsplit, int K, int L, float* A, hipStream_t st) {
  hipLaunchKernelGGL(ssys_reduce_kernel, dim3(tri_count(K)), dim3(320), 0, st, slabs, nsplit, K, L, A);
}

// b[k][l] = sum_j Pv_jk G_jl over the local columns (Pv = the contraction's partial slabs, summed in slab order).
// Block = 64 columns: Pv and G tiles through LDS, thread (k, l) sums its 64 products; the per-block partials are summed
// in block order by ssys_reduce_kernel.
// (a body for blocks of 256 threads -- it rides in scol_gram_kernel's launch --: thread t takes the entries k = t / 32 + 8 i, l = t % 32)
__device__ __forceinline__ void ssys_b_body(const SSysBArgs& a, int block) {
  __shared__ float pv[64][33], g[64][33];
  const int j0 = block * 64;
  for (int e = threadIdx.x; e < 64 * 32; e += 256) {
    const int jj = e >> 5, cc = e & 31, j = j0 + jj;
    float p = 0.f, gg = 0.f;
    if (j < a.n) {
      gg = a.G[(size_t)(a.n0 + j) * 32 + cc];
      for (int t0 = 0; t0 < a.split; t0 += 8) {                     // eight loads in flight, added in slab order
        float w[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) w[t] = t0 + t < a.split ? a.slabs[((size_t)(t0 + t) * a.n_pad + j) * 32 + cc] : 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) p += w[t];
      }
    }
    pv[jj][cc] = p; g[jj][cc] = gg;
  }
  __syncthreads();
  const int k0 = threadIdx.x >> 5, l = threadIdx.x & 31;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int jj = 0; jj < 64; ++jj) {
    const float gv = g[jj][l];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = fmaf(pv[jj][k0 + 8 * i], gv, s[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (k0 + 8 * i < a.K && l < a.L) a.b[(size_t)block * a.K * a.L + (k0 + 8 * i) * a.L + l] = s[i];
}
__global__ __launch_bounds__(256) void ssys_b_kernel(SSysBArgs a) { ssys_b_body(a, (int)blockIdx.x); }
void launch_ssys_b(const SSysBArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(ssys_b_kernel, dim3(ssys_b_blocks(a.n)), dim3(256), 0, st, a);
}

// out = sum of the per-block partial vectors, in block order
__global__ void ssys_sum_parts_kernel(const float* slabs, int nsplit, size_t n, float* A) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int t0 = 0; t0 < nsplit; t0 += 8) {                         // eight loads in flight, added in part order
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = t0 + j < nsplit ? slabs[(size_t)(t0 + j) * n + e] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  A[e] = s;
}
void launch_ssys_sum_parts(const float* slabs, int nsplit, size_t n, float* A, hipStream_t st) {
  hipLaunchKernelGGL(ssys_sum_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slabs, nsplit, n, A);
}

constexpr int kSsysWarmBlocks = 64;   // blocks (on XCD 0) that read the fold's part of A ahead of the chain kernel
// Ti_k = (I + N_k)^-1 for the chain's rows (ssys_chain_kernel, below): N_k[l][l''] = B[l''][l] / B[l][l] for l'' < l, B the
// diagonal block k of A.  Unit lower triangular: column j of the inverse by forward substitution in fp64, one lane per column,
// the row of N a broadcast LDS read; ~500 FMAs of one wave, beside the residual's dot products on other CUs.  A dead entry
// (tau_p = tau A_aa <= 0 in fp32: what the chain tests) gets the unit row: it is drawn as 0 whatever stands before it.
// Out: [32][32] per k, zeros beyond L (unit diagonal inside).
__device__ __forceinline__ void ssys_tinv_body(const float* A, int K, int L, int k, float tau, float* out) {
  __shared__ double Nl[32][32];
  const int n2 = K * L, tid = threadIdx.x;
  for (int e = tid; e < 1024; e += 256) {
    const int l = e >> 5, m = e & 31;
    double v = 0.0;
    if (l < L && m < l) {
      const float dg = A[(size_t)(k * L + l) * n2 + k * L + l];
      if (tau * dg > 0.0f) v = (double)A[(size_t)(k * L + m) * n2 + k * L + l] / (double)dg;
    }
    Nl[l][m] = v;
  }
  __syncthreads();
  if (tid >= 32) return;
  const int j = tid;
  double X[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    double s = i == j ? 1.0 : 0.0;
#pragma unroll
    for (int m = 0; m < i; ++m) s = fma(-Nl[i][m], X[m], s);
    X[i] = (i < L && j < L) ? s : 0.0;
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) out[(size_t)k * 1024 + i * 32 + j] = (float)X[i];
}

// r = b - A S (fp64 dots: b and A S nearly cancel at convergence), one wave per row.  Lanes 0-3 of the wave also make the first
// four sampler candidates of the row's entry for the coming chain (random words only: they depend on (entry, iteration, key),
// not on A) -- a Philox call each, hidden behind the dot product instead of standing at the head of the one-block chain kernel.
// (bparts != nullptr: b is still in its nparts per-block parts -- summed here, in part order, as ssys_sum_parts_kernel would, and
// written to b as well.)  K blocks behind the rows' make the chain's Ti (tinv != nullptr).
__global__ __launch_bounds__(256) void ssys_residual_kernel(const float* A, float* b, const float* bparts, int nparts, const float* S, int n2, float* r,
                                                            float4* cands, uint32_t it, uint32_t key0, uint32_t key1, float* tinv, int K, int L, const float* tau,
                                                            float4* own8, float4* recT, float* Tn) {
#pragma clang fp contract(off)
  // the grid: K blocks for Ti (tinv != nullptr; FIRST: one wave of ~750 dependent fp64 instructions each, the longest blocks of
  // the launch), the rows' blocks, the warm-up blocks
  const int rblocks = (n2 + 3) / 4, kt = tinv ? K : 0;
  if ((int)blockIdx.x < kt) { ssys_tinv_body(A, K, L, (int)blockIdx.x, *tau, tinv); return; }
  if ((int)blockIdx.x >= rblocks + kt) {
    // Warm-up for the chain kernel that follows: it is ONE block, which the dispatcher puts on XCD 0 (block i of a grid goes to
    // XCD i mod 8: tools/micro/xcc.hip), and its fold streams the upper block triangle of A (2 MB) through that one CU -- out
    // of the Infinity Cache at ~24 B/clk, out of its own XCD's L2 several times faster.  So the blocks of this range that sit on
    // XCD 0 read that triangle once (the rows dealt round kWarm blocks); the other seven of every eight leave at once.
    const int ws = (rblocks + kt + 7) & ~7, wb = (int)blockIdx.x - ws;
    if (wb < 0 || (wb & 7) != 0) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a0 = wb >> 3; a0 < n2; a0 += kSsysWarmBlocks) {
      const int c0 = (a0 / L + 2) * L;                              // the columns of the rows of S behind the next one
      const float4* rowp = reinterpret_cast<const float4*>(A + (size_t)a0 * n2);
      for (int t = c0 / 4 + (int)threadIdx.x; t < n2 / 4; t += 256) { const float4 v = rowp[t]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e38f) r[0] = acc.x;   // (never: keeps the loads)
    return;
  }
  const int lane = threadIdx.x & 63, row = ((int)blockIdx.x - kt) * 4 + (threadIdx.x >> 6);
  if (row >= n2) return;
  if (cands && lane < 4) {
    const U4 rr = philox4x32_10(0u, (uint32_t)row, it, kStreamS + 16u * (uint32_t)lane, key0, key1);
    const TnCand cd = tn_cand_pre(rr.x, rr.y);
#ifdef BNMTF_EXPERIMENTS
    cands[row * 4 + lane] = make_float4(cd.nl, cd.z, u23(rr.y), 0.f);   // (what the entry-by-entry chain reads)
#endif
    if (own8) {                                                     // the chain's records (ssys_chain_kernel): everything of a draw that needs tau_p and the random words only
      const TnPre pre = tn_fast_pre(*tau * A[(size_t)row * n2 + row]);
      const float4 rc4 = make_float4(cd.z * pre.irt, 2.0f * cd.nl - 2.0f, 2.0f * cd.nl * pre.irt, cd.sw);
      recT[row * 4 + lane] = rc4;
      if (lane == 0) {
        own8[2 * row] = pre.live ? make_float4(rc4.x, pre.rcp, -kTnA0 * pre.irt, -pre.tpirt) : make_float4(0.f, 0.f, -__builtin_inff(), 0.f);
        own8[2 * row + 1] = make_float4(rc4.y, rc4.z, rc4.w, 0.f);
      }
    }
  }
  // (so: the row's own old values up to itself put back -- sum_{l <= lp} S_(k,l) A[(k,l)][(k,lp)], row = (k, lp): the part of the
  // same products that lies in the row's diagonal block at or before the diagonal; A is symmetric.  The chain adds it to r.)
  const int blk0 = L > 0 ? row / L * L : 0;
  double s = 0.0, so = 0.0;
  if ((n2 & 3) == 0) {                                              // 16 bytes per lane and load, all of a row's loads in flight at once (n2 <= 1024: four)
    float4 av[4], sv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = 4 * lane + 256 * i;
      av[i] = t < n2 ? *reinterpret_cast<const float4*>(A + (size_t)row * n2 + t) : make_float4(0.f, 0.f, 0.f, 0.f);
      sv[i] = t < n2 ? *reinterpret_cast<const float4*>(S + t) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = 4 * lane + 256 * i;
      const float ae[4] = {av[i].x, av[i].y, av[i].z, av[i].w}, se[4] = {sv[i].x, sv[i].y, sv[i].z, sv[i].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double p = (double)ae[c] * (double)se[c];
        s += p;
        if (t + c >= blk0 && t + c <= row) so += p;
      }
    }
  } else {
    for (int t = lane; t < n2; t += 64) {
      const double p = (double)A[(size_t)row * n2 + t] * (double)S[t];
      s += p;
      if (t >= blk0 && t <= row) so += p;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { s += __shfl_xor(s, m, 64); so += __shfl_xor(so, m, 64); }
  if (lane == 0 && Tn) Tn[row] = (float)so;
  if (lane == 0) {
    float bv;
    if (bparts) {
      bv = 0.f;
      for (int t0 = 0; t0 < nparts; t0 += 8) {                       // eight loads in flight, added in part order
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = t0 + j < nparts ? bparts[(size_t)(t0 + j) * n2 + row] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) bv += v[j];
      }
      b[row] = bv;
    } else bv = b[row];
    r[row] = (float)((double)bv - s);
  }
}
void launch_ssys_residual(const float* A, float* b, const float* bparts, int nparts, const float* S, int n2, float* r, hipStream_t st, float* cands, uint32_t it, uint32_t key0, uint32_t key1,
                          float* tinv, int K, int L, const float* tau, float* own8, float* recT, float* Tn) {
  hipLaunchKernelGGL(ssys_residual_kernel, dim3((tinv && (L & 3) == 0 && n2 >= 256) ? (((n2 + 3) / 4 + K + 7) & ~7) + 8 * kSsysWarmBlocks : (n2 + 3) / 4 + (tinv ? K : 0)), dim3(256), 0, st, A, b, bparts, nparts, S, n2, r, reinterpret_cast<float4*>(cands), it, key0, key1, tinv, K, L, tau,
                     reinterpret_cast<float4*>(own8), reinterpret_cast<float4*>(recT), Tn);
}

#ifdef BNMTF_EXPERIMENTS   // the chain as rounds 2-4 walked it, entry by entry: for same-box comparisons (make EXPERIMENTS=1, BNMTF_SCHAIN=seq)
// The K.L sequential conditionals, row-major (k, l) (bnmtf_gibbs_optimised.py:157-160), one block of 8 waves.
// Measured with -DCHAIN_CLOCK (tools/variant.sh): a lone wave issues an instruction every ~9 cycles, so the chain's cost is
// its instruction count: ~195 cycles a step in the normal regime, ~280 more in the translated-exponential one.
//   * the residual r = b - A S lives in LDS.  Row k of S is walked by wave 0, lane l owning entry (k, l).  Its numerator is
//       -lambda + tau (r_l + sum_{l'' <= l} Sold_l'' A_l''l  -  sum_{l'' < l} Snew_l'' A_l''l):
//     the first sum (the row's own old values put back into the residual) does not depend on the chain and is made by a
//     background wave a row ahead; the second is ONE FMA per step against the row's diagonal block, staged in LDS
//     already multiplied by -tau.
//   * everything else that does not depend on the chain is off it too: the first four candidates of EVERY entry are made
//     by the whole block before the first row, already scaled by the entry's sigma (tau_p = tau A_aa is known up front),
//     so that in the normal regime a step is: readlane (numerator) -> one FMA per candidate lane (x = numer / tau_p +
//     z sigma) -> class test (accepted iff x is a non-negative finite number: the event z >= -mu sqrt(tau_p) of
//     oracle/rng.py, stated on x) -> ballot -> readlane -> the FMA above; its LDS operands are read two steps ahead.
//     The regime test is a compare of the numerator with a per-entry threshold (-A0 sqrt(tau_p)); the translated-
//     exponential regime, dead entries and a whole batch of rejections leave through ONE wave-uniform branch.
//   * the other waves run a row ahead of the chain and never wait for memory inside a row: the loads they issue while wave 0
//     walks row k (blocks of A for row k+2, the rows of A that fold row k's deltas into the residual of rows >= k+2)
//     are consumed during row k+1, when the deltas exist.  A is symmetric: a column is read as a coalesced row, 16 bytes
//     per lane.  The one row that cannot wait for the fold -- the next one -- gets the last deltas from wave 0 itself,
//     out of registers, against the staged off-diagonal block.
//   cond >= 0: only evaluate entry `cond` (numer, tau_p) and change nothing -- the tauS / muS hook.
template <int UPDATE>      // 0: draws, 1: mode updates (ICM / the deterministic harness)
__global__ __launch_bounds__(512) void ssys_chain_seq_kernel(SSysChainArgs a) {
#pragma clang fp contract(off)
  constexpr int NH = 4;                                            // hoisted candidates per entry
  // Od(k, k+1) is staged during row k-1 and read during row k+1: three buffers.  Spare rows / entries behind Om and the
  // candidate records: the two-steps-ahead reads of the chain run past the end.
  __shared__ __align__(16) float r[1024];
  __shared__ __align__(16) float delta[2][32];
  __shared__ float snl[32 + 4];
  __shared__ float Sl[1024], laml[1024], Om[2][35 * 33], Od[3][32 * 33], Tn[2][32];
  __shared__ float4 recA[(1024 + 3) * NH];                         // {z sigma, 1 / tau_p, regime threshold, sigma}
  __shared__ float2 recB[(1024 + 3) * NH];                         // {-log u1, u2}: the translated-exponential regime
  const int K = a.K, L = a.L, n2 = K * L, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // in an SGPR: the role branches below are jumps, not exec masks
  const float tau = *a.tau;
#ifdef CHAIN_CLOCK
  const unsigned long long c_k0 = __builtin_amdgcn_s_memtime();
#endif
  if (a.cond >= 0) {                                               // tauS(k,l) / muS(k,l) hook: the residual already holds everything
    if (tid == 0) {
      const float aaa = a.A[(size_t)a.cond * n2 + a.cond];
      a.numer_out[0] = (double)fmaf(tau, a.r0[a.cond] + a.S[a.cond] * aaa, -a.lambdaS[a.cond]);
      a.tau_out[0] = (double)(tau * aaa);
    }
    return;
  }
  // ---- background register pipeline: issued in one row, consumed in the next.  Eight waves (256 VGPRs each: the
  // pipeline is held in registers); wave 4 shares wave 0's SIMD (waves of a workgroup go round the four SIMDs) and
  // stays idle -- whatever it issued would take issue slots from the chain.  That leaves 6 waves, 384 threads.
  constexpr int NT = 512, NB = 384, QB = 3, QF = 5, kOwnWave = 7;   // kOwnWave: the background wave that also makes the own-row term (the last one: its fold items run out first)                // block elements / fold items per thread: L.L <= QB NB, 2 (n2 - 2 L) <= QF NB
  const bool bg = (wave & 3) != 0;
  const int bt = (wave - 1 - (wave >> 2)) * 64 + lane;
  float sm[QB], sd[QB], own[16];            // blocks of the row after next; its own-row operands (wave 1: 16 per half)
  float4 fold[QF][4];                                                // A[(k, h + 8 j)][4 columns]: the rows that fold row k's deltas, QF items a thread
  const bool vec = (L & 3) == 0;
  const __amdgpu_buffer_rsrc_t rsAm = panel_rsrc(a.A, (size_t)n2 * n2 * 4);
  auto issue_blocks = [&](int kk) {                                // diagonal block of row kk and block (kk, kk+1) -> registers
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int t = bt + q * NB;
      if (t < L * L) {
        const int l1 = t / L, l2 = t % L;
        sm[q] = a.A[(size_t)(kk * L + l1) * n2 + kk * L + l2];
        if (kk + 1 < K) sd[q] = a.A[(size_t)(kk * L + l1) * n2 + (kk + 1) * L + l2];
      }
    }
    if (wave == kOwnWave) {                                         // rows kk L + lb + 0..15 of the diagonal block, column lp: lane part in a VGPR, row part in SGPRs,
      const int voff = 4 * ((lane & 32 ? 16 : 0) * n2 + (lane & 31));   // nothing predicated (past the matrix the descriptor returns 0; l > lp is masked when it is used)
#pragma unroll
      for (int l = 0; l < 16; ++l) own[l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsAm, voff, 4 * ((kk * L + l) * n2 + kk * L), 0));
    }
  };
  auto store_blocks = [&](int kk) {
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int t = bt + q * NB;
      if (t < L * L) {
        const int l1 = t / L, l2 = t % L;
        Om[kk & 1][l1 * 33 + l2] = -tau * sm[q];
        if (kk + 1 < K) Od[kk % 3][l1 * 33 + l2] = sd[q];
      }
    }
    if (wave == kOwnWave) {                                         // sum_{l <= lp} S_(kk,l) A[(kk,l)][(kk,lp)], row kk not walked yet
      const int lb = lane & 32 ? 16 : 0, lp = lane & 31;
      float s = 0.f;
#pragma unroll
      for (int l = 0; l < 16; ++l) s = fmaf(lb + l <= lp ? Sl[kk * L + lb + l] : 0.f, own[l], s);   // (lp < L for every lane that is kept: lb + l stays inside the row)
      s += __shfl_xor(s, 32, 64);
      if (lane < 32) Tn[kk & 1][lane] = lane < L ? s : 0.f;
    }
  };
  auto issue_fold = [&](int kk) {                                  // rows of A for the deltas of row kk, columns of rows >= kk+2
    const int t0 = (kk + 2) * L, items = 2 * (n2 - t0);            // (n2 - t0) / 4 column groups x 8 residues of l
    if (vec) {
#pragma unroll
      for (int q = 0; q < QF; ++q) {
        const int w = bt + q * NB;
        if (w < items) {
          const int g = w >> 3, h = w & 7;
          const float* col = a.A + (size_t)(kk * L + h) * n2 + t0 + 4 * g;
#pragma unroll
          for (int j = 0; j < 4; ++j) fold[q][j] = h + 8 * j < L ? *reinterpret_cast<const float4*>(col + (size_t)(8 * j) * n2) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  };
  // ---- prologue
  for (int e = tid; e < n2; e += NT) {
    r[e] = a.r0[e]; Sl[e] = a.S[e]; laml[e] = a.lambdaS[e];
    if (UPDATE == 0) {
      const TnPre pre = tn_fast_pre(tau * a.A[(size_t)e * n2 + e]);
      const float thr = pre.live ? -kTnA0 * pre.tpirt : __builtin_inff();    // a dead entry always leaves through the slow branch
#pragma unroll
      for (int c = 0; c < NH; ++c) {
        const float4 cd = a.cands[e * NH + c];                      // {-log u1, z, u2, -}: made by ssys_residual_kernel, off this kernel's critical path
        recA[e * NH + c] = make_float4(cd.y * pre.irt, pre.rcp, thr, pre.irt);
        recB[e * NH + c] = make_float2(cd.x, cd.z);
      }
    }
  }
  if (tid < 64) { delta[0][tid & 31] = 0.f; delta[1][tid & 31] = 0.f; }
  for (int t = tid; t < 2 * 35 * 33; t += NT) (&Om[0][0])[t] = 0.f;      // lanes beyond L read (and discard) these: keep them finite
  for (int t = tid; t < 3 * 32 * 33; t += NT) (&Od[0][0])[t] = 0.f;
  __syncthreads();
  if (bg) { issue_blocks(0); store_blocks(0); if (K > 1) { issue_blocks(1); store_blocks(1); } }
  __syncthreads();
#ifdef CHAIN_CLOCK
  unsigned long long c_pro = 0, c_steps = 0, c_wait = 0, c_slow = 0, c_t0 = __builtin_amdgcn_s_memtime(), c_begin = c_t0; int n_slow = 0;
#endif
  // two loops, one per role, meeting at the same K barriers: the register allocation is the larger of the two roles, not their sum
  if (wave == 0) {
    for (int k = 0; k < K; ++k) {
      const int cur = k & 1;
      {
      const bool on = lane < L;
      const int l32 = lane & 31;                                    // lanes >= 32 mirror lanes 0-31: every LDS address below is valid, no exec juggling
      const int me = k * L + (on ? lane : 0);
      float my_eta = r[me];
      if (k > 0) {                                                  // the previous row's deltas, which the background pass has not folded in yet
        // (a lone wave issues an instruction every ~7 cycles: what counts here is the number of instructions.  Each half
        // of the wave takes 16 of the 32 rows; rows and columns beyond L hold zeros, no predicates.)
        const float* od = Od[(k - 1) % 3] + (lane & 32 ? 16 * 33 : 0) + l32;
        const float* dq = delta[cur ^ 1] + (lane & 32 ? 16 : 0);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = fmaf(dq[j], od[j * 33], acc);
        acc += __shfl_xor(acc, 32, 64);
        my_eta -= acc;
      }
      const float* om = Om[cur];
      const float my_taup = -om[l32 * 33 + l32], my_sold = Sl[me];
      float numer_v = fmaf(tau, my_eta + Tn[cur][l32], -laml[me]);  // every old value of the row put back; the new ones enter as the chain walks
      const int cbase = k * L * NH + (lane & (NH - 1));
      // -tau A[(k,l)][(k,lane)] and candidate `lane & 3` of entry l: three steps ahead, in four register sets
      const float* omp = om + l32; const float4* pa = recA + cbase; const float2* pb = recB + cbase;
      float rowq[4]; float4 raq[4]; float2 rbq[4];
#pragma unroll
      for (int q = 0; q < 3; ++q) { rowq[q] = omp[q * 33]; raq[q] = pa[q * NH]; rbq[q] = pb[q * NH]; }
      float* snp = snl;                                             // the row's new values, by entry (every lane writes the same word)
      const int cmask = lane < NH ? 0x1C0 : 0;                      // +0, +denormal, +normal
      auto step = [&](int l, int q, float row, float4 ra, float2 rb) {
        const float numer = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, numer_v), l));
        float snew;
        if (UPDATE == 0) {
          float xc = fmaf(numer, ra.y, ra.x);                       // normal regime, live entry: accepted iff x is +0, +denormal or +normal
          unsigned long long mc;
          asm("v_cmp_class_f32 %0, %1, %2" : "=s"(mc) : "v"(xc), "v"(cmask));   // cmask: the class bits in the candidate lanes, 0 (never) in the others
          unsigned long long m = mc & ~__builtin_amdgcn_ballot_w64(numer <= ra.z);
          if (__builtin_expect(m == 0ull, 0)) {
#ifdef CHAIN_CLOCK
            ++n_slow; const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
            // translated-exponential regime, a dead entry, or the whole batch rejected.  The entry's constants are in the
            // record: sigma, 1 / tau_p, sqrt(tau_p) = -thr / A0 (thr = +inf marks a dead entry)
            const float irt = ra.w, rcp = ra.y, tpirt = -4.0f * ra.z;
            static_assert(kTnA0 == 0.25f, "tpirt above is -thr / A0");
            TnFast tp; tp.mu = numer * rcp; tp.irt = irt; tp.a = -tp.mu * tpirt; tp.live = true; tp.tail = tp.a >= kTnA0;
            tp.d = 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(tp.a, tp.a, 4.0f)) + tp.a); tp.ilam = __builtin_amdgcn_rcpf(tp.a + tp.d);
            if (__builtin_amdgcn_ballot_w64(ra.z < __builtin_inff() && isfinite(tp.a)) != 0ull) {
              if (__builtin_amdgcn_ballot_w64(tp.tail) != 0ull) {   // the hoisted candidates, in this regime
                const float e = rb.x * tp.ilam, t = e - tp.d;
                xc = e * irt;
                unsigned long long mt;
                asm("v_cmp_class_f32 %0, %1, %2" : "=s"(mt) : "v"(xc), "s"(0x1C0));
                m = (mt & __builtin_amdgcn_ballot_w64(rb.y <= __builtin_amdgcn_exp2f(-0.72134752f * t * t))) & ((1ull << NH) - 1ull);
              }                                                     // (normal regime here: the threshold compare and tp.tail disagree in the last bit, or all four were rejected)
              for (uint32_t round = 0; m == 0ull && round < 64u; ++round) {   // candidates NH + 64 round + lane
                const U4 rr = philox4x32_10(0u, (uint32_t)(k * L + l), a.it, kStreamS + 16u * ((uint32_t)NH + round * 64u + (uint32_t)lane), a.key0, a.key1);
                const bool acc = tn_eval_fast(tp, rr.x, rr.y, &xc);
                m = __builtin_amdgcn_ballot_w64(acc && isfinite(xc) && xc >= 0.0f);
              }
            }
            if (m == 0ull) { xc = 0.f; m = 1ull; }                  // dead entry, or 4100 rejections
#ifdef CHAIN_CLOCK
            c_slow += __builtin_amdgcn_s_memtime() - ts0;
#endif
          }
          snew = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xc), __builtin_ctzll(m)));
        } else {
          const float tau_p = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_taup), l));
          const float mu = numer / tau_p;
          snew = fmaxf((tau_p > 0.f && mu > 0.f) ? mu : 0.f, a.min_x);
        }
        snp[q] = snew;
        numer_v = fmaf(snew, row, numer_v);                          // what lanes l' > l see when their turn comes
      };
#ifdef CHAIN_CLOCK
      { const unsigned long long tt = __builtin_amdgcn_s_memtime(); c_pro += tt - c_t0; c_t0 = tt; }
#endif
      auto quad = [&](int l, auto guarded) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (!decltype(guarded)::value || l + q < L) {
            rowq[(q + 3) & 3] = omp[(q + 3) * 33];
            if (UPDATE == 0) { raq[(q + 3) & 3] = pa[(q + 3) * NH]; rbq[(q + 3) & 3] = pb[(q + 3) * NH]; }
            step(l + q, q, rowq[q], raq[q], rbq[q]);
          }
        }
        omp += 4 * 33; pa += 4 * NH; pb += 4 * NH; snp += 4;
      };
      if ((L & 3) == 0) for (int l = 0; l < L; l += 4) quad(l, std::false_type{});
      else              for (int l = 0; l < L; l += 4) quad(l, std::true_type{});
#ifdef CHAIN_CLOCK
      { const unsigned long long tt = __builtin_amdgcn_s_memtime(); c_steps += tt - c_t0; c_t0 = tt; }
#endif
      const float my_s = snl[l32];
      const float my_delta = on ? my_s - my_sold : 0.f;
      if (on) { Sl[k * L + lane] = my_s; delta[cur][lane] = my_delta; }
      }
      __syncthreads();
#ifdef CHAIN_CLOCK
      { const unsigned long long tt = __builtin_amdgcn_s_memtime(); c_wait += tt - c_t0; c_t0 = tt; }
#endif
    }
  } else {
#ifdef CHAIN_CLOCK
    unsigned long long b_work = 0, b_wait = 0, b_t0 = __builtin_amdgcn_s_memtime(), b_cons = 0;
#endif
    for (int k = 0; k < K; ++k) {
      const int cur = k & 1;
      if (bg) {
      // ---- consume what was issued during the previous row
      if (k >= 1 && k + 1 < K) store_blocks(k + 1);
      if (k > 0) {
        const float* dp = delta[cur ^ 1];
        const int t0 = (k + 1) * L;
        if (vec) {
          const int items = 2 * (n2 - t0);
#pragma unroll
          for (int q = 0; q < QF; ++q) {
            const int w = bt + q * NB;
            if (w < items) {                                         // item = (four columns, l mod 8)
              const int g = w >> 3, h = w & 7;
              float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float d = dp[(h + 8 * j) & 31];
                s.x = fmaf(d, fold[q][j].x, s.x); s.y = fmaf(d, fold[q][j].y, s.y); s.z = fmaf(d, fold[q][j].z, s.z); s.w = fmaf(d, fold[q][j].w, s.w);
              }
              // the eight residues sit in eight adjacent lanes (the item count is a multiple of eight): a fixed-order butterfly
#pragma unroll
              for (int mk = 1; mk <= 4; mk <<= 1) {
                s.x += __shfl_xor(s.x, mk, 64); s.y += __shfl_xor(s.y, mk, 64); s.z += __shfl_xor(s.z, mk, 64); s.w += __shfl_xor(s.w, mk, 64);
              }
              if (h == 0) {
                float4* rp = reinterpret_cast<float4*>(&r[t0 + 4 * g]);
                float4 o = *rp;
                o.x -= s.x; o.y -= s.y; o.z -= s.z; o.w -= s.w;
                *rp = o;
              }
            }
          }
        } else {
          for (int t = t0 + bt; t < n2; t += NB) {
            const float* col = a.A + (size_t)((k - 1) * L) * n2 + t;
            float s = r[t];
            for (int l0 = 0; l0 < L; l0 += 8) {
              float av[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) av[j] = l0 + j < L ? col[(size_t)(l0 + j) * n2] : 0.f;
#pragma unroll
              for (int j = 0; j < 8; ++j) s = fmaf(-dp[(l0 + j) & 31], av[j], s);
            }
            r[t] = s;
          }
        }
      }
#ifdef CHAIN_CLOCK
      { const unsigned long long tt = __builtin_amdgcn_s_memtime(); b_cons += tt - b_t0; }
#endif
      // ---- issue for the next row
      if (k + 2 < K) { issue_blocks(k + 2); issue_fold(k); }
      }
#ifdef CHAIN_CLOCK
      { const unsigned long long tt = __builtin_amdgcn_s_memtime(); b_work += tt - b_t0; b_t0 = tt; }
#endif
      // LDS traffic only: the loads just issued stay in flight across the barrier (they are consumed in the next row;
      // __syncthreads() would wait for them here, a memory latency per row in front of the chain)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef CHAIN_CLOCK
      { const unsigned long long tt = __builtin_amdgcn_s_memtime(); b_wait += tt - b_t0; b_t0 = tt; }
#endif
    }
#ifdef CHAIN_CLOCK
    if ((tid == 64 || tid == 7 * 64) && a.it == 30u) printf("bg wave %d: work %llu (consume %llu) wait %llu\n", wave, b_work, b_cons, b_wait);
#endif
  }
  for (int e = tid; e < n2; e += NT) a.S[e] = Sl[e];               // the walked rows, after the last barrier
#ifdef CHAIN_CLOCK
  if (tid == 0 && (a.it == 30u || a.it == 31u)) printf("chain it %u: prologue %llu, row-pro %llu, steps %llu, barrier %llu cycles; slow %d taking %llu\n", a.it, c_begin - c_k0, c_pro, c_steps, c_wait, n_slow, c_slow);
#endif
}

#endif

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the chain by ROWS.  The L conditionals of row k are one unit-lower-triangular system: with B the diagonal block of
// A, base_l = -lambda_l + tau (r_l + sum_{l'' <= l} Sold_l'' B_l''l) (everything that does not depend on the row's new values)
// and c_l = base_l / tau_p,l + z_l sigma_l (the entry's first candidate), the sequential steps are
//     x_l = c_l - sum_{l'' < l} N_ll'' x_l'' ,   N_ll'' = B_l''l / B_ll        <=>        (I + N) x = c        <=>        x = Ti c ,
// as long as every entry is in the normal regime and accepts its first candidate.  Ti = (I + N)^-1 depends on A only: K small
// triangular inversions, made in fp64 by blocks that ride behind ssys_residual_kernel.  So the chain wave does NOT walk the
// entries: it forms c (one FMA per lane), multiplies by Ti (16 FMAs per lane, the two halves of the wave take half of the
// sum each) and TESTS the row in one go -- lane l is fine when x_l is a non-negative finite number and mu_l = x_l - z_l sigma_l
// lies in the normal regime.  The first lane f that is not fine has a VALID conditional mean (it depends on the lanes before
// it only, and they are fine): it is drawn on its own (the other hoisted candidates; the translated-exponential regime;
// Philox rounds) and the value it ends with is put into the lanes behind it by ONE more FMA, x_l += Ti[l][f] (x_f' - x_f) --
// the row's solution for the right-hand side with c_f replaced --, then the test is repeated for the lanes behind f.
// tools/micro/lone_wave.hip: a lone wave issues a vector instruction every ~4 cycles, dependent or not (8 for rcp / sqrt; ~8
// per instruction on a compare -> scalar -> readlane hop; an LDS read comes back after ~64).  A row is ~90 instructions and
// three LDS round trips; a lane that leaves the fast form ~35 instructions, its records fetched while the lane before it is
// still being drawn.  Mode updates (ICM, the deterministic harness): the same with c_l = base_l / tau_p,l and "fine" = the
// clamp leaves x_l alone.
// The translated-exponential draw (oracle/rng.py: e = nl / lam, accepted iff u2 <= exp(-(e - d)^2 / 2), x = e / sqrt(tau_p);
// a = -mu sqrt(tau_p), d = 2 / (sqrt(a^2 + 4) + a), lam = a + d) is stated with 1 / lam = d (Robert's rate: lam d = 1):
//     rc = 1 / (sqrt(a^2 + 4) + a),  t = e - d = (2 nl - 2) rc,  x = (2 nl sigma) rc,  accepted iff |t| <= sqrt(-2 ln u2)
// -- the same events and values up to rounding, one reciprocal and no exponential on the chain; (2 nl - 2), 2 nl sigma and the
// square root are per-candidate constants made by ssys_residual_kernel beside its dot products.
// Rounding: Ti c sums the same products as the forward substitution in another order (both fp32; against the fp64 oracle they
// are equally far, tests/test_bnmtf_gibbs_gpu.py); the chain is deterministic and the same on 1 and N GPUs (Ti is made from the
// summed A).
template <int UPDATE>      // 0: draws, 1: mode updates
__global__ __launch_bounds__(512) void ssys_chain_kernel(SSysChainArgs a) {
#pragma clang fp contract(off)
  constexpr int NH = 4, TS = 36;                                   // hoisted candidates per entry; row stride of a staged Ti (16-byte rows, conflict-free b128 reads)
  __shared__ __align__(16) float r[1024];
  __shared__ __align__(16) float delta[2][32];
  __shared__ __align__(16) float cs[32];
  __shared__ __align__(16) float Sl[1024], laml[1024], Tn[1024];
  __shared__ __align__(16) float Od[3][32 * 32];                   // block (kk, kk+1) of A, by kk mod 3: stored during row kk-1, read during row kk+1
  __shared__ __align__(16) float Ti[2][32 * TS];                   // Ti of row kk, by kk mod 2: stored during row kk-1
  __shared__ float4 own8[2 * 1024];                                // per entry: {z0 sigma, 1 / tau_p, -A0 sigma, -sqrt(tau_p)}, {2 nl0 - 2, 2 nl0 sigma, sqrt(-2 ln u2_0), -}; dead: {0, 0, -inf, 0}
  __shared__ float4 recT[UPDATE == 0 ? 1024 * NH : 1];             // per candidate: {z_c sigma, 2 nl - 2, 2 nl sigma, sqrt(-2 ln u2)}: the cold path
  const int K = a.K, L = a.L, n2 = K * L, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float tau = *a.tau;
#ifdef CHAIN_CLOCK
  const unsigned long long c_k0 = __builtin_amdgcn_s_memtime();
#endif
  if (a.cond >= 0) {                                               // tauS(k,l) / muS(k,l) hook: the residual already holds everything
    if (tid == 0) {
      const float aaa = a.A[(size_t)a.cond * n2 + a.cond];
      a.numer_out[0] = (double)fmaf(tau, a.r0[a.cond] + a.S[a.cond] * aaa, -a.lambdaS[a.cond]);
      a.tau_out[0] = (double)(tau * aaa);
    }
    return;
  }
  // ---- the other waves.  Waves 1-3 and 5-7 FOLD: the rows of A that put a row's deltas into the residual of the rows behind it
  // are loaded TWO rows ahead of their use into two register sets that take turns (a row is short now: one row of slack did not
  // cover the loads' latency); every thread issues the same twenty loads per row whatever its items are (out of range: the buffer
  // descriptor returns zeros), and these waves issue no other vector-memory operation inside the loop, so the compiler's wait in
  // front of a set's first use is "all but the newest twenty".  Wave 4 -- on wave 0's SIMD, where it costs the chain ~40 issue
  // slots a row -- STAGES what the chain wave reads besides the residual: Ti of a row and the block of A towards the next row,
  // loaded two rows ahead, stored one row ahead (registers in between; its waits concern nothing else).
  constexpr int NT = 512, NB = 384, QF = 5;
  const bool bg = (wave & 3) != 0;
  const int bt = (wave - 1 - (wave >> 2)) * 64 + lane;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) void* lds_ptr;
  f32x4 fs0[QF][4], fs1[QF][4];
  const bool vec = (L & 3) == 0;
  const __amdgpu_buffer_rsrc_t rsAm = panel_rsrc(a.A, (size_t)n2 * n2 * 4);
  // (invalid items are sent past the end of the descriptor by ARITHMETIC on the offset -- a select makes this compiler split the
  // loads over divergent branches with a full wait in between)
  constexpr int kPast = 0x40000000;
  int f_off[QF], f_row[4];                                          // fold items: byte offset of (row h, column 4 g) / kPast for the rows h + 8 j >= L
#pragma unroll
  for (int q = 0; q < QF; ++q) { const int w = bt + q * NB; f_off[q] = 4 * ((w & 7) * n2 + 4 * (w >> 3)); }
#pragma unroll
  for (int j = 0; j < 4; ++j) f_row[j] = (bt & 7) + 8 * j < L ? 4 * 8 * j * n2 : kPast;
  // the stager's pieces: 16 bytes per lane and piece; Ti of a row = 4 pieces (the row's 32 x 32 floats, contiguous), the block
  // (kk, kk+1) of A = 4 pieces (a lane's four floats out of its row of A; rows / columns beyond L and the block behind the last
  // row: past the descriptor, zeros)
  f32x4 tq[4], oq[4];
  auto stage_issue = [&](int kk) {
    const int sb = kk + 1 < K ? 4 * (kk * L * n2 + (kk + 1) * L) : kPast;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = j * 64 + lane;
      tq[j] = *reinterpret_cast<const f32x4*>(a.Tinv + (size_t)kk * 1024 + 4 * idx);
      const int l1 = idx >> 3, c4 = (idx & 7) * 4;
      const int past = (((L - 1 - l1) | (L - 1 - c4)) >> 31) & kPast;
      oq[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsAm, 4 * (l1 * n2 + c4) + past, sb, 0));
    }
  };
  auto stage_store = [&](int kk) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = j * 64 + lane;
      *reinterpret_cast<f32x4*>(Ti[kk & 1] + (idx >> 3) * TS + (idx & 7) * 4) = tq[j];
      if (vec) *reinterpret_cast<f32x4*>(Od[kk % 3] + 4 * idx) = oq[j];
    }
    if (!vec)                                                       // L not a multiple of four (no aligned 16-byte pieces): the block element by element, at once
      for (int t = lane; t < 1024; t += 64) {
        const int l1 = t >> 5, l2 = t & 31;
        Od[kk % 3][t] = (l1 < L && l2 < L && kk + 1 < K) ? a.A[(size_t)(kk * L + l1) * n2 + (kk + 1) * L + l2] : 0.f;
      }
  };
  auto issue_fold = [&](int kk, f32x4 (&fs)[QF][4]) {              // A[(kk, h + 8 j)][t0 + 4 g ..+3], t0 = (kk + 2) L: item w = (g, h)
    const int t0 = (kk + 2) * L, items = 2 * (n2 - t0);
    const int sbase = 4 * (kk * L * n2 + t0);
#pragma unroll
    for (int q = 0; q < QF; ++q) {
      const int w = bt + q * NB;
      const int past = ((items - 1 - w) >> 31) & kPast;            // w >= items
#pragma unroll
      for (int j = 0; j < 4; ++j)
        fs[q][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsAm, f_off[q] + f_row[j] + past, sbase, 0));
    }
  };
  f32x2 rq0 = {0.f, 0.f}, rq1 = {0.f, 0.f};
  auto consume_fold = [&](int kk, const f32x4 (&fs)[QF][4], const float* dp) {   // r[t] -= sum_l delta_(kk,l) A[(kk,l)][t] for t >= (kk + 2) L
    const int t0 = (kk + 2) * L, items = 2 * (n2 - t0);
#pragma unroll
    for (int q = 0; q < QF; ++q) {
      const int w = bt + q * NB, g = w >> 3, h = w & 7;
      f32x4 sv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = dp[(h + 8 * j) & 31];
        sv.x = fmaf(d, fs[q][j].x, sv.x); sv.y = fmaf(d, fs[q][j].y, sv.y); sv.z = fmaf(d, fs[q][j].z, sv.z); sv.w = fmaf(d, fs[q][j].w, sv.w);
      }
      // the eight residues sit in eight adjacent lanes: xor 1, xor 2, then the other quad of the eight (the same tree as a butterfly)
      // (one instruction per step and component: through the builtin the compiler makes a DPP move and an add; s_nop 1 = the
      // wait states a DPP read needs behind the vector instruction that wrote its source -- the four components hide each other's)
      asm volatile("s_nop 1\n\t"
                   "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf"
                   : "+v"(sv.x), "+v"(sv.y), "+v"(sv.z), "+v"(sv.w));
      if (h == 0 && w < items) {
        // r[t0 + 4 g ..+3] -= sv as ONE asm statement on two register pairs of its own that live across the rows: through C++ the
        // loaded quad lands in registers of the set being consumed and the compiler waits for the loads it believes pending on them
        // -- on one of the two sets that was every load in flight
        const uint32_t ra = (uint32_t)(size_t)(__attribute__((address_space(3))) float*)(&r[t0 + 4 * g]);
        const f32x2 s01 = {sv.x, sv.y}, s23 = {sv.z, sv.w};
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_pk_add_f32 %0, %0, %3 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %1, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "ds_write_b64 %2, %0\n\tds_write_b64 %2, %1 offset:8"
                     : "+v"(rq0), "+v"(rq1) : "v"(ra), "v"(s01), "v"(s23) : "memory");
      }
    }
  };
  // ---- prologue: every array the chain reads is a straight copy of what ssys_residual_kernel left in global memory -- LDS-DMA
  // through buffer descriptors (1 KiB per wave instruction, no registers, past the end: zeros), ONE memory round trip.  The
  // background waves' first loads go out around it.
  {
    int ch = wave;                                                  // chunks of 1 KiB, dealt round the eight waves across all the arrays
    auto dma = [&](const void* src, size_t bytes, void* dst, int chunks) {
      const __amdgpu_buffer_rsrc_t rs = panel_rsrc(reinterpret_cast<const float*>(src), bytes);
      for (; ch < chunks; ch += 8)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(reinterpret_cast<char*>(dst) + (size_t)ch * 1024), 16, lane * 16, ch * 1024, 0, 0);
      ch -= chunks;
    };
    const int c4 = (n2 * 4 + 1023) / 1024;
    dma(a.r0, (size_t)n2 * 4, r, c4);
    dma(a.S, (size_t)n2 * 4, Sl, c4);
    dma(a.lambdaS, (size_t)n2 * 4, laml, c4);
    dma(a.Tn, (size_t)n2 * 4, Tn, c4);
    dma(a.own8, (size_t)n2 * 32, own8, (n2 * 32 + 1023) / 1024);
    if (UPDATE == 0) dma(a.recT, (size_t)n2 * 64, recT, (n2 * 64 + 1023) / 1024);
  }
  if (tid < 64) { delta[0][tid & 31] = 0.f; delta[1][tid & 31] = 0.f; }
  if (wave == 4) { stage_issue(0); stage_store(0); }
  // (the builtin, not an asm statement: the compiler has to SEE that the copies have landed, or it puts a full wait in front of
  // every LDS read it cannot tell apart from their destinations -- inside the row loops)
  __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0)
  if (wave == 4) stage_issue(min(1, K - 1));
  // (row 0 runs the same code as every other row -- it "consumes" a set of zeros against deltas of zero, loaded from past the
  // matrix --, so that at every first use of a set exactly twenty newer loads are in flight, on the first trip as on all others:
  // the compiler's wait count is the worst case over the ways into the loop)
  if (bg && vec) { issue_fold(K, fs1); issue_fold(0, fs0); }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef CHAIN_CLOCK
  unsigned long long c_pro = 0, c_fix = 0, c_wait = 0, c_t0 = __builtin_amdgcn_s_memtime(), c_begin = c_t0; int n_fix = 0, n_cold = 0;
#endif
  if (wave == 0) {
    const int l32 = lane & 31, hb = lane & 32 ? 16 : 0;
    const bool on = l32 < L;
    const uint32_t lmask = L >= 32 ? 0xffffffffu : ((1u << L) - 1u);
    for (int k = 0; k < K; ++k) {
      const int cur = k & 1;
      const int me = k * L + (on ? l32 : 0);
      float my_eta = r[me];
      const float4 o0 = own8[2 * me], o1 = own8[2 * me + 1];
      const float my_sold = Sl[me], my_lam = laml[me], my_tn = Tn[me];
      float trow[32];                                               // the lane's row of Ti: the product below, and the column a corrected lane spreads by
      {
        const float4* tp4 = reinterpret_cast<const float4*>(Ti[cur] + l32 * TS);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float4 v = tp4[j]; trow[4 * j] = v.x; trow[4 * j + 1] = v.y; trow[4 * j + 2] = v.z; trow[4 * j + 3] = v.w; }
      }
      if (k > 0) {                                                  // the previous row's deltas, which the background pass has not folded in yet
        const float* od = Od[(k - 1) % 3] + hb * 32 + l32;
        const float* dq = delta[cur ^ 1] + hb;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = fmaf(dq[j], od[j * 32], acc);
        my_eta -= half_swap_sum(acc);
      }
      const float base = fmaf(tau, my_eta + my_tn, -my_lam);
      float c = on ? fmaf(base, o0.y, UPDATE == 0 ? o0.x : 0.f) : 0.f;
      // (an overflowed numerator: the entry is drawn as a dead one, 0 -- and must not poison the row's product)
      const uint32_t forced = (uint32_t)__builtin_amdgcn_ballot_w64(!(fabsf(c) < __builtin_inff())) & lmask;
      c = fabsf(c) < __builtin_inff() ? c : 0.f;
      cs[l32] = c;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float x = 0.f;
      {
        const float4* cp4 = reinterpret_cast<const float4*>(cs);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 v = cp4[j];
          x = fmaf(trow[4 * j], v.x, x); x = fmaf(trow[4 * j + 1], v.y, x); x = fmaf(trow[4 * j + 2], v.z, x); x = fmaf(trow[4 * j + 3], v.w, x);
        }
      }
#ifdef CHAIN_CLOCK
      { const unsigned long long tt = __builtin_amdgcn_s_memtime(); c_pro += tt - c_t0; c_t0 = tt; }
#endif
      // A branch on a condition that comes out of a vector compare costs ~50 cycles on a lone wave (tools/micro/lone_wave.hip), so
      // the loop of the usual correction -- a lane in the translated-exponential regime whose first candidate there is accepted --
      // is tested at its BOTTOM: one such branch per correction.  Everything else (nothing left to correct; a lane that needs its
      // other candidates) leaves it.
      uint32_t todo = lmask, bad = 0u, nrm = 0u, easy = 0u; float dlv = 0.f, xv = 0.f; int f = 0; bool hot = false;
      auto test = [&]() {                                           // lane f's correction and new value, where the lane can make them itself
        if (UPDATE == 0) {
          // every lane: its conditional mean, and -- should it turn out to be the first lane in the translated-exponential
          // regime -- its first candidate there (header): the lane a correction is due for has a final mean
          const float mu = x - o0.x, aa = mu * o0.w;
          const float rc = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(aa, aa, 4.0f)) + aa);
          const float tt = o1.x * rc;
          xv = o1.y * rc;
          dlv = xv - x;
          unsigned long long mc;
          asm("v_cmp_class_f32 %0, %1, %2" : "=s"(mc) : "v"(x), "s"(0x1C0));     // +0, +denormal, +normal
          nrm = (uint32_t)__builtin_amdgcn_ballot_w64(mu > o0.z);                 // normal regime: a = -mu sqrt(tau_p) < A0 (a dead entry: always)
          bad = todo & (~((uint32_t)mc & nrm) | forced);
          easy = (uint32_t)__builtin_amdgcn_ballot_w64(fabsf(tt) <= o1.z) & ~nrm & ~forced;
        } else {
          xv = fmaxf((o0.y > 0.f && x > 0.f) ? x : 0.f, a.min_x);
          dlv = xv - x;
          bad = todo & (~(uint32_t)__builtin_amdgcn_ballot_w64(xv == x) | forced);
          easy = ~forced;
        }
        hot = (bad & (0u - bad) & easy) != 0u;                       // the lowest lane to correct exists and makes its own value
        f = __builtin_ctz(bad | 0x80000000u);
      };
      auto apply = [&](float xnew, float dl) {
        todo &= 0xfffffffeu << f;
        x = fmaf(trow[f], dl, x);                                   // trow[f] = 0 in the lanes before f: their values stay as they are, bit for bit
        x = lane == f ? xnew : x;                                   // lane f takes its value as it is (the mirror lane f + 32 keeps its own: nothing reads the upper half's x)
      };
      test();
      for (;;) {
        while (hot) {
#ifdef CHAIN_CLOCK
          ++n_fix;
#endif
          apply(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xv), f)),
                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlv), f)));
          test();
        }
        if (bad == 0u) break;
#ifdef CHAIN_CLOCK
        ++n_fix; ++n_cold;
#endif
        float xnew;
        if (UPDATE == 0) {                                          // the other hoisted candidates (the first one in the normal regime), Philox rounds, overflow
          const float vf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x - o0.x), f));
          float xc = 0.f; unsigned long long m = 0ull;
          if (((forced >> f) & 1u) == 0u) {
            const float4 rec = recT[(k * L + f) * NH + (lane & (NH - 1))];
            const float aa = vf * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o0.w), f));
            const bool tail = ((nrm >> f) & 1u) == 0u;
            if (tail) {
              const float rc = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(aa, aa, 4.0f)) + aa);
              xc = rec.z * rc;
              m = __builtin_amdgcn_ballot_w64(fabsf(rec.y * rc) <= rec.w) & ((1ull << NH) - 1ull);
            } else {
              unsigned long long mt;
              xc = vf + rec.x;
              asm("v_cmp_class_f32 %0, %1, %2" : "=s"(mt) : "v"(xc), "s"(0x1C0));
              m = mt & ((1ull << NH) - 1ull);
            }
            if (m == 0ull) {                                        // all four rejected: Philox rounds, candidates NH + 64 round + lane
              const float sg = -4.0f * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o0.z), f));
              static_assert(kTnA0 == 0.25f, "sigma above is -thr / A0");
              TnFast tp; tp.mu = vf; tp.irt = sg; tp.a = aa; tp.live = true; tp.tail = tail;
              tp.d = 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(aa, aa, 4.0f)) + aa); tp.ilam = __builtin_amdgcn_rcpf(aa + tp.d);
              if (isfinite(aa) && sg < __builtin_inff())
                for (uint32_t round = 0; m == 0ull && round < 64u; ++round) {
                  const U4 rr = philox4x32_10(0u, (uint32_t)(k * L + f), a.it, kStreamS + 16u * ((uint32_t)NH + round * 64u + (uint32_t)lane), a.key0, a.key1);
                  const bool acc = tn_eval_fast(tp, rr.x, rr.y, &xc);
                  m = __builtin_amdgcn_ballot_w64(acc && isfinite(xc) && xc >= 0.0f);
                }
            }
          }
          if (m == 0ull) { xc = 0.f; m = 1ull; }                    // overflow, or 4100 rejections
          xnew = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xc), __builtin_ctzll(m)));
        } else {
          xnew = fmaxf(0.f, a.min_x);
        }
        apply(xnew, xnew - __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), f)));
        test();
      }
#ifdef CHAIN_CLOCK
      { const unsigned long long tt = __builtin_amdgcn_s_memtime(); c_fix += tt - c_t0; c_t0 = tt; }
#endif
      if (lane < L) { Sl[k * L + lane] = x; delta[cur][lane] = x - my_sold; }
      __syncthreads();
#ifdef CHAIN_CLOCK
      { const unsigned long long tt = __builtin_amdgcn_s_memtime(); c_wait += tt - c_t0; c_t0 = tt; }
#endif
    }
  } else if (wave == 4) {                                            // the stager: its own loop (its registers do not meet the fold's)
    for (int k = 0; k < K; ++k) {
      if (k + 1 < K) { stage_store(k + 1); stage_issue(min(k + 2, K - 1)); }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else {
#ifdef CHAIN_CLOCK
    unsigned long long b_work = 0, b_wait = 0, b_t0 = __builtin_amdgcn_s_memtime();
#endif
    auto row_end = [&]() {
#ifdef CHAIN_CLOCK
      { const unsigned long long tt = __builtin_amdgcn_s_memtime(); b_work += tt - b_t0; b_t0 = tt; }
#endif
      // LDS traffic only: the loads just issued stay in flight across the barrier
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef CHAIN_CLOCK
      { const unsigned long long tt = __builtin_amdgcn_s_memtime(); b_wait += tt - b_t0; b_t0 = tt; }
#endif
    };
    if (vec) {
      // two rows per trip, the odd last row peeled (a branch inside the trip makes the compiler move the sets between registers);
      // nothing else in this loop touches vector memory (a second path through it would set the wait counts to its worst case)
      auto bg_row = [&](int k, f32x4 (&fs)[QF][4]) {                // fs: issued two rows ago = the rows of A for row k-1's deltas
        consume_fold(k - 1, fs, delta[(k & 1) ^ 1]); issue_fold(k + 1, fs);   // (past the last rows: every offset beyond the matrix, zeros; every wave here folds: no branch)
        row_end();
      };
      int k = 0;
      for (; k + 1 < K; k += 2) {
        bg_row(k, fs1);                                             // row k (even) consumes the fold of row k-1 (odd): set 1
        bg_row(k + 1, fs0);
      }
      if (k < K) bg_row(k, fs1);
    } else {                                                        // L not a multiple of four: element by element, no prefetch
      for (int k = 0; k < K; ++k) {
        if (k > 0 && k + 1 < K) {
          const float* dp = delta[(k & 1) ^ 1];
          const int t0 = (k + 1) * L;
          for (int t = t0 + bt; t < n2; t += NB) {
            const float* col = a.A + (size_t)((k - 1) * L) * n2 + t;
            float s = r[t];
            for (int l0 = 0; l0 < L; l0 += 8) {
              float av[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) av[j] = l0 + j < L ? col[(size_t)(l0 + j) * n2] : 0.f;
#pragma unroll
              for (int j = 0; j < 8; ++j) s = fmaf(-dp[(l0 + j) & 31], av[j], s);
            }
            r[t] = s;
          }
        }
        row_end();
      }
    }
#ifdef CHAIN_CLOCK
    if (lane == 0 && a.it == 30u) printf("bg wave %d (simd %u): work %llu wait %llu\n", wave, (__builtin_amdgcn_s_getreg((3 << 11) | (4 << 6) | 4) & 3u), b_work, b_wait);   // HW_REG_HW_ID bits 5:4
#endif
  }
  for (int e = tid; e < n2; e += NT) a.S[e] = Sl[e];
#ifdef CHAIN_CLOCK
  if (tid == 0 && (a.it == 30u || a.it == 31u || a.it == 3000u)) printf("chain it %u: prologue %llu, rows (solve) %llu, fixes %llu (%d entries, %d cold), barrier %llu cycles\n", a.it, c_begin - c_k0, c_pro, c_fix, n_fix, n_cold, c_wait);
#endif
}
void launch_ssys_chain(const SSysChainArgs& a, hipStream_t st) {
#ifdef BNMTF_EXPERIMENTS
  static const bool seq = getenv("BNMTF_SCHAIN") != nullptr && !strcmp(getenv("BNMTF_SCHAIN"), "seq");
  if (seq) {
    if (a.update == 0) hipLaunchKernelGGL(ssys_chain_seq_kernel<0>, dim3(1), dim3(512), 0, st, a);
    else               hipLaunchKernelGGL(ssys_chain_seq_kernel<1>, dim3(1), dim3(512), 0, st, a);
    return;
  }
#endif
  if (a.update == 0) hipLaunchKernelGGL(ssys_chain_kernel<0>, dim3(1), dim3(512), 0, st, a);
  else               hipLaunchKernelGGL(ssys_chain_kernel<1>, dim3(1), dim3(512), 0, st, a);
}

}  // namespace bnmtf
