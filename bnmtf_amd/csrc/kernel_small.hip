// Small models: the whole run(n) of one BNMF Gibbs model in ONE launch, one 16-wave block (one CU) per model; a batch of
// independent models -- the folds x ranks x restarts of a model search -- is one grid.
//
// The reference's loop (bnmf_gibbs_optimised.py:133-155; tauU/muU :167-171, tauV/muV :173-177) in the exact Gram +
// sparse-complement form of the large path (kernel_sweep.hip header), laid out for ONE CU:
//
//  * both factors live in LDS for the whole launch, row major with an odd row stride (33): a gather of column k over
//    arbitrary rows, a row read by the MFMA B fragment and a per-unit write are all (nearly) conflict-free.  The region of
//    the factor being updated holds, during its half sweep, the running numerator base
//        G_uk = tau P_uk - lambda_uk - tau sum_l x_ul C_lk
//    instead (P = R~ . Xo the masked contraction, C = Xo^T Xo); entry (u, k) is replaced by the new x_uk when column k is
//    done, so after the sweep the region holds the new factor and no relayout pass exists.
//  * the contraction is the block's own: v_mfma_f32_16x16x4_f32 tiles, R~ streamed from L2 with the unit index on the
//    lanes (1, 2 or 4 units per lane and load, picked so that the items fill the four SIMDs), the other factor's rows as
//    the B fragment straight from LDS; the K x K term - X.C rides in the same accumulators (32 more inner steps); the
//    epilogue scales by tau and subtracts lambda.
//  * q_ij = U_i . V_j on the MISSING entries sits in registers of "entry threads" (up to 32 slots per thread, a unit's
//    entries over consecutive threads); a column is: deferred update of column k-1 + gather of column k + two partial sums
//    per entry thread -> LDS -> the unit's own thread (thread u <-> unit u) sums its segment, forms (numer, tau_p), draws.
//    After the draw the unit thread folds delta into the numerator bases of the columns still to come.
//  * draws: Philox-4x32-10 keyed exactly as in the large kernels (row, column, iteration, stream | candidate << 4), first
//    accepted candidate of the fixed candidate sequence (oracle/rng.py).  Candidate 0 is evaluated by the unit thread; the
//    units that rejected it are compacted into a block-wide list and get their next W candidates evaluated by W lanes
//    each (W = 1024 / rejected, a power of two <= 64), again until the list is empty: no thread loops over a divergent
//    rejection chain.
//  * between the half sweeps q is handed over through a per-model array in L2 (the other direction holds the same entries
//    in another order: a permutation table per direction); the rows sweep of every `refresh`-th iteration rebuilds q from
//    the factors, as the large path does.
//  * Gram matrices in fp64 on v_mfma_f64_16x16x4_f64 from the LDS copy; masked SSE / MSE / R^2 / Rp from the Gram
//    identities (kernel_misc.hip finish_kernel: same formulas), tau from the host-staged Gamma variate.
#include <algorithm>
#include <type_traits>

#include "../../include/bnmtf_hip.h"
#include "kernels.h"
#include "sweep_common.h"

namespace bnmtf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x2n __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2n __attribute__((ext_vector_type(2)));

// Pointers that come out of the launch descriptor are generic to the compiler: it would emit FLAT loads and stores, which also
// count on the LDS counter -- so that every LDS-only barrier of the column loop would wait for the global traffic in flight
// (the first version of this kernel: 60 us per toy iteration).  G(p): the same pointer in the global address space.
template <typename T> __device__ __forceinline__ __attribute__((address_space(1))) T* G(T* p) { return (__attribute__((address_space(1))) T*)p; }
#ifdef BNMTF_SMALL_TIMING
// debug build only (tools/variant.sh small_timing kernel_small.hip -DBNMTF_SMALL_TIMING): shader-clock sums per phase, printed by block 0
#define STAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); ph[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
constexpr int kS = kSmallStride;
// Two block barriers.  bar_lds: LDS traffic only -- global loads issued ahead (the next column's old values) and global stores
// (samples, state) stay in flight across it; a wave waiting for its own global stores in front of every barrier of the
// column loop is what made the first version of this kernel as slow as the multi-launch path.  bar_all: also the wave's
// global stores and loads -- where another thread of the block reads what this one wrote to global memory.
__device__ __forceinline__ void bar_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void bar_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// misc area behind the factors and the two Gram copies (floats), for a block of NT threads: the fixed part first, then the
// sweep's exchange arrays -- which the Gram's scratch re-uses between the sweeps
template <int NT> struct Misc {
  static constexpr int cnt = 0;                    // int [4] lengths of the three rotating lists of rejected units
  static constexpr int red = 4;                    // double [NT / 64][4] + [4] block reductions
  static constexpr int tau = red + 136;            // float tau
  static constexpr int c64 = tau + 4;              // double [32][32] Gram of the rows factor (the end of the iteration takes <C_rows, C_cols> from it)
  static constexpr int colsum = c64 + 2048;        // double [2][32] column sums of the two factors
  static constexpr int part = colsum + 128;        // float2 [NT] partial (sum q v, sum v^2) of every entry thread
  static constexpr int numer = part + 2 * NT;      // [NT] numer of a unit that rejected candidate 0
  static constexpr int taup = numer + NT;          // [NT] its tau_p
  static constexpr int xk = taup + NT;             // [NT] its old value
  static constexpr int dl = xk + NT;               // [NT] delta of the column just drawn, per unit
  static constexpr int xnw = dl + NT;              // [NT] draw made in a retry round, per unit
  static constexpr int list = xnw + NT;            // uint16 [3][NT] rejected units
  static constexpr int floats = list + 3 * NT / 2;
  // Gram scratch (between the sweeps, from `part` on): NT / 256 waves x 3 tiles x 256 doubles, 8 slices x 32 doubles
  static constexpr int gram_waves = NT / 256;
  static_assert(gram_waves * 3 * 256 * 2 + 8 * 32 * 2 <= floats - part, "the Gram scratch re-uses the sweep's exchange area");
  static_assert(red % 2 == 0 && c64 % 2 == 0 && part % 2 == 0, "doubles are 8-byte aligned");
  static_assert(8 * NT <= floats - part, "the tri-factorisation's packed second moments go through the exchange area eight at a time");
};

// (tri-factorisation, L > 0: behind the two Gram copies the Gram of F [K rows], the Gram of G [L rows], S [K rows] and S^T [L rows])
__host__ __device__ inline int small_tri_floats(int K, int L) { return L > 0 ? (2 * K + 2 * L) * kS : 0; }
// The S step's dense form (the K L x K L system in LDS, the chain on one wave) for the ranks the reference searches (K, L <= 10:
// 55 packed second moments per unit fit a thread's registers, the system is <= 100 x 100).  Its system and the packed products
// take the F region's place while the chain runs (F comes back from global memory) -- or, when that region is too small, space
// of their own behind the exchange arrays.
constexpr int kTriPairs = kTriDenseK * (kTriDenseK + 1) / 2;            // (kTriDenseK, small_tri_dense: kernels.h -- the path rule asks, too)
__host__ __device__ inline int small_tri_dense_floats(int K, int L) { return ((K * L * K * L + 3) & ~3) + 64 * 64; }
__host__ __device__ inline bool small_tri_dense_overlays(int I, int K, int L) { return (I + 1) * kS >= small_tri_dense_floats(K, L); }
__host__ __device__ inline int tri_pair_index(int a, int b) { return a * kTriDenseK - a * (a - 1) / 2 + (b - a); }      // a <= b < 10
__device__ __forceinline__ void tri_pair(int p, int& a, int& b) {
  int k = 0, st = 0;
#pragma unroll
  for (int t = 0; t < kTriDenseK - 1; ++t) { const int nxt = st + (kTriDenseK - k); if (p >= nxt) { st = nxt; ++k; } }
  a = k; b = k + (p - st);
}
__host__ __device__ inline int small_misc_offset(int I, int J, int K = 0, int L = 0) { return ((I + 1) * kS + (J + 1) * kS + 2 * 32 * kS + small_tri_floats(K, L) + 3) & ~3; }
size_t small_lds_bytes(int I, int J, int nt, int K, int L) {
  const int extra = (small_tri_dense(K, L) && !small_tri_dense_overlays(I, K, L)) ? small_tri_dense_floats(K, L) : 0;
  return sizeof(float) * (size_t)(small_misc_offset(I, J, K, L) + (nt <= 256 ? Misc<256>::floats : (nt <= 512 ? Misc<512>::floats : Misc<1024>::floats)) + extra);
}

// sum of four doubles over the block, in a fixed order (wave: xor butterfly; block: wave 0 .. 15); every thread gets the sums
template <int NT>
__device__ __forceinline__ void block_sum4(double v[4], double* red, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v[t] += __shfl_xor(v[t], m, 64);
  }
  bar_lds();
  if (lane == 0) { red[wave * 4 + 0] = v[0]; red[wave * 4 + 1] = v[1]; red[wave * 4 + 2] = v[2]; red[wave * 4 + 3] = v[3]; }
  bar_lds();
  if (tid < 4) {
    double s = 0.0;
#pragma unroll 4
    for (int w = 0; w < NT / 64; ++w) s += red[w * 4 + tid];
    red[64 + tid] = s;
  }
  bar_lds();
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = red[64 + t];
}

// C = X^T X (fp64) of the factor in `reg` ([n][33] in LDS): v_mfma_f64_16x16x4_f64 over 4 units per step.  The steps are
// dealt to FOUR groups (step j to group j mod 4) whose partial tiles are added as (g0 + g1) + (g2 + g3) -- by four waves, two
// waves or one wave depending on the block size, but always in that order: a model gives the same bits in a 256-thread block
// of its own and in a 1024-thread block of a mixed batch.  Results, all in LDS: Cs ([32][33], fp32: what the other direction's
// sweep reads), the column sums (fp64, 8 slices added in slice order), and either the fp64 Gram itself (c64_out: the rows
// factor) or <C_other, C> (c64_dot: the cols factor, at the end of the iteration; returned by thread 0, zero elsewhere).
// Columns >= K of the region are zero.
template <int NT>
__device__ __forceinline__ double small_gram(const float* reg, int n, int K, float* scratch, float* Cs, double* colsum, double* c64_out, const double* c64_dot, int tid,
                                             int rows_out = 32) {
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  constexpr int GW = Misc<NT>::gram_waves;
  double* scr = reinterpret_cast<double*>(scratch);         // [GW waves][3 tiles][256]
  double* scs = scr + GW * 3 * 256;                         // [8 slices][32 columns]
  const bool two = K > 16;
  bar_lds();
  if (wave < GW) {
    auto group = [&](int g, f64x4& a00, f64x4& a01, f64x4& a11) {
      a00 = f64x4{0, 0, 0, 0}; a01 = a00; a11 = a00;
      for (int u0 = 4 * g; u0 < n; u0 += 16) {
        const int u = u0 + lk;
        const double x0 = u < n ? (double)reg[u * kS + li] : 0.0;
        const double x1 = (u < n && two) ? (double)reg[u * kS + 16 + li] : 0.0;
        a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, a00, 0, 0, 0);
        if (two) {
          a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, a01, 0, 0, 0);
          a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, a11, 0, 0, 0);
        }
      }
    };
    f64x4 s00, s01, s11;
    if constexpr (GW == 4) group(wave, s00, s01, s11);
    else {
      f64x4 t00, t01, t11;
      group(GW == 2 ? 2 * wave : 0, s00, s01, s11);
      group(GW == 2 ? 2 * wave + 1 : 1, t00, t01, t11);
      s00 += t00; s01 += t01; s11 += t11;
      if constexpr (GW == 1) {
        f64x4 v00, v01, v11;
        group(2, t00, t01, t11);
        group(3, v00, v01, v11);
        t00 += v00; t01 += v01; t11 += v11;
        s00 += t00; s01 += t01; s11 += t11;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      scr[(wave * 3 + 0) * 256 + r * 64 + lane] = s00[r];
      scr[(wave * 3 + 1) * 256 + r * 64 + lane] = s01[r];
      scr[(wave * 3 + 2) * 256 + r * 64 + lane] = s11[r];
    }
  }
  if (tid < 256) {   // column sums: 8 slices of the units, one column per lane
    const int k = tid & 31, sl = tid >> 5;
    double cs = 0.0;
    for (int u = sl; u < n; u += 8) cs += (double)reg[u * kS + k];
    scs[sl * 32 + k] = cs;
  }
  bar_lds();
  for (int te = tid; te < 768; te += NT) {
    const int tile = te >> 8, e = te & 255, r = e >> 6, ln = e & 63;
    double s;
    if constexpr (GW == 4) s = (scr[tile * 256 + e] + scr[(3 + tile) * 256 + e]) + (scr[(6 + tile) * 256 + e] + scr[(9 + tile) * 256 + e]);
    else if constexpr (GW == 2) s = scr[tile * 256 + e] + scr[(3 + tile) * 256 + e];
    else s = scr[tile * 256 + e];
    const int a = (ln >> 4) + 4 * r + (tile == 2 ? 16 : 0), b = (ln & 15) + (tile >= 1 ? 16 : 0);    // C/D of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 reg
    if (a < rows_out) Cs[a * kS + b] = (float)s;          // (rows_out < 32: a Gram copy that holds the factor's own width only -- the rest is zero)
    if (c64_out) c64_out[a * 32 + b] = s;
    double pr = 0.0;
    if (c64_dot) pr = c64_dot[a * 32 + b] * s;
    if (tile == 1) {
      if (b < rows_out) Cs[b * kS + a] = (float)s;
      if (c64_out) c64_out[b * 32 + a] = s;
      if (c64_dot) pr += c64_dot[b * 32 + a] * s;
    }
    scr[tile * 256 + e] = pr;            // (this thread alone reads and writes entry (tile, e))
  }
  if (tid >= NT - 32) {
    const int k = tid - (NT - 32);
    double s = 0.0;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) s += scs[sl * 32 + k];
    colsum[k] = s;
  }
  bar_lds();
  double dot = 0.0;
  if (c64_dot && tid < 64) {
#pragma unroll 4
    for (int j = 0; j < 12; ++j) dot += scr[tid + 64 * j];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
    if (tid != 0) dot = 0.0;
  }
  bar_lds();
  return dot;
}

// The masked contraction with the K x K term and the epilogue: regOwn[u][k] = tau (sum_r big[r][u] Xo[r][k] - sum_l X[u][l] C[l][k]) - lambda[u][k].
// An item = 16 UW units x 16 columns: lane (i = lane & 15, kk = lane >> 4) loads UW consecutive units of inner row r0 + kk
// (A fragments of UW MFMAs: MFMA t covers the units u0 + UW i + t) and one B element Xo[r0 + kk][k0 + i] from LDS.  R~ comes
// from L2 (a microsecond away): the loads of the next NB steps are in flight while the MFMAs of the current NB run; `big` has
// round_up(m, 32) rows, the ones behind m zero.
struct ContractArgs {
#ifdef BNMTF_SMALL_TIMING
  unsigned long long* cph;
#endif
  const float* big; const float* XT; const float* lambda; float* PT;
  int n, m, ldb, ldn, K; float tau;
  uint32_t regO_b, CsO_b, regOwn_b;       // LDS byte addresses of the other factor, its Gram, and the own factor's region
  bool raw = false;                       // the bare product sum_r big[r][u] Xo[r][k] into PT, nothing else (the tri-factorisation's Pv = R~^T F)
};
template <int UW, int NT>
__device__ __forceinline__ void small_contract(const ContractArgs& d, int tid) {
  lds_cf* regO = (lds_cf*)(uintptr_t)d.regO_b;
  lds_cf* CsO = (lds_cf*)(uintptr_t)d.CsO_b;
  lds_fp regOwn = (lds_fp)(uintptr_t)d.regOwn_b;
  const float tau = d.tau; const int K = d.K;
  float* const PT = d.PT;
  constexpr int NB = UW == 4 ? 4 : 8;
  auto load_units = [](const float* p, float (&av)[UW]) {
    if constexpr (UW == 4) { const f32x4 t = *G(reinterpret_cast<const f32x4*>(p)); av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w; }
    else if constexpr (UW == 2) { const f32x2n t = *G(reinterpret_cast<const f32x2n*>(p)); av[0] = t.x; av[1] = t.y; }
    else av[0] = *G(p);
  };
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int kt = (K + 15) >> 4, ub = (d.n + 16 * UW - 1) / (16 * UW), items = ub * kt;
  // batches of NB inner steps (4 rows each): nb_big of R~ (its rows behind m are zero), then the K x K term's: A = X^T (rows
  // behind K zero), B = -C
  const int nb_big = ((d.m + 31) & ~31) / (4 * NB), nb = d.raw ? nb_big : nb_big + (K + 4 * NB - 1) / (4 * NB);
  for (int item = wave; item < items; item += NT / 64) {
    const int u0 = (item / kt) * 16 * UW, k0 = (item % kt) * 16;
    f32x4 acc[UW], pacc[UW];
#pragma unroll
    for (int t = 0; t < UW; ++t) acc[t] = pacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ap = d.big + u0 + UW * li + (size_t)lk * d.ldb;
    const float* xp = d.XT + u0 + UW * li + (size_t)lk * d.ldn;
    auto load_batch = [&](int bi, float (&dst)[NB][UW]) {
      const float* src = bi < nb_big ? ap + (size_t)(bi * 4 * NB) * d.ldb : xp + (size_t)((bi - nb_big) * 4 * NB) * d.ldn;
      const int ld = bi < nb_big ? d.ldb : d.ldn;
#pragma unroll
      for (int s = 0; s < NB; ++s) load_units(src + (size_t)(4 * s) * ld, dst[s]);
    };
    // operand sets used in turn (no register copies: a copy would wait for the loads it is meant to overlap): two sets for the
    // wide items -- the loads of batch bi + 1 in flight while the 16 MFMAs of batch bi run --, four for the one-unit-per-lane items,
    // whose batch is 8 dependent MFMAs (320 cycles) against an L2 round trip of 500-900; the batch's B elements come out of LDS
    // ahead of its MFMAs
    auto run_batch = [&](int bi, const float (&cur)[NB][UW]) {
      if (bi == nb_big) {
#pragma unroll
        for (int t = 0; t < UW; ++t) pacc[t] = acc[t];
      }
      const bool big = bi < nb_big;
      const int r0 = (big ? bi : bi - nb_big) * 4 * NB + lk;
      float bv[NB];
      if (big) {
#pragma unroll
        for (int s = 0; s < NB; ++s) bv[s] = regO[min(r0 + 4 * s, d.m) * kS + k0 + li];        // rows >= m of the factor: its zero row
      } else {
#pragma unroll
        for (int s = 0; s < NB; ++s) bv[s] = -CsO[(r0 + 4 * s) * kS + k0 + li];
      }
#pragma unroll
      for (int s = 0; s < NB; ++s)
#pragma unroll
        for (int t = 0; t < UW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[s][t], bv[s], acc[t], 0, 0, 0);
    };
#ifdef BNMTF_SMALL_TIMING
    unsigned long long ct0 = __builtin_readcyclecounter();
#endif
    if constexpr (UW == 1 && NT < 1024) {       // (a 16-wave block has 128 registers per lane: two sets there)
      float o0[NB][UW], o1[NB][UW], o2[NB][UW], o3[NB][UW];
      load_batch(0, o0);
      if (nb > 1) load_batch(1, o1);
      if (nb > 2) load_batch(2, o2);
      for (int bi = 0; bi < nb; bi += 4) {
        if (bi + 3 < nb) load_batch(bi + 3, o3);
        run_batch(bi, o0);
        if (bi + 1 >= nb) break;
        if (bi + 4 < nb) load_batch(bi + 4, o0);
        run_batch(bi + 1, o1);
        if (bi + 2 >= nb) break;
        if (bi + 5 < nb) load_batch(bi + 5, o1);
        run_batch(bi + 2, o2);
        if (bi + 3 >= nb) break;
        if (bi + 6 < nb) load_batch(bi + 6, o2);
        run_batch(bi + 3, o3);
      }
    } else {
      float opa[NB][UW], opb[NB][UW];
      load_batch(0, opa);
      for (int bi = 0; bi < nb; bi += 2) {
        if (bi + 1 < nb) load_batch(bi + 1, opb);
        run_batch(bi, opa);
        if (bi + 1 >= nb) break;
        if (bi + 2 < nb) load_batch(bi + 2, opa);
        run_batch(bi + 1, opb);
      }
    }
#ifdef BNMTF_SMALL_TIMING
    { const unsigned long long t_ = __builtin_readcyclecounter(); d.cph[0] += t_ - ct0; ct0 = t_; d.cph[2] += 1; }
#endif
    // C/D of the 16x16 f32 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
    const int k = k0 + li;
    if (d.raw) {
#pragma unroll
      for (int t = 0; t < UW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int u = u0 + UW * (4 * lk + r) + t;
          if (u < d.n && k < K) *G(PT + (size_t)k * d.ldn + u) = acc[t][r];
        }
      continue;
    }
    float lam[UW][4];                      // the prior rates: all loads first (one after the other they were 16 L2 round trips per item)
#pragma unroll
    for (int t = 0; t < UW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int u = min(u0 + UW * (4 * lk + r) + t, d.n - 1);          // (unconditional: the loads go out together)
        lam[t][r] = *G(d.lambda + u * 32 + k);
      }
#pragma unroll
    for (int t = 0; t < UW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int u = u0 + UW * (4 * lk + r) + t;
        if (u < d.n && k < K) {
          regOwn[u * kS + k] = fmaf(tau, acc[t][r], -lam[t][r]);
          if (PT) *G(PT + (size_t)k * d.ldn + u) = pacc[t][r];
        }
      }
#ifdef BNMTF_SMALL_TIMING
    { const unsigned long long t_ = __builtin_readcyclecounter(); d.cph[1] += t_ - ct0; }
#endif
  }
}

// The in-place products of the tri-factorisation: row u of `reg` (Win values) becomes its product with S (TR = false:
// out_c = sum_w x_w S[w][c], S as Sm = S) or with S^T (the caller passes Sm = S^T), Wout values and zeros up to 32.  One thread per
// row, the row in registers (WB = Win rounded up to 8; the rows of Sm are zero behind Win).
template <int WB, int NT>
__device__ __forceinline__ void small_row_product(float* reg, int n, int Wout, const float* SmT, int tid) {
  for (int u = tid; u < n; u += NT) {
    float x[WB];
#pragma unroll
    for (int w = 0; w < WB; ++w) x[w] = reg[u * kS + w];
    for (int c = 0; c < 32; ++c) {
      float acc = 0.f;
      if (c < Wout) {
#pragma unroll
        for (int w = 0; w < WB; ++w) acc = fmaf(x[w], SmT[c * kS + w], acc);      // SmT[c][w] = Sm[w][c]: out_c = sum_w x_w Sm[w][c]
      }
      reg[u * kS + c] = acc;
    }
  }
}
template <int NT>
__device__ __forceinline__ void small_times_S(float* reg, int n, int Win, int Wout, const float* SmT, int tid) {
  if (Win <= 8) small_row_product<8, NT>(reg, n, Wout, SmT, tid);
  else if (Win <= 16) small_row_product<16, NT>(reg, n, Wout, SmT, tid);
  else small_row_product<32, NT>(reg, n, Wout, SmT, tid);
}

template <int EM, int NT, bool TRI>
__global__ __launch_bounds__(NT) void small_gibbs_kernel(const SmallLaunch* __restrict__ all) {
  extern __shared__ float lds[];
  const SmallLaunch& L = all[blockIdx.x];
  const int tid = threadIdx.x;
  const int I = L.rows.n, J = L.cols.n, K = L.K;
  const int Lc = TRI ? L.L : K;            // width of the cols factor (tri-factorisation: G is J x L)
  const bool tri_dense = TRI && small_tri_dense(K, Lc);
  float* regR = lds;
  float* regC = regR + (I + 1) * kS;
  float* CsR = regC + (J + 1) * kS;
  float* CsC = CsR + 32 * kS;
  // tri-factorisation: Gram of F [K rows], Gram of G [L rows], S [K rows], S^T [L rows] (rows of 33, zero behind the widths)
  float* CfT = CsC + 32 * kS;
  float* CgT = CfT + (TRI ? K : 0) * kS;
  float* Ss = CgT + (TRI ? Lc : 0) * kS;
  float* SsT = Ss + (TRI ? K : 0) * kS;
  float* misc = lds + small_misc_offset(I, J, K, TRI ? Lc : 0);
  typedef Misc<NT> MS;
  float2* part = reinterpret_cast<float2*>(misc + MS::part);
  float* numer_s = misc + MS::numer; float* taup_s = misc + MS::taup; float* xk_s = misc + MS::xk;
  float* dl = misc + MS::dl; float* xnw = misc + MS::xnw;
  uint16_t* lists = reinterpret_cast<uint16_t*>(misc + MS::list);
  int* cnt = reinterpret_cast<int*>(misc + MS::cnt);
  double* red = reinterpret_cast<double*>(misc + MS::red);
  float* tau_s = misc + MS::tau;
  double* c64R = reinterpret_cast<double*>(misc + MS::c64);
  double* csum = reinterpret_cast<double*>(misc + MS::colsum);
  float* gscr = misc + MS::part;

  // ---- the state into LDS: both factors (zero pads, zero row behind the last), the transposed copies the column loop reads
  for (int t = tid; t < (I + 1) * kS; t += NT) { const int u = t / kS, k = t - u * kS; regR[t] = (u < I && k < K) ? L.rows.X[u * 32 + k] : 0.f; }
  for (int t = tid; t < (J + 1) * kS; t += NT) { const int u = t / kS, k = t - u * kS; regC[t] = (u < J && k < Lc) ? L.cols.X[u * 32 + k] : 0.f; }
  for (int t = tid; t < 2 * 32 * kS + (TRI ? small_tri_floats(K, Lc) : 0); t += NT) CsR[t] = 0.f;
  for (int t = tid; t < I * K; t += NT) { const int u = t % I, k = t / I; L.rows.XT[(size_t)k * L.rows.ldn + u] = L.rows.X[u * 32 + k]; }
  for (int t = tid; t < J * Lc; t += NT) { const int u = t % J, k = t / J; L.cols.XT[(size_t)k * L.cols.ldn + u] = L.cols.X[u * 32 + k]; }
  if (tid < 4) cnt[tid] = 0;
  if (tid == 0) { *tau_s = *L.tau_f; L.clock[0] = wall_clock64(); }
  bar_all();
  if constexpr (TRI) {
    for (int t = tid; t < K * Lc; t += NT) { const int k = t / Lc, l = t - k * Lc; const float v = L.S[t]; Ss[k * kS + l] = v; SsT[l * kS + k] = v; }
    small_gram<NT>(regC, J, Lc, gscr, CgT, csum + 32, nullptr, nullptr, tid, Lc);      // (the Grams of the effective factors are formed ahead of each half sweep)
  } else {
    small_gram<NT>(regR, I, K, gscr, CsR, csum, c64R, nullptr, tid);
    small_gram<NT>(regC, J, K, gscr, CsC, csum + 32, nullptr, nullptr, tid);
  }

  int rr = 0;                                // running retry-round number: list rr % 3 is the one being filled / read
  const bool draw = L.update == BNMTF_UPDATE_DRAW;
#ifdef BNMTF_SMALL_TIMING
  unsigned long long ph[24] = {0}, cph[6] = {0}, tlast = __builtin_readcyclecounter();
  int nretry = 0;
#endif

  for (int it = 0; it < L.n_iter; ++it) {
    const unsigned long long it_abs = L.it0 + (unsigned long long)it;
    const uint32_t it32 = (uint32_t)it_abs;
    double st_px = 0.0, st_q = 0.0, st_q2 = 0.0, st_dot = 0.0;       // statistics of the cols sweep (this thread's share)

#pragma unroll 1
    for (int dir = 0; dir < 2; ++dir) {
      // (an opaque copy of the thread index per half sweep: the address arithmetic of this phase is not hoisted out of the
      // iteration loop, where it would sit in ~45 registers across every other phase)
      int tq = tid;
      asm volatile("" : "+v"(tq));
      const int lane = tq & 63;
      const SmallDirDev d = dir == 0 ? L.rows : L.cols;
      const float* oq = dir == 0 ? L.cols.q : L.rows.q;
      float* regOwn = dir == 0 ? regR : regC;
      const float* regO = dir == 0 ? regC : regR;
      const float* CsO = dir == 0 ? CsC : CsR;
      const uint32_t stream = dir == 0 ? kStreamRows : kStreamCols;
      const float tau = *tau_s;
      const int n = d.n, m = d.m, em = d.em;
      const bool entry = tq < d.nthreads, unit = tq < n;
      const int Kd = dir == 0 ? K : Lc;         // columns of this half sweep (= width of the own factor and of the other, effective, one)
      if constexpr (TRI) {
        // the other factor of this half sweep in place: G S^T (J x K) for the F sweep, F S (I x L) for the G sweep, and its Gram
        small_times_S<NT>(dir == 0 ? regC : regR, m, dir == 0 ? Lc : K, Kd, dir == 0 ? Ss : SsT, tq);
        if (Kd < 32)                               // (the own region held the other sweep's effective factor: nothing behind this sweep's width)
          for (int t = tq; t < n * (32 - Kd); t += NT) { const int u = t / (32 - Kd), k = Kd + t - u * (32 - Kd); regOwn[u * kS + k] = 0.f; }
        // (+ the fp64 Gram and the column sums of F S: the end of the iteration)
        small_gram<NT>(dir == 0 ? regC : regR, m, Kd, gscr, dir == 0 ? CsC : CsR, dir == 0 ? csum + 32 : csum, dir == 0 ? nullptr : c64R, nullptr, tq);
        STAMP(16);
      }

      // ---- the Philox words of the first nc0 candidates of every (unit, column) of this half sweep, by all threads at once
      // (they depend on the counters only; inside the column loop a Philox call is ~800 cycles of a wave on the critical path)
      constexpr int nc0 = 1;
      u32x2n* tab = reinterpret_cast<u32x2n*>(d.tab);
      if (draw)
        for (int e = tq; e < Kd * nc0 * d.ldn; e += NT) {
          const int u = e % d.ldn, kc = e / d.ldn;
          if (u < n) {
            const U4 r = philox4x32_10((uint32_t)u, (uint32_t)(kc / nc0), it32, stream + 16u * (uint32_t)(kc % nc0), L.key0, L.key1);
            *G(tab + e) = u32x2n{r.x, r.y};
          }
        }
      STAMP(0);
      // ---- contraction: regOwn = G (nothing of the sweep's per-thread state is live yet: the operand pipeline has the registers)
      {
        const int kt = (Kd + 15) >> 4;
        ContractArgs ca;
#ifdef BNMTF_SMALL_TIMING
        ca.cph = cph + 3 * dir;
#endif
        ca.big = d.big; ca.XT = d.XT; ca.lambda = d.lambda; ca.PT = dir == 1 ? L.PT : nullptr;
        ca.n = n; ca.m = m; ca.ldb = d.ldb; ca.ldn = d.ldn; ca.K = Kd; ca.tau = tau;
        ca.regO_b = (uint32_t)(uintptr_t)(lds_fp)regO; ca.CsO_b = (uint32_t)(uintptr_t)(lds_fp)CsO; ca.regOwn_b = (uint32_t)(uintptr_t)(lds_fp)regOwn;
        if (TRI && dir == 1) {
          // the tri-factorisation's G sweep: R~^T (F S) = (R~^T F) S from the S step's pass over R~ (Pv: L.ZT [K][ldn]); the own
          // factor's old values from its transposed copy (this loop overwrites the region)
          for (int o = tq; o < n * Kd; o += NT) {
            const int j = o / Kd, l = o - j * Kd;
            float pv = 0.f, kk = 0.f;
            for (int k2 = 0; k2 < K; ++k2) pv = fmaf(*G(L.ZT + (size_t)k2 * d.ldn + j), SsT[l * kS + k2], pv);
            for (int l2 = 0; l2 < Kd; ++l2) kk = fmaf(*G(d.XT + (size_t)l2 * d.ldn + j), CsO[l2 * kS + l], kk);
            regOwn[j * kS + l] = fmaf(tau, pv - kk, -*G(d.lambda + j * 32 + l));
            *G(L.PT + (size_t)l * d.ldn + j) = pv;
          }
        }
        else if (((n + 63) / 64) * kt >= NT / 64) small_contract<4, NT>(ca, tq);
        else if (((n + 31) / 32) * kt >= NT / 128) small_contract<2, NT>(ca, tq);
        else small_contract<1, NT>(ca, tq);
      }
      bar_all();                               // (PT is read by other threads than the ones that stored it)
      STAMP(1);

      // ---- this thread's slots, and q of its entries: handed over by the other direction's sweep, or -- every `refresh`-th
      // iteration in the rows sweep -- rebuilt from the factors (the own one as it was: its transposed copy in L2)
      uint32_t jj[EM / 2];
      // (EM = 64: the gathered values of the previous column are not kept -- 64 more registers -- but gathered again for the update)
      constexpr bool KEEPV = EM <= 32;
      float q[EM], vp[KEEPV ? EM : 2];
      int myunit = 0;
      if (entry) myunit = *G(d.unit_of + tq);
#pragma unroll
      for (int h = 0; h < EM / 2; ++h) {
        uint32_t j0 = (uint32_t)(m * kS), j1 = (uint32_t)(m * kS);      // (33 j: the first word of factor row j; 33 m: the zero row)
        if (entry && 2 * h < em) { j0 = *G(d.idx + (2 * h) * kSmallThreads + tq); j1 = *G(d.idx + (2 * h + 1) * kSmallThreads + tq); }
        jj[h] = j0 | (j1 << 16);
        q[2 * h] = q[2 * h + 1] = 0.f;
        if constexpr (KEEPV) vp[2 * h] = vp[2 * h + 1] = 0.f;
      }
      // (the dense S step does not carry q: the G sweep behind it forms q from F S and G)
      const bool prepass = (dir == 0 && (it_abs % L.refresh == 0 || (it == 0 && !L.q_valid))) || (dir == 1 && tri_dense);
      if (prepass) {
        if (entry)
          for (int k = 0; k < Kd; ++k) {
            const float xv = *G(d.XT + (size_t)k * d.ldn + myunit);
#pragma unroll
            for (int h = 0; h < EM / 2; ++h)
              if (2 * h < em) {
                q[2 * h] = fmaf(xv, regO[(jj[h] & 0xFFFFu) + k], q[2 * h]);
                q[2 * h + 1] = fmaf(xv, regO[(jj[h] >> 16) + k], q[2 * h + 1]);
              }
          }
      } else if (entry) {
#pragma unroll
        for (int e = 0; e < EM; ++e)
          if (e < em) {
            const uint32_t p = *G(d.perm + e * kSmallThreads + tq);
            q[e] = p != kSmallNone ? *G(oq + p) : 0.f;
          }
      }

      STAMP(2);
      // ---- the K sequential columns
      // (candidate role: thread (cc, cu) takes candidate cc < nc0 of unit cu's draw; with nc0 = 1 that is the unit's own thread)
      const int T = d.ldn;
      const int cc = tq / T, cu = tq - cc * T;
      const bool cand_on = draw && cc < nc0 && cu < n;
      int seg0 = 0, segn = 0;
      float xk_n = 0.f, pv_n = 0.f;             // column k + 1's old value and Pv, on their way from L2 during column k
      u32x2n cw_n = {0u, 0u};                   // ... and this thread's candidate words
      if (unit) {
        seg0 = *G(d.seg + 2 * tq); segn = *G(d.seg + 2 * tq + 1);
        xk_n = *G(d.XT + tq);
        if (dir == 1) pv_n = *G(L.PT + tq);
      }
      if (cand_on) cw_n = *G(tab + (size_t)cc * T + cu);
#pragma unroll 1
      for (int k = 0; k < Kd; ++k) {
        if (entry) {
          const float dlt = k > 0 ? dl[myunit] : 0.f;
          float qv = 0.f, vv = 0.f;
          const float* col = regO + k;
#pragma unroll
          for (int h = 0; h < EM / 2; ++h)
            if (2 * h < em) {
              asm volatile("" : "+v"(jj[h]));            // opaque: the 32 word addresses are made per gather, not kept in 32 more registers
              const float v0 = col[jj[h] & 0xFFFFu], v1 = col[jj[h] >> 16];
              if constexpr (KEEPV) {
                q[2 * h] = fmaf(dlt, vp[2 * h], q[2 * h]);
                q[2 * h + 1] = fmaf(dlt, vp[2 * h + 1], q[2 * h + 1]);
                vp[2 * h] = v0; vp[2 * h + 1] = v1;
              } else if (k > 0) {
                q[2 * h] = fmaf(dlt, col[(int)(jj[h] & 0xFFFFu) - 1], q[2 * h]);
                q[2 * h + 1] = fmaf(dlt, col[(int)(jj[h] >> 16) - 1], q[2 * h + 1]);
              }
              qv = fmaf(q[2 * h], v0, qv); vv = fmaf(v0, v0, vv);
              qv = fmaf(q[2 * h + 1], v1, qv); vv = fmaf(v1, v1, vv);
            }
          part[tq] = float2{qv, vv};
        }
        // what does not wait for the partial sums: this column's prefetched operands into place, the next column's on their way,
        // the candidate's word-only half (log, sqrt, cos)
        const float xk = xk_n, pv = pv_n;
        float gk = 0.f, ckk = 0.f;
        TnCand cand = {0.f, 0.f, 0.f};
        if (cand_on) cand = tn_cand_pre(cw_n.x, cw_n.y);
        if (k + 1 < Kd) {
          if (unit) {
            xk_n = *G(d.XT + (size_t)(k + 1) * T + tq);
            if (dir == 1) pv_n = *G(L.PT + (size_t)(k + 1) * T + tq);
          }
          if (cand_on) cw_n = *G(tab + ((size_t)(k + 1) * nc0 + cc) * T + cu);
        }
        if (unit) { gk = regOwn[tq * kS + k]; ckk = CsO[k * kS + k]; }
        STAMP(3);
        bar_lds();
        STAMP(4);
        bool done = true;
        float xnew = 0.f;
        if (unit) {
          float qv = 0.f, vv = 0.f;
          for (int g = seg0; g < seg0 + segn; ++g) { const float2 p = part[g]; qv += p.x; vv += p.y; }
          const float corr = fmaf(-xk, vv, qv);
          const float tau_p = tau * (ckk - vv);
          const float numer = fmaf(tau, fmaf(xk, ckk, corr), gk);
          if (draw) {
            // candidate 0 (its Philox words made in bulk, its word-only half before the barrier): accepted by ~3 draws in 4
            const TnFast tf = tn_fast_params(numer, tau_p);
            float xc;
            const bool acc = tn_cand_post(tf, cand, &xc);
            if (!tf.live) xnew = 0.f;                          // (a dead conditional draws 0: oracle/rng.py)
            else if (acc) xnew = tn_guard(xc);
            else {
              done = false;
              numer_s[tq] = numer; taup_s[tq] = tau_p; xk_s[tq] = xk;
              const int pos = atomicAdd(&cnt[rr % 3], 1);
              lists[(rr % 3) * NT + pos] = (uint16_t)tq;
            }
            if (done) dl[tq] = xnew - xk;
          } else {
            const float mu = numer / tau_p;
            xnew = fmaxf((tau_p > 0.f && mu > 0.f) ? mu : 0.f, L.min_x);
            dl[tq] = xnew - xk;
          }
        }
        bar_lds();
        STAMP(5);
        if (draw) {
          // ---- retry rounds: the units that rejected candidate 0 get their next W candidates evaluated at once, by as many waves
          // as that takes (a Philox call is ~740 cycles of a SIMD: speculative candidates are not free)
          uint32_t c0 = (uint32_t)nc0;
          for (;;) {
            const int cur = rr % 3, nxt = (rr + 1) % 3;
            const int nrej = cnt[cur];
            if (nrej == 0) break;
            int W = NT / nrej;
            W = W >= 64 ? 16 : (W >= 4 ? 4 : (W >= 2 ? 2 : 1));          // (a handful of stragglers: 16 candidates each, one round)
            if ((tq & ~63) < nrej * W) {
              const int li = tq / W, c = tq & (W - 1);
              const bool active = li < nrej;
              const int u = active ? (int)lists[cur * NT + li] : 0;
              const TnFast tf = tn_fast_params(numer_s[u], taup_s[u]);
              const U4 r = philox4x32_10((uint32_t)u, (uint32_t)k, it32, stream + 16u * (c0 + (uint32_t)c), L.key0, L.key1);
              float xc;
              const bool acc = tn_eval_fast(tf, r.x, r.y, &xc) && active;
              const unsigned long long mask = __ballot(acc);
              const int gbase = lane & ~(W - 1);
              const unsigned long long gm = (mask >> gbase) & ((1ull << W) - 1ull);
              const int src = gm ? gbase + __ffsll((long long)gm) - 1 : lane;
              const float xsel = __shfl(xc, src, 64);
              if (active && c == 0) {
                if (gm) { const float xn = tn_guard(xsel); dl[u] = xn - xk_s[u]; xnw[u] = xn; }
                else if (c0 + (uint32_t)W >= 4096u) { dl[u] = -xk_s[u]; xnw[u] = 0.f; }
                else { const int pos = atomicAdd(&cnt[nxt], 1); lists[nxt * NT + pos] = (uint16_t)u; }
              }
            }
            if (tq == 0) cnt[(rr + 2) % 3] = 0;
            c0 += (uint32_t)W;
            ++rr;
            bar_lds();
#ifdef BNMTF_SMALL_TIMING
            ++nretry;
#endif
          }
          STAMP(8);
        }
        // ---- the unit's thread: the new value into the factor's region, delta into the columns still to come
        if (unit) {
          if (!done) xnew = xnw[tq];
          const float delta = xnew - xk;
          *G(d.XT + (size_t)k * T + tq) = xnew;
          regOwn[tq * kS + k] = xnew;
          if (dir == 1) st_px += (double)pv * (double)xnew;
          const float t = tau * delta;
          for (int k2 = k + 1; k2 < Kd; k2 += 8) {
            float g8[8], c8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int kk = min(k2 + j, 31); g8[j] = regOwn[tq * kS + kk]; c8[j] = CsO[k * kS + kk]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) if (k2 + j < Kd) regOwn[tq * kS + k2 + j] = fmaf(-t, c8[j], g8[j]);
          }
        }
        STAMP(9);
      }
      // ---- the last column's update of q; q goes where the other direction finds it
      if (entry) {
        const float dlt = dl[myunit];
        double lq = 0.0, lq2 = 0.0;            // (summed here and added once: the iteration's accumulators live through every phase)
#pragma unroll
        for (int e = 0; e < EM; ++e)
          if (e < em) {
            float vlast;
            if constexpr (KEEPV) vlast = vp[e];
            else vlast = regO[((jj[e >> 1] >> (16 * (e & 1))) & 0xFFFFu) + (Kd - 1)];
            q[e] = fmaf(dlt, vlast, q[e]);
            *G(d.q + e * kSmallThreads + tq) = q[e];
            lq += (double)q[e]; lq2 += (double)q[e] * (double)q[e];
          }
        if (dir == 1) { st_q += lq; st_q2 += lq2; }
      }
      bar_all();                               // q and the transposed factor: read by other threads in the next half sweeps
      STAMP(10);
      // ---- Gram of the new factor; state, sample, posterior sums
      if constexpr (TRI) {       // the factors' own Grams are the S step's; the sweeps read the effective factors' (formed ahead of them)
        st_dot = small_gram<NT>(regOwn, n, Kd, gscr, dir == 0 ? CfT : CgT, dir == 0 ? csum : csum + 32, nullptr, dir == 0 ? nullptr : c64R, tq, Kd);
      } else {
        if (dir == 0) small_gram<NT>(regOwn, n, K, gscr, CsR, csum, c64R, nullptr, tq);
        else st_dot = small_gram<NT>(regOwn, n, K, gscr, CsC, csum + 32, nullptr, c64R, tq);
      }
      if (tq < 4) cnt[tq] = 0;                 // (no list is in flight here; rr keeps counting)
      STAMP(11);
      {
        float* smp = dir == 0 ? L.U_s : L.V_s;
        double* ex = dir == 0 ? L.expR : L.expC;
        const bool add = ex && L.exp_burn >= 0 && it >= L.exp_burn && (it - L.exp_burn) % L.exp_thin == 0;
        for (int t = tq; t < n * Kd; t += NT) {
          const int u = t / Kd, k = t - u * Kd;
          const float v = regOwn[u * kS + k];
          *G(d.X + u * 32 + k) = v;
          if (smp) *G(smp + ((size_t)it * n + u) * Kd + k) = v;
          if (add) *G(ex + u * 32 + k) += (double)v;
        }
      }
      bar_lds();
      STAMP(12);

      // ================================================= the S step of the tri-factorisation (bnmtf_gibbs_optimised.py:157-160, 201-205)
      // The conditional of S_kl needs sums over the OBSERVED entries of w_ij = F_ik G_jl; in the exact Gram + sparse-complement form
      //   tau_p = tau (Cf_kk Cg_ll - sum_miss w^2),   numer = -lambda + tau (r_kl + sum_miss q w + S_kl (Cf_kk Cg_ll - sum_miss w^2)),
      //   r = F^T R~ G - Cf S Cg (kept current: r -= delta Cf[:, k] (x) Cg[l, :]),   q_ij = (F S G^T)_ij on the missing entries.
      // Two forms.  Sequential: q in the registers of the F sweep's entry threads (a thread's entries share i: one F_ik
      // per thread and row of S), the two sums over the missing entries = the column loop's gather of G's column l, reduced over the
      // whole block by wave 0, which draws.  Dense (K, L <= 10): the whole system A = Cf (x) Cg - sum_miss (f f^T) (x) (g g^T) in LDS,
      // the K L steps on one wave.
      if constexpr (TRI) if (dir == 0) {
        const int n2 = K * Lc;
        float* rS = reinterpret_cast<float*>(c64R);          // [K L] (the fp64 Gram's area is free until the G sweep's effective factor is formed)
        float* tmpS = rS + 1024;
        for (int t = tq; t < J * kS; t += NT) { const int u = t / kS, k = t - u * kS; regC[t] = k < Lc ? *G(L.cols.X + u * 32 + k) : 0.f; }     // G back into its region
        if (draw && !tri_dense)                    // (the dense form keeps its candidates in LDS)
          for (int e = tq; e < n2 * 4; e += NT) {
            const U4 r = philox4x32_10(0u, (uint32_t)(e >> 2), it32, kStreamS + 16u * (uint32_t)(e & 3), L.key0, L.key1);
            *G(reinterpret_cast<u32x2n*>(L.stab) + e) = u32x2n{r.x, r.y};
          }
        for (int pp = tq; pp < n2 && !tri_dense; pp += NT) {               // T = S Cg
          const int kk = pp / Lc, ll = pp - kk * Lc;
          float a = 0.f;
          for (int l2 = 0; l2 < Lc; ++l2) a = fmaf(Ss[kk * kS + l2], CgT[l2 * kS + ll], a);
          tmpS[pp] = a;
        }
        bar_lds();
        {
          ContractArgs ca;
#ifdef BNMTF_SMALL_TIMING
          ca.cph = cph;
#endif
          // Pv = R~^T F (J x K): b = Pv^T G here, and R~^T (F S) = Pv S for the G sweep -- its pass over R~ is this one
          const SmallDirDev& dc = L.cols;
          ca.big = dc.big; ca.XT = dc.XT; ca.lambda = dc.lambda; ca.PT = L.ZT;
          ca.n = dc.n; ca.m = dc.m; ca.ldb = dc.ldb; ca.ldn = dc.ldn; ca.K = K; ca.tau = tau; ca.raw = true;
          ca.regO_b = (uint32_t)(uintptr_t)(lds_fp)regR; ca.CsO_b = (uint32_t)(uintptr_t)(lds_fp)CsC; ca.regOwn_b = (uint32_t)(uintptr_t)(lds_fp)regC;
          small_contract<2, NT>(ca, tq);           // (one shape for this product: a third of the sweep's)
        }
        for (int pp = tq; pp < n2; pp += NT) {               // Cf T
          const int kk = pp / Lc, ll = pp - kk * Lc;
          float a = 0.f;
          for (int k2 = 0; k2 < K && !tri_dense; ++k2) a = fmaf(CfT[kk * kS + k2], tmpS[k2 * Lc + ll], a);
          rS[pp] = a;                            // (dense: 0 -- r = b - A S follows when A is there)
        }
        bar_all();                               // (Pv is read by other threads than the ones that stored it)
        for (int p0 = 0; p0 < n2; p0 += NT / 8) {                // r = Pv^T G - Cf S Cg: eight threads per entry, fp64 sums
          const int pp = p0 + (tq >> 3), s8 = tq & 7;
          double a = 0.0;
          if (pp < n2) {
            const int kk = pp / Lc, ll = pp - kk * Lc;
            const int ldc = L.cols.ldn;
            for (int j0 = s8; j0 < J; j0 += 64) {          // (eight of Pv's values on their way from L2 at a time)
              float zv[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) zv[u] = *G(L.ZT + (size_t)kk * ldc + min(j0 + 8 * u, J - 1));
#pragma unroll
              for (int u = 0; u < 8; ++u) if (j0 + 8 * u < J) a += (double)regC[(j0 + 8 * u) * kS + ll] * (double)zv[u];
            }
          }
          a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
          if (pp < n2 && s8 == 0) rS[pp] = (float)(a - (double)rS[pp]);
        }
        STAMP(18);
        if (!tri_dense) {
          // this thread's slots and q again (the sweep's registers are gone; q as the sweep stored it)
          // (the previous step's G values are gathered again for the deferred update of q, not kept: the registers go to q)
          uint32_t sj[EM / 2];
          float sq[EM];
#pragma unroll
          for (int h = 0; h < EM / 2; ++h) {
            uint32_t j0 = (uint32_t)(m * kS), j1 = (uint32_t)(m * kS);
            if (entry && 2 * h < em) { j0 = *G(d.idx + (2 * h) * kSmallThreads + tq); j1 = *G(d.idx + (2 * h + 1) * kSmallThreads + tq); }
            sj[h] = j0 | (j1 << 16);
            sq[2 * h] = sq[2 * h + 1] = 0.f;
            if (entry && 2 * h < em) { sq[2 * h] = *G(d.q + (2 * h) * kSmallThreads + tq); sq[2 * h + 1] = *G(d.q + (2 * h + 1) * kSmallThreads + tq); }
          }
          const int kkA = tq / Lc, llA = tq - kkA * Lc;        // the entry of r this thread keeps current
          STAMP(14);
          float dprev = 0.f;                       // delta of the step before x this thread's F value of that step's row
          int lprev = 0;
          u32x2n scw = {0u, 0u};
          const u32x2n* stab = reinterpret_cast<const u32x2n*>(L.stab);
          bar_all();                               // (rS, the candidate words)
          if (draw && tq < 4) scw = *G(stab + tq);
          float lam_n = tq < 64 ? *G(L.lambdaS) : 0.f;
#pragma unroll 1
          for (int k = 0; k < K; ++k) {
            const float f = entry ? regR[myunit * kS + k] : 0.f;
#pragma unroll 1
            for (int l = 0; l < Lc; ++l) {
              const int pp = k * Lc + l;
              {
                float qg = 0.f, gg = 0.f;
                const float* col = regC + l;
                const float* colp = regC + lprev;
                if (entry) {
#pragma unroll
                  for (int h = 0; h < EM / 2; ++h)
                    if (2 * h < em) {
                      asm volatile("" : "+v"(sj[h]));
                      const float g0 = col[sj[h] & 0xFFFFu], g1 = col[sj[h] >> 16];
                      sq[2 * h] = fmaf(dprev, colp[sj[h] & 0xFFFFu], sq[2 * h]);          // (dprev = 0 ahead of the first step)
                      sq[2 * h + 1] = fmaf(dprev, colp[sj[h] >> 16], sq[2 * h + 1]);
                      qg = fmaf(sq[2 * h], g0, qg); gg = fmaf(g0, g0, gg);
                      qg = fmaf(sq[2 * h + 1], g1, qg); gg = fmaf(g1, g1, gg);
                    }
                }
                // the wave's two sums by DPP (every wave: the ones without entries add zeros), one pair per wave to LDS
                const float wx = half_swap_sum(half_sum(f * qg)), wy = half_swap_sum(half_sum(f * f * gg));
                if ((tq & 63) == 0) part[tq >> 6] = float2{wx, wy};
              }
              // wave 0 draws: its candidates' word-only halves and the next step's words ahead of the barrier
              TnCand cand = {0.f, 0.f, 0.f};
              const float lamS = lam_n;
              if (tq < 64) {
                if (draw && tq < 4) cand = tn_cand_pre(scw.x, scw.y);
                if (pp + 1 < n2) {
                  if (draw && tq < 4) scw = *G(stab + (size_t)(pp + 1) * 4 + tq);
                  lam_n = *G(L.lambdaS + pp + 1);
                }
              }
              bar_lds();
              if (tq < 64) {
                // the (at most 16) waves' sums: one 16-lane row, added by DPP in an order that does not depend on the block size
                const float2 pt = tq < NT / 64 ? part[tq] : float2{0.f, 0.f};
                const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, dpp_xor_row_sum(pt.x))));
                const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, dpp_xor_row_sum(pt.y))));
                const float sold = Ss[k * kS + l];
                const float adiag = CfT[k * kS + k] * CgT[l * kS + l] - sy;
                const float tau_p = tau * adiag;
                const float numer = fmaf(tau, rS[pp] + sx + sold * adiag, -lamS);
                float xnew = 0.f;
                if (draw) {
                  const TnFast tf = tn_fast_params(numer, tau_p);
                  float xc = 0.f;
                  const bool acc = tq < 4 && tn_cand_post(tf, cand, &xc);
                  unsigned long long mask = __ballot(acc);
                  if (tf.live) {
                    if (mask) xnew = tn_guard(__shfl(xc, __ffsll((long long)mask) - 1, 64));
                    else
                      for (uint32_t c0 = 4u; c0 < 4096u; c0 += 64u) {        // (one draw in ~250 rejects its first four candidates)
                        const uint32_t c = c0 + (uint32_t)tq;
                        const U4 r = philox4x32_10(0u, (uint32_t)pp, it32, kStreamS + 16u * c, L.key0, L.key1);
                        const bool a2 = tn_eval_fast(tf, r.x, r.y, &xc) && c < 4096u;
                        mask = __ballot(a2);
                        if (mask) { xnew = tn_guard(__shfl(xc, __ffsll((long long)mask) - 1, 64)); break; }
                      }
                  }
                } else {
                  const float mu = numer / tau_p;
                  xnew = fmaxf((tau_p > 0.f && mu > 0.f) ? mu : 0.f, L.min_x);
                }
                if (tq == 0) { dl[0] = xnew - sold; Ss[k * kS + l] = xnew; SsT[l * kS + k] = xnew; }
              }
              bar_lds();
              const float delta = dl[0];
              if (tq < n2) rS[tq] = fmaf(-delta * CfT[kkA * kS + k], CgT[l * kS + llA], rS[tq]);
              for (int p2 = tq + NT; p2 < n2; p2 += NT) {          // (more entries than threads: the small blocks with wide factors)
                const int kk = p2 / Lc, ll = p2 - kk * Lc;
                rS[p2] = fmaf(-delta * CfT[kk * kS + k], CgT[l * kS + ll], rS[p2]);
              }
              dprev = delta * f; lprev = l;
            }
          }
          STAMP(15);
          // the last step's update of q; q goes where the G sweep finds it; S: state, sample, posterior sum
          if (entry) {
#pragma unroll
            for (int e = 0; e < EM; ++e)
              if (e < em) {
                const float vlast = regC[((sj[e >> 1] >> (16 * (e & 1))) & 0xFFFFu) + lprev];
                *G(d.q + e * kSmallThreads + tq) = fmaf(dprev, vlast, sq[e]);
              }
          }
        } else {
          // ---------------- the dense form: A = Cf (x) Cg - sum_miss (f f^T) (x) (g g^T) in LDS, the K L steps on ONE wave
          constexpr int KD = kTriDenseK, PD = kTriPairs;
          const int lane = tq & 63, wave = tq >> 6;
          const SmallDirDev& dc = L.cols;
          float* buf = gscr;                       // [8][NT]: eight packed second moments of every entry thread at a time
          // (the packed index is the 10-wide triangle's whatever K and L are: pairs behind the last one of the model's ranks are zero and skipped)
          const int pmaxK = tri_pair_index(K - 1, K - 1) + 1, pmaxL = tri_pair_index(Lc - 1, Lc - 1) + 1;
          // (1) W'_j = sum_{i in miss(j)} f_i f_i^T (packed upper triangle of the 10-wide index: columns >= K of F are zero) per
          // entry thread of the G sweep's layout, then per unit: the unit's threads in thread order
          {
            float acc[PD];
#pragma unroll
            for (int pi = 0; pi < PD; ++pi) acc[pi] = 0.f;
            const bool ce = tq < dc.nthreads;
            if (ce) {
              uint32_t off = *G(dc.idx + tq);
              for (int e = 0; e < dc.em; ++e) {
                const uint32_t off_n = e + 1 < dc.em ? (uint32_t)*G(dc.idx + (e + 1) * kSmallThreads + tq) : 0u;
                float fv[KD];
#pragma unroll
                for (int k2 = 0; k2 < KD; ++k2) fv[k2] = regR[off + k2];
                int pi = 0;
#pragma unroll
                for (int k2 = 0; k2 < KD; ++k2)
#pragma unroll
                  for (int k3 = k2; k3 < KD; ++k3) { acc[pi] = fmaf(fv[k2], fv[k3], acc[pi]); ++pi; }
                off = off_n;
              }
            }
            STAMP(19);
            uint16_t* segL = reinterpret_cast<uint16_t*>(tmpS);          // [J][2] the units' entry threads, read seven times: in LDS
            for (int t = tq; t < 2 * J; t += NT) segL[t] = *G(dc.seg + t);
#pragma unroll
            for (int ch = 0; ch < (PD + 7) / 8; ++ch) {
              if (ch * 8 >= pmaxK) break;
              bar_lds();
              if (ce) {
#pragma unroll
                for (int c = 0; c < 8; ++c) buf[c * NT + tq] = ch * 8 + c < PD ? acc[ch * 8 + c < PD ? ch * 8 + c : 0] : 0.f;
              }
              bar_lds();
              for (int o = tq; o < J * 8; o += NT) {
                const int j = o >> 3, c = o & 7;
                const int s0 = segL[2 * j], sn = segL[2 * j + 1];
                float sm = 0.f;
                for (int t = s0; t < s0 + sn; ++t) sm += buf[c * NT + t];
                *G(L.Wg + (size_t)j * 56 + ch * 8 + c) = sm;
              }
            }
          }
          bar_all();                               // (W' is read by other threads; F's region is free from here on)
          STAMP(20);
          float* Ad = small_tri_dense_overlays(I, K, Lc) ? regR : misc + MS::floats;        // [n2][n2] the system, then [64][64] the packed products
          float* Am = Ad + ((n2 * n2 + 3) & ~3);
          // (2) Am[(k <= k')][(l <= l')] = sum_j W'_j[(k k')] G_jl G_jl' on the f32 matrix cores: 4 x 4 tiles of 16 x 16, inner index j
          for (int tile = wave; tile < 16; tile += NT / 64) {
            const int ti = tile >> 2, tj = tile & 3;
            if (16 * ti >= pmaxK || 16 * tj >= pmaxL) continue;
            const int pk = 16 * ti + (lane & 15), pl = 16 * tj + (lane & 15), jq = lane >> 4;
            int la = 0, lb = 0;
            tri_pair(pl < PD ? pl : 0, la, lb);
            f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
            const int pkc = min(pk, 55);
            for (int j0 = 0; j0 < J; j0 += 16) {           // four inner steps' operands first (W' comes from L2), then their products
              float av[4], bv[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int j = j0 + 4 * u + jq, jc = min(j, J - 1);
                av[u] = *G(L.Wg + (size_t)jc * 56 + pkc);
                bv[u] = regC[jc * kS + la] * regC[jc * kS + lb];
                if (!(j < J && pk < 56)) av[u] = 0.f;
                if (!(j < J && pl < PD)) bv[u] = 0.f;
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc4, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Am[(16 * ti + 4 * jq + r) * 64 + 16 * tj + (lane & 15)] = acc4[r];      // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
          }
          bar_lds();
          STAMP(21);
          // (3) the system: A[(k l)][(k' l')] = Cf_kk' Cg_ll' - Am[(k k')][(l l')]; a thread keeps one column p' and walks the rows
          {
            const int pc = tq & 127, rg = tq >> 7;
            if (pc < n2) {
              const int k2 = pc / Lc, l2 = pc - k2 * Lc;
              constexpr int RS = NT / 128;
              const int dk = RS / Lc, dl2 = RS - dk * Lc;
              int k1 = rg / Lc, l1 = rg - k1 * Lc;
              for (int pr = rg; pr < n2; pr += RS, k1 += dk, l1 += dl2) {
                if (l1 >= Lc) { l1 -= Lc; ++k1; }
                const int pk = tri_pair_index(min(k1, k2), max(k1, k2)), pl = tri_pair_index(min(l1, l2), max(l1, l2));
                Ad[pr * n2 + pc] = fmaf(CfT[k1 * kS + k2], CgT[l1 * kS + l2], -Am[pk * 64 + pl]);
              }
            }
          }
          bar_lds();
          STAMP(22);
          // (4) r = b - A S
          for (int p0 = 0; p0 < n2; p0 += NT / 8) {          // eight threads per row, each every eighth column
            const int pr = p0 + (tq >> 3), s8 = tq & 7;
            float a = 0.f;
            if (pr < n2) {
              int k2 = s8 / Lc, l2 = s8 - k2 * Lc;
              const int dk = 8 / Lc, dl2 = 8 - dk * Lc;
              for (int pc = s8; pc < n2; pc += 8, k2 += dk, l2 += dl2) {
                if (l2 >= Lc) { l2 -= Lc; ++k2; }
                a = fmaf(Ad[pr * n2 + pc], Ss[k2 * kS + l2], a);
              }
            }
            a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
            if (pr < n2 && s8 == 0) rS[pr] -= a;
          }
          // the chain's first four candidates per entry (their halves that need the random words only) and the prior rates into LDS,
          // where the packed products were: a step of the chain is shorter than a round trip to L2
          float4* candL = reinterpret_cast<float4*>(Am);           // [n2][4]
          float* lamL = Am + 4 * 4 * 128;                           // [n2]
          for (int e = tq; e < n2 * 4; e += NT) {
            TnCand cd = {0.f, 0.f, 0.f};
            if (draw) {
              const U4 r = philox4x32_10(0u, (uint32_t)(e >> 2), it32, kStreamS + 16u * (uint32_t)(e & 3), L.key0, L.key1);
              cd = tn_cand_pre(r.x, r.y);
            }
            candL[e] = float4{cd.nl, cd.z, cd.sw, 0.f};
          }
          for (int e = tq; e < n2; e += NT) lamL[e] = *G(L.lambdaS + e);
          STAMP(14);
          bar_lds();
          // (5) the chain: lane p keeps r_p and S_p (and those of p + 64); a step reads its entry's row of A (one step ahead), forms
          // the conditional from registers, tests the four candidates in lanes 0-3, and folds delta into r.  What a lone wave pays
          // for is dependent instructions and, most of all, branches on vector conditions (tools/micro/lone_wave.hip): the parts
          // that need tau_p only are made a step ahead, S is selected by lane (no LDS round trip, no masked store), one branch per
          // step (the draw whose first four candidates were all rejected).
          if (tq < 64) {
            float r0 = lane < n2 ? rS[lane] : 0.f, r1 = lane + 64 < n2 ? rS[lane + 64] : 0.f;
            float s0 = 0.f, s1 = 0.f;
            { const int ka = lane / Lc, kb = (lane + 64) / Lc;
              if (lane < n2) s0 = Ss[ka * kS + lane - ka * Lc];
              if (lane + 64 < n2) s1 = Ss[kb * kS + lane + 64 - kb * Lc]; }
            float4 cn = candL[lane & 3];
            float lam_n = lamL[0];
            float a0 = lane < n2 ? Ad[lane] : 0.f, a1 = lane + 64 < n2 ? Ad[lane + 64] : 0.f, ad = Ad[0];
            TnPre pre_n = tn_fast_pre(tau * ad);
            auto chain = [&](auto draws) {             // (two copies of the loop: the update rule is not a branch of every step)
            constexpr bool DRAW = decltype(draws)::value;
#pragma unroll 1
            for (int sidx = 0; sidx < n2; ++sidx) {
              const TnCand cand = {cn.x, cn.y, cn.z};
              const float lamS = lam_n;
              const float c0 = a0, c1 = a1, adiag = ad;
              const TnPre pre = pre_n;
              if (sidx + 1 < n2) {                   // the next step's row, candidates, rate and tau_p parts on their way
                cn = candL[(sidx + 1) * 4 + (lane & 3)];
                lam_n = lamL[sidx + 1];
                const float* row = Ad + (sidx + 1) * n2;
                a0 = lane < n2 ? row[lane] : 0.f; a1 = lane + 64 < n2 ? row[lane + 64] : 0.f; ad = row[sidx + 1];
                pre_n = tn_fast_pre(tau * ad);
              }
              const int sl = sidx & 63;
              const float rs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sidx < 64 ? r0 : r1), sl));
              const float sold = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sidx < 64 ? s0 : s1), sl));
              const float tau_p = tau * adiag;
              const float numer = fmaf(tau, fmaf(sold, adiag, rs), -lamS);
              float xnew;
              if constexpr (DRAW) {
                const TnFast tf = tn_fast_post(pre, numer);
                float xc = 0.f;
                const bool acc = tn_cand_post(tf, cand, &xc);          // (every lane holds candidate lane & 3: no masked section)
                unsigned long long mask = __ballot(acc) & 0xFull;
                const int src = mask ? __ffsll((long long)mask) - 1 : 0;
                const float xsel = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tn_guard(xc)), src));
                xnew = (tf.live && mask) ? xsel : 0.f;
                if (__builtin_expect(tf.live && !mask, 0))
                  for (uint32_t cb = 4u; cb < 4096u; cb += 64u) {          // (one draw in ~250 rejects its first four candidates)
                    const uint32_t c = cb + (uint32_t)lane;
                    const U4 r = philox4x32_10(0u, (uint32_t)sidx, it32, kStreamS + 16u * c, L.key0, L.key1);
                    const bool a2 = tn_eval_fast(tf, r.x, r.y, &xc) && c < 4096u;
                    mask = __ballot(a2);
                    if (mask) { xnew = tn_guard(__shfl(xc, __ffsll((long long)mask) - 1, 64)); break; }
                  }
              } else {
                const float mu = numer / tau_p;
                xnew = fmaxf((tau_p > 0.f && mu > 0.f) ? mu : 0.f, L.min_x);
              }
              const float delta = xnew - sold;
              r0 = fmaf(-delta, c0, r0); r1 = fmaf(-delta, c1, r1);
              const bool mine = lane == sl;
              s0 = (mine && sidx < 64) ? xnew : s0;
              s1 = (mine && sidx >= 64) ? xnew : s1;
            }
            };
            if (draw) chain(std::true_type{}); else chain(std::false_type{});
            { const int ka = lane / Lc, kb = (lane + 64) / Lc;
              if (lane < n2) { Ss[ka * kS + lane - ka * Lc] = s0; SsT[(lane - ka * Lc) * kS + ka] = s0; }
              if (lane + 64 < n2) { Ss[kb * kS + lane + 64 - kb * Lc] = s1; SsT[(lane + 64 - kb * Lc) * kS + kb] = s1; } }
          }
          STAMP(15);
          bar_lds();
          if (small_tri_dense_overlays(I, K, Lc))        // F back into its region (and its zero row)
            for (int t = tq; t < (I + 1) * kS; t += NT) { const int u = t / kS, k2 = t - u * kS; regR[t] = (u < I && k2 < K) ? *G(L.rows.X + u * 32 + k2) : 0.f; }
        }
        {
          const bool add = L.expS && L.exp_burn >= 0 && it >= L.exp_burn && (it - L.exp_burn) % L.exp_thin == 0;
          for (int t = tq; t < n2; t += NT) {
            const int kk = t / Lc, ll = t - kk * Lc;
            const float v = Ss[kk * kS + ll];
            *G(L.S + t) = v;
            if (L.S_s) *G(L.S_s + (size_t)it * n2 + t) = v;
            if (add) *G(L.expS + t) += (double)v;
          }
        }
        bar_all();
        STAMP(17);
      }
    }

    // ---- end of the iteration (kernel_misc.hip finish_kernel): SSE from the Gram identities, tau, the three metrics
    {
      int tf_ = tid;
      asm volatile("" : "+v"(tf_));
      double v[4] = {st_dot, st_px, st_q, st_q2};
      block_sum4<NT>(v, red, tf_);
      if (tf_ == 0) {
        double sp1 = 0.0;
        for (int k = 0; k < Lc; ++k) sp1 += csum[k] * csum[32 + k];
        const double srp = v[1], sp = sp1 - v[2], spp = v[0] - v[3];
        const double nobs = L.n_obs;
        const double sse = L.sumR2 - 2.0 * srp + spp;
        const double alpha_s = L.alpha + 0.5 * nobs, beta_s = L.beta + 0.5 * sse;
        double tau;
        if (L.update == BNMTF_UPDATE_ICM) tau = (alpha_s - 1.0) / beta_s;
        else if (L.update != BNMTF_UPDATE_DRAW) tau = alpha_s / beta_s;
        else tau = *G(L.gunit + it) / beta_s;
        *G(L.tau_d) = tau; *G(L.tau_f) = (float)tau; *tau_s = (float)tau;
        const double ss_tot = L.sumR2 - L.sumR * L.sumR / nobs;
        const double cov = srp - L.sumR * sp / nobs;
        const double vpred = spp - sp * sp / nobs;
        auto* rec = G(L.rec + (size_t)it * 5);
        rec[0] = tau;
        rec[1] = sse / nobs;
        rec[2] = ss_tot != 0.0 ? 1.0 - sse / ss_tot : __longlong_as_double(0x7ff0000000000000LL);
        rec[3] = cov / (sqrt(ss_tot) * sqrt(vpred));
        rec[4] = sse;
        if (L.exp_tau && L.exp_burn >= 0 && it >= L.exp_burn && (it - L.exp_burn) % L.exp_thin == 0) *G(L.exp_tau) += tau;
        *G(L.clock + it + 1) = wall_clock64();
      }
      bar_lds();
      STAMP(13);
    }
  }
#ifdef BNMTF_SMALL_TIMING
  if (blockIdx.x == 0 && (tid == 0 || tid == NT - 64))
    printf("small kernel thread %d, %d iterations, cycles per iteration: table %llu contract %llu qinit %llu | columns: entry+prefetch %llu bar %llu unit %llu cand %llu pick %llu retry %llu (%d rounds) fixup %llu | qstore %llu gram %llu copy %llu finish %llu | tri: effective factor + Gram %llu S setup %llu S steps %llu S tail %llu\n",
           tid, L.n_iter, ph[0] / L.n_iter, ph[1] / L.n_iter, ph[2] / L.n_iter, ph[3] / L.n_iter, ph[4] / L.n_iter, ph[5] / L.n_iter, ph[6] / L.n_iter, ph[7] / L.n_iter,
           ph[8] / L.n_iter, nretry, ph[9] / L.n_iter, ph[10] / L.n_iter, ph[11] / L.n_iter, ph[12] / L.n_iter, ph[13] / L.n_iter,
           ph[16] / L.n_iter, (ph[14] + ph[18] + ph[19] + ph[20] + ph[21] + ph[22]) / L.n_iter, ph[15] / L.n_iter, ph[17] / L.n_iter);
  if (TRI && blockIdx.x == 0 && tid == 0)
    printf("   S setup: G back, R~ G, b %llu | packed second moments %llu per-unit sums %llu | products (MFMA) %llu system %llu r, candidates %llu\n",
           ph[18] / L.n_iter, ph[19] / L.n_iter, ph[20] / L.n_iter, ph[21] / L.n_iter, ph[22] / L.n_iter, ph[14] / L.n_iter);
  if (blockIdx.x == 0 && (tid & 63) == 0)
    printf("   wave %2d contraction per iteration: rows loop %llu epilogue %llu (%llu items) | cols loop %llu epilogue %llu (%llu items)\n", tid >> 6,
           cph[0] / L.n_iter, cph[1] / L.n_iter, cph[2] / L.n_iter, cph[3] / L.n_iter, cph[4] / L.n_iter, cph[5] / L.n_iter);
#endif
}

template <int EM, int NT, bool TRI>
static void launch_small_inst(const SmallLaunch* dev_launches, int n_models, size_t lds_bytes, hipStream_t st) {
  static std::atomic<uint64_t> ok{0};
  if (allow_full_lds((const void*)small_gibbs_kernel<EM, NT, TRI>, ok)) hipLaunchKernelGGL((small_gibbs_kernel<EM, NT, TRI>), dim3(n_models), dim3(NT), lds_bytes, st, dev_launches);
}
template <int EM, bool TRI>
static void launch_small_em(const SmallLaunch* dev_launches, int n_models, int nt, size_t lds_bytes, hipStream_t st) {
  if (nt <= 256) launch_small_inst<EM, 256, TRI>(dev_launches, n_models, lds_bytes, st);
  else if (nt <= 512) launch_small_inst<EM, 512, TRI>(dev_launches, n_models, lds_bytes, st);
  else launch_small_inst<EM, 1024, TRI>(dev_launches, n_models, lds_bytes, st);
}
// em: slots per entry thread (8 / 16 / 32), nt: threads per block (256 / 512 / 1024) -- the largest any model of the batch needs
template <bool TRI>
static void launch_small_kind(const SmallLaunch* dev_launches, int n_models, int em, int nt, size_t lds_bytes, hipStream_t st) {
  if (em <= 8) launch_small_em<8, TRI>(dev_launches, n_models, nt, lds_bytes, st);
  else if (em <= 16) launch_small_em<16, TRI>(dev_launches, n_models, nt, lds_bytes, st);
  else if (em <= 32) launch_small_em<32, TRI>(dev_launches, n_models, nt, lds_bytes, st);
  // (the classes above 32 slots only exist where the 32-slot one needs more than 1024 threads; they gather the previous column's
  // values again instead of keeping them: 40 and 48 slots still fit a 16-wave block's 128 registers per lane, 64 spill)
  else if (em <= 40) launch_small_inst<40, 1024, TRI>(dev_launches, n_models, lds_bytes, st);
  else if (em <= 48) launch_small_inst<48, 1024, TRI>(dev_launches, n_models, lds_bytes, st);
  else launch_small_inst<64, 1024, TRI>(dev_launches, n_models, lds_bytes, st);
}
// tri: every model of the batch is a tri-factorisation (SmallLaunch::L > 0); a batch is of one kind
void launch_small_gibbs(const SmallLaunch* dev_launches, int n_models, int em, int nt, size_t lds_bytes, hipStream_t st, bool tri) {
  if (n_models <= 0) return;
  if (tri) launch_small_kind<true>(dev_launches, n_models, em, nt, lds_bytes, st);
  else launch_small_kind<false>(dev_launches, n_models, em, nt, lds_bytes, st);
}

}  // namespace bnmtf
