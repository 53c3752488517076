// Variational tri-factorisation (bnmtf_vb_optimised.py:160-288): the pieces that are not already the BNMF-VB sweep, the
// effective-factor products or the dense S system.
//   * the F update of column k is the BNMF-VB update against the effective factor  V_jk = sum_l E[S_kl] E[G_jl]  with
//     second moment  S2_jk = sum_l E[S_kl^2] E[G_jl^2] - sum_l E[S_kl]^2 E[G_jl]^2 + V_jk^2  (:242-243), MINUS the
//     covariance term (:246)   cov_ik = sum_l E[S_kl] mvG_il (FS_il - E[F_ik] E[S_kl]),
//     mvG_il = sum_j M_ij varG_jl,  FS_il = sum_k' E[F_ik'] E[S_k'l]  -- per unit an L-vector, kept current as the unit's
//     columns change (kernel_sweep.hip takes it as SweepArgs::cov_*).  G likewise with the roles swapped (:265-273).
//   * the S entries are coordinate steps on the dense system of kernel_ssys.hip built from second moments, walked in
//     the (shuffled) order the host hands over: ssys_chain_vb_kernel.
//   * exp_square_diff (:235-239) = four masked bilinear sums, evaluated directly in fp64 by metric_kernel on factor
//     matrices tri_factors_kernel lays out.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "sweep_common.h"
#include "tn_mean_table.h"

namespace bnmtf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// mean and second moment of the effective factor:  Xe[r][c] = sum_t X[r][t] S(t,c),
// S2e[r][c] = sum_t (varX + X^2)[r][t] (varS + S^2)(t,c) - sum_t X^2[r][t] S^2(t,c) + Xe^2     (S(t,c) = S[t][c] or S[c][t])
__global__ __launch_bounds__(256) void small_product_vb_kernel(SmallProductVbArgs a) {
  // S(t, c) at [t][c] in LDS whichever way it is read: consecutive lanes (c) on consecutive banks (round 6: the transposed read
  // c L + t was a 32-way bank conflict -- 16.8 us for the F side against 5 us for the G side)
  __shared__ float Ss[32 * 32], Vs[32 * 32];
  for (int t = threadIdx.x; t < a.K * a.L; t += 256) {
    const int k = t / a.L, l = t % a.L;
    const int at = a.transposeS ? l * 32 + k : k * 32 + l;
    Ss[at] = a.S[t]; Vs[at] = a.varS[t];
  }
  __syncthreads();
  const int inner = a.transposeS ? a.L : a.K, outw = a.transposeS ? a.K : a.L;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)a.rows * outw; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e / outw), c = (int)(e % outw);
    const float* x = a.X + (size_t)r * 32;
    const float* vx = a.varX + (size_t)r * 32;
    float m = 0.f, s2 = 0.f, sq = 0.f;
    for (int t = 0; t < inner; ++t) {
      const int si = t * 32 + c;
      const float xs = x[t], ss = Ss[si];
      m = fmaf(xs, ss, m);
      s2 = fmaf(vx[t] + xs * xs, Vs[si] + ss * ss, s2);
      sq = fmaf(xs * xs, ss * ss, sq);
    }
    a.out[(size_t)r * 32 + c] = m;
    a.outS2[(size_t)r * 32 + c] = (s2 - sq) + m * m;
  }
}
void launch_small_product_vb(const SmallProductVbArgs& a, hipStream_t st) {
  const int outw = a.transposeS ? a.K : a.L;
  const int blocks = (int)std::min<size_t>(2048, ((size_t)a.rows * outw + 255) / 256);
  hipLaunchKernelGGL(small_product_vb_kernel, dim3(blocks), dim3(256), 0, st, a);
}

// mv[u][c] = sum over the unit's OBSERVED inner indices of V[.][c] = colsum_c - sum_{e in miss(u)} V[idx_e][c]:
// one wave per unit, a 32-lane half takes one missing entry per trip (a coalesced 128-byte row of V), eight in flight
__global__ __launch_bounds__(256) void masked_colsum_kernel(MaskedColsumArgs a) {
  const int lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31;
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= a.n) return;
  const uint32_t s0 = a.slot_ptr[u], s1 = a.slot_ptr[u + 1];
  float acc = 0.f;
  for (uint32_t e0 = s0; e0 < s1; e0 += 16) {
    uint32_t ii[8]; float v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) ii[t] = a.idx[e0 + 2u * t + half];       // padding slots point at the zero row
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = a.V[(size_t)ii[t] * 32 + c];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc += v[t];
  }
  acc += __shfl_xor(acc, 32, 64);
  if (half == 0) a.out[(size_t)u * 32 + c] = (float)(a.colsum2[c] - a.C64[(size_t)c * 32 + c]) - acc;   // sum var = sum S2 - sum E^2
}
void launch_masked_colsum(const MaskedColsumArgs& a, hipStream_t st) {
  if (a.n > 0) hipLaunchKernelGGL(masked_colsum_kernel, dim3((a.n + 3) / 4), dim3(256), 0, st, a);
}


// update_S(k,l) + update_exp_S(k,l) for the entries order[0 .. n_order) in that order (bnmtf_vb_optimised.py:172-176):
// one block, thread t owns entry t of the residual r = b - A~ E[S]; the owner of a step forms tauS = exptau A~_aa,
// muS = (-lambda + exptau (r_a + A~_aa E[S_a])) / tauS and the TN mean, posts delta = E_new - E_old
// through LDS, and every thread folds delta A~[a][t] (a coalesced row: A~ is symmetric) into its residual.
// A step costs what its owner's lane issues between two barriers (a lone wave: ~8 cycles an instruction), so everything that does
// not depend on the chain is taken off it (round 6: 0.93 -> ~0.3 us a step): tauS, its reciprocal and lambda are per-thread
// constants formed before the first step; the mean is the sweeps' fp32 routine (device_rng.h: 2.5e-7 against 40-digit values; the
// fp64 routine was ~1 500 cycles of dependent fp64 arithmetic per step); the variance -- which feeds nothing in the chain -- is
// evaluated behind the last step by every thread for its own entry, from the (mu, tau) the step stored; the barrier of a step
// waits for LDS traffic only, so the rows of A~ prefetched for later steps stay in flight.
// only_params: write mu/tau of the ordered entries, leave the moments alone (update_S without update_exp_S).
__global__ __launch_bounds__(1024) void ssys_chain_vb_kernel(SSysChainVbArgs a) {
  constexpr int PF = 12;                                           // rows of A~ prefetched ahead of their step
  __shared__ float dl[2];
  __shared__ int ordl[1024 + PF], posl[1024];
  const int n2 = a.K * a.L, t = threadIdx.x, n_order = a.n_order;
  const float tau = *a.tau;
  const bool mine = t < n2;
  posl[t] = -1;
  __syncthreads();
  for (int i = t; i < n_order + PF; i += 1024) {
    const int ai = i < n_order ? a.order[i] : 0;
    ordl[i] = ai;
    if (i < n_order) posl[ai] = i;
  }
  __syncthreads();
  const int mypos = posl[t];                                       // the step that updates this thread's entry (-1: none)
  float r = mine ? a.r0[t] : 0.f;
  float e = mine ? a.E[t] : 0.f;
  const float aaa = mine ? a.A[(size_t)t * n2 + t] : 1.f;       // A~[t][t]
  const float tau_p = tau * aaa, inv_tp = 1.0f / tau_p;
  const float sig = __builtin_amdgcn_rsqf(tau_p), xs = tau_p * sig;   // (as tn_moments_f32 forms them)
  const float lam = mine ? a.lambdaS[t] : 0.f;
  float mu_t = 0.f;
  // (unconditional loads -- ordl is padded with PF zeros, a thread beyond the system reads its last column: a predicated load
  // is merged into its register behind a vmcnt(0), i.e. every step waited for a trip to L2: 0.58 us a step, round 6)
  const uint32_t tcol = (uint32_t)(mine ? t : n2 - 1);
  float pf[PF];
#pragma unroll
  for (int q = 0; q < PF; ++q) pf[q] = a.A[(uint32_t)ordl[q] * (uint32_t)n2 + tcol];
  auto step = [&](int i, float arow) {
    if (i == mypos) {
      const float numer = fmaf(tau, fmaf(aaa, e, r), -lam);
      mu_t = numer * inv_tp;
      float enew = e;
      if (!a.only_params) {
        // x = -mu sqrt(tau) <= -6: the routine's lambda term is below half an ulp of |x| and its result is sig |x|, bit for bit
        // -- the common case of an entry of S away from zero (its mean many standard deviations above it)
        const float x = -mu_t * xs;
        if (x <= -6.0f) {
          const float ef = sig * (-x);
          enew = isfinite(ef) ? ef : 0.f;
        } else {
          float vf;
          tn_moments_f32(mu_t, tau_p, &enew, &vf);                  // (the variance's part of the routine is dead code here)
        }
      }
      dl[i & 1] = enew - e;
      e = enew;
    }
    // (LDS traffic only: __syncthreads would also wait for the row loads in flight -- a trip to L2 on every step)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    r = fmaf(-dl[i & 1], arow, r);
  };
  // whole groups of PF steps: straight-line code, every register of the ring reloaded right behind its use; then the rest
  int i0 = 0;
  for (; i0 + PF <= n_order; i0 += PF) {
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      step(i0 + q, pf[q]);
      pf[q] = a.A[(uint32_t)ordl[i0 + q + PF] * (uint32_t)n2 + tcol];      // (behind its use: the reload lands in the same register, no copies -- and no drain -- at the loop's end)
    }
  }
#pragma unroll
  for (int q = 0; q < PF; ++q)
    if (i0 + q < n_order) step(i0 + q, pf[q]);
  if (mypos >= 0) {
    a.mu[t] = mu_t; a.tauq[t] = tau_p;
    if (!a.only_params) {
      float ef, vf;
      tn_moments_f32(mu_t, tau_p, &ef, &vf);
      a.E[t] = e; a.var[t] = vf;
    }
  }
}

// The same chain for a whole pass (every step with its moments), walked in BLOCKS of 64 steps (round 6).  The kernel above pays a
// block barrier and two LDS round trips per step (0.28 us a step, 285 us for 1 024 entries at K = L = 32: 40 % of an iteration
// once the sweeps and exp_square_diff were off the generic kernels).  Here thread p owns STEP p of the order (entry order[p]), so
// 64 consecutive steps live in the lanes of ONE wave and a step's delta travels by v_readlane:
//   * in its block, a wave runs   nm = tau r + c;  d = nm g - E_old;  delta = readlane(d, i);  r -= delta Adiag[i][lane]
//     for i = 0 .. 63 -- every lane evaluates its own candidate each step, lane i's is the step's.  c = tau A~_aa E_old - lambda,
//     g and the regime test's factor are per-lane constants.  Adiag[i][.] = A~[order[i]][order[lane]] comes from LDS with its
//     lower triangle and diagonal ZEROED, so a lane's residual freezes at its own step and mu, E of the step are re-formed from it
//     behind the block (no per-step capture);
//   * the common regime -x = mu sqrt(tau) >= 6 (an entry many standard deviations above zero: the fp32 routine returns
//     sig |x|, its lambda term below half an ulp) is the two FMAs above; a step whose lane fails the test leaves through a
//     wave-uniform branch into the routine (device_rng.h tn_moments_f32);
//   * behind a block's barrier the waves of LATER blocks fold its 64 deltas into their residuals (64 FMAs against rows they
//     loaded while the block ran); the next block's wave is the only one anybody waits for.  A wave stages its own diagonal
//     block two blocks ahead (three LDS regions).
// ~12 instructions a step instead of ~35 + a barrier: measured 285 -> see DESIGN.md 4.3.
// A~ in the order of a pass: out[s][p] = A~[order[s]][order[p]].  The blocked chain reads rows of this for whole waves (a lane = a
// step): out of A~ itself those were 64-lane gathers inside 4 KiB rows -- ~28 cache lines a load instruction, 27 000 line requests
// per block of the chain from ONE compute unit, which is what its hand-overs waited for (round 6: 10 000 cycles each) -- here all
// 256 units share the gathers once and the chain's loads are contiguous.
__global__ __launch_bounds__(256) void ssys_permute_kernel(const float* __restrict__ A, const int* __restrict__ order, int n, float* __restrict__ out) {
  const int s = blockIdx.x;
  const float* row = A + (size_t)order[s] * n;
  for (int p = threadIdx.x; p < n; p += 256) out[(size_t)s * n + p] = row[order[p]];
}
void launch_ssys_permute(const float* A, const int* order, int n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(ssys_permute_kernel, dim3(n), dim3(256), 0, st, A, order, n, out);
}

__global__ __launch_bounds__(1024) void ssys_chain_vb_blocked_kernel(SSysChainVbArgs a) {
  __shared__ __align__(16) float dlt[1024];
  __shared__ float diag[3 * 64 * 64 + 8 * 64];
  const int n2 = a.K * a.L, p = threadIdx.x, lane = p & 63, n_order = a.n_order;      // (n_order == n2: a whole pass)
  const int w = __builtin_amdgcn_readfirstlane(p >> 6);
  const int nblk = (n_order + 63) / 64;
  const bool on = p < n_order;
  dlt[p] = 0.f;
  const uint32_t ent = (uint32_t)a.order[on ? p : n_order - 1];
  const float tau = *a.tau;
  float r = a.r0[ent];
  const float e_old = a.E[ent];
  const float aaa = a.A[(size_t)ent * n2 + ent];
  const float tau_p = tau * aaa, inv_tp = 1.0f / tau_p;
  const float sig = __builtin_amdgcn_rsqf(tau_p), xs = tau_p * sig;      // (as tn_moments_f32 forms them)
  const float c = fmaf(tau * aaa, e_old, -a.lambdaS[ent]);                // numer = tau (r + A~_aa E) - lambda = tau r + c
  const float hh = inv_tp * xs;                                           // -x = numer hh
  const float g = hh * sig;                                               // E_new = sig (-x) = numer g
  const float thr = hh > 0.f ? 6.0f / hh : __builtin_inff();              // -x >= 6  <=>  numer >= thr
  const float q1 = -hh * kTnMeanInvW;                                     // (x - x0) / width = numer q1 - x0 / width
  const float tc0 = kTnMeanTab[lane][0], tc1 = kTnMeanTab[lane][1], tc2 = kTnMeanTab[lane][2],
              tc3 = kTnMeanTab[lane][3], tc4 = kTnMeanTab[lane][4], tc5 = kTnMeanTab[lane][5];      // segment `lane` of the mean's table
  auto rl = [](float v, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k)); };
  __syncthreads();
  float ab[64];
  // rows of A~ of block b's steps, this thread's column
  // rows of the permuted system (a.Aperm): descriptor + the row's byte offset in an SGPR + the lane's column in ONE VGPR
  const __amdgpu_buffer_rsrc_t rsA = panel_rsrc(a.Aperm, (size_t)n2 * n2 * 4);
  const int col_b = p * 4;
  auto load_rows = [&](int b) {
#pragma unroll
    for (int j = 0; j < 64; ++j)                   // (rows past the last step: the descriptor returns 0, the delta is 0)
      ab[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, col_b, (b * 64 + j) * n2 * 4, 0));
  };
  auto stage_diag = [&]() {                        // this wave's own block, strictly upper triangle
    load_rows(w);
    const int reg = (w % 3) * 4096 + lane;         // (indexed off the LDS array itself: a float* picked out of three turns into 64-bit flat arithmetic)
    int ln = on ? lane : -1;
    asm volatile("" : "+v"(ln));                   // opaque: the 64 lane masks are made here, not hoisted out of the block loop into 128 scalar registers
#pragma unroll
    for (int j = 0; j < 64; ++j) diag[reg + j * 64] = ln > j ? ab[j] : 0.f;
  };
  if (w < 3 && w < nblk) stage_diag();
  // (three phases per wave, not one loop over the blocks with the roles inside: there the 64 row registers stayed live through
  // the active block's code -- 135 spilled)
#ifdef CHAIN_CLOCK
  __shared__ unsigned long long cks[16][4];
  const unsigned long long ck0 = __builtin_amdgcn_s_memtime();
  unsigned long long ck_fold = 0;
  int n_slow = 0, n_far = 0;
#endif
  for (int blk = 0; blk < w; ++blk) {              // ---- the blocks ahead of mine: their rows, their deltas into my residual
    load_rows(blk);                                // on their way while the block runs
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (LDS traffic only: the rows stay in flight)
#ifdef CHAIN_CLOCK
    const unsigned long long ckf = __builtin_amdgcn_s_memtime();
#endif
    int dbase = 64 * blk;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float4 d4 = *reinterpret_cast<const float4*>(&dlt[dbase + 4 * j]);
      r = fmaf(-d4.x, ab[4 * j], r); r = fmaf(-d4.y, ab[4 * j + 1], r);
      r = fmaf(-d4.z, ab[4 * j + 2], r); r = fmaf(-d4.w, ab[4 * j + 3], r);
      // (16 deltas in flight at a time -- the next group's address "depends" on this group's sum: all 64 read ahead are 64 more live registers)
      if ((j & 3) == 3) asm volatile("" : "+v"(dbase), "+v"(r));
    }
#ifdef CHAIN_CLOCK
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (blk == w - 1) ck_fold = __builtin_amdgcn_s_memtime() - ckf;
#endif
    __builtin_amdgcn_sched_barrier(0);             // (the fold's FMAs are not interleaved with the next 64 loads: that is 128 live registers)
    if (w == blk + 2 && w >= 3) stage_diag();      // its region: free since the barrier behind block w - 3
  }
  {                                                // ---- my block
    const int blk = w;
    // (rows read four steps ahead through ONE running index and immediate offsets; 8 spare rows behind the regions keep the last reads inside the array)
    int reg = (w % 3) * 4096 + lane;
    const int cnt = min(64, n_order - 64 * blk);
    float a0 = diag[reg], a1 = diag[reg + 64], a2 = diag[reg + 128], a3 = diag[reg + 192];
    auto one = [&](int i, float arow) {
      const float nm = fmaf(tau, r, c);
      float d = fmaf(nm, g, -e_old);
      const unsigned long long ok = __ballot(nm >= thr);                  // (a NaN fails the test: the routine's guards see it)
      if (__builtin_expect(!((ok >> i) & 1ull), 0)) {
        // -6 < x < 26: the segment table (tn_mean_table.h) -- lane i's segment number into a scalar register, that segment's six
        // coefficients out of their lanes by v_readlane, Horner in t = frac: ~20 instructions where the routine is ~50
        const float xq = fmaxf(fmaf(nm, q1, -kTnMeanX0 * kTnMeanInvW), -1.0f);      // ((x - x0) / width; a NaN becomes -1)
        const int kseg = __builtin_amdgcn_readlane((int)__builtin_floorf(xq), i);
        float en;
        if (__builtin_expect((unsigned)kseg < (unsigned)kTnMeanSeg, 1)) {
          const float t = __builtin_amdgcn_fractf(xq);
          float pz = rl(tc5, kseg);
          pz = fmaf(pz, t, rl(tc4, kseg)); pz = fmaf(pz, t, rl(tc3, kseg)); pz = fmaf(pz, t, rl(tc2, kseg));
          pz = fmaf(pz, t, rl(tc1, kseg)); pz = fmaf(pz, t, rl(tc0, kseg));
          en = sig * pz;
        } else {
          // x >= 26 (an entry pinned at zero) or not a number: the routine, in the regime of lane i's x (its sign is the numerator's)
          const unsigned long long xneg = __ballot(nm > 0.0f);
          const float mu = nm * inv_tp;
          en = ((xneg >> i) & 1ull) ? tn_mean_f32_regime<false>(mu, tau_p) : tn_mean_f32_regime<true>(mu, tau_p);
        }
        d = en - e_old;
#ifdef CHAIN_CLOCK
        ++n_slow;
#endif
      }
      const float delta = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), i));
      r = fmaf(-delta, arow, r);
    };
#ifdef CHAIN_CLOCK
    const unsigned long long ck1 = __builtin_amdgcn_s_memtime();
#endif
    int i = 0;
    for (; i + 4 <= cnt; i += 4) {
      one(i, a0);     a0 = diag[reg + 256];
      one(i + 1, a1); a1 = diag[reg + 320];
      one(i + 2, a2); a2 = diag[reg + 384];
      one(i + 3, a3); a3 = diag[reg + 448];
      reg += 256;
    }
    if (i < cnt) one(i, a0);
    if (i + 1 < cnt) one(i + 1, a1);
    if (i + 2 < cnt) one(i + 2, a2);
#ifdef CHAIN_CLOCK
    const unsigned long long ck2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { cks[w][0] = ck1; cks[w][1] = ck_fold; cks[w][2] = ck2; cks[w][3] = (unsigned long long)n_slow | ((unsigned long long)n_far << 32); }
#endif
    // the lane's residual froze at its own step: its mu, E and delta once more, for the other waves and the results
    // (a lane whose step took the table or the routine: the routine's mean here -- the table's agrees with it to 3e-7)
    const float nm = fmaf(tau, r, c);
    const float mu = nm * inv_tp;
    float ef, vf;
    tn_moments_f32(mu, tau_p, &ef, &vf);
    const bool fast = nm >= thr;
    const float en = fast ? nm * g : ef;
    dlt[p] = on ? (fast ? fmaf(nm, g, -e_old) : ef - e_old) : 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (on) { a.mu[ent] = mu; a.tauq[ent] = tau_p; a.E[ent] = en; a.var[ent] = vf; }      // (behind the barrier: nobody waits for these)
  }
  for (int blk = w + 1; blk < nblk; ++blk) asm volatile("s_barrier" ::: "memory");      // ---- the blocks behind mine
#ifdef CHAIN_CLOCK
  if (w == nblk - 1 && lane == 0) {
    const unsigned long long ck3 = __builtin_amdgcn_s_memtime();
    unsigned long long st = 0, fo = 0, gaps = 0; int ns = 0;
    for (int b = 0; b < nblk; ++b) { st += cks[b][2] - cks[b][0]; fo += cks[b][1]; ns += (int)(cks[b][3] & 0xffffffffu); if (b) gaps += cks[b][0] - cks[b - 1][2]; }
    printf("chain: %llu cycles from the prologue's end to the last wave's end; steps %llu, critical folds %llu, hand-overs (end of a block's steps -> start of the next one's) %llu, slow steps %d of %d; first block started %llu in, gap 0->1 %llu, 7->8 %llu\n",
           ck3 - ck0, st, fo, gaps, ns, n_order, cks[0][0] - ck0, cks[1][0] - cks[0][2], cks[8][0] - cks[7][2]);
  }
#endif
}
void launch_ssys_chain_vb(const SSysChainVbArgs& a, hipStream_t st) {
  // BNMTF_VB_CHAIN=steps: the barrier-per-step kernel for whole passes too (A/B switch)
  static const bool steps = [] { const char* e = getenv("BNMTF_VB_CHAIN"); return e && !strcmp(e, "steps"); }();
  if (a.only_params || a.n_order < 64 || a.n_order > 1024 || a.n_order != a.K * a.L || !a.Aperm || steps) { hipLaunchKernelGGL(ssys_chain_vb_kernel, dim3(1), dim3(1024), 0, st, a); return; }
  hipLaunchKernelGGL(ssys_chain_vb_blocked_kernel, dim3(1), dim3(((a.n_order + 63) / 64) * 64), 0, st, a);
}

// fp64 factor matrices of the masked bilinear sums of exp_square_diff (:235-239), for metric_kernel (sum over the mask of
// A_i . B_j):   which = 0:  A = E[F] E[S]                        (I x L),      B = E[G]                 -> SSE and the metrics
//               which = 1:  A = [E2F E2S | -E[F]^2 E[S]^2]       (I x 2L),     B = [E2G | E[G]^2]       -> second term
//               which = 2:  A = [varF | (E[F]E[S])^2 - E[F]^2E[S]^2]  (I x (K+L)),  B = [(E[S]E[G]^T)^2 - E[S]^2 (E[G]^2)^T | varG]   -> third + fourth
// (E2X = varX + E[X]^2).  side = 0 writes A (rows = I), side = 1 writes B (rows = J).
__global__ __launch_bounds__(256) void tri_factors_kernel(TriFactorArgs a) {
  __shared__ float Ss[32 * 32], Vs[32 * 32];
  for (int t = threadIdx.x; t < a.K * a.L; t += 256) { Ss[t] = a.S[t]; Vs[t] = a.varS[t]; }
  __syncthreads();
  const int K = a.K, L = a.L;
  const int width = a.which == 0 ? L : (a.which == 1 ? 2 * L : K + L);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)a.rows * width; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e / width), c = (int)(e % width);
    const float* x = a.X + (size_t)r * 32;
    const float* vx = a.varX + (size_t)r * 32;
    double out = 0.0;
    if (a.side == 0) {                                              // rows of F
      if (a.which == 0) { for (int k = 0; k < K; ++k) out += (double)x[k] * (double)Ss[k * L + c]; }
      else if (a.which == 1) {
        const int l = c % L;
        if (c < L) { for (int k = 0; k < K; ++k) { const double xs = x[k], ss = Ss[k * L + l]; out += ((double)vx[k] + xs * xs) * ((double)Vs[k * L + l] + ss * ss); } }
        else       { for (int k = 0; k < K; ++k) { const double xs = x[k], ss = Ss[k * L + l]; out -= xs * xs * ss * ss; } }
      } else {
        if (c < K) out = (double)vx[c];
        else {
          const int l = c - K;
          double m = 0.0, sq = 0.0;
          for (int k = 0; k < K; ++k) { const double xs = x[k], ss = Ss[k * L + l]; m += xs * ss; sq += xs * xs * ss * ss; }
          out = m * m - sq;
        }
      }
    } else {                                                        // rows of G
      if (a.which == 0) out = (double)x[c];
      else if (a.which == 1) { const int l = c % L; const double xs = x[l]; out = c < L ? (double)vx[l] + xs * xs : xs * xs; }
      else {
        if (c < K) {
          double m = 0.0, sq = 0.0;
          for (int l = 0; l < L; ++l) { const double xs = x[l], ss = Ss[c * L + l]; m += xs * ss; sq += xs * xs * ss * ss; }
          out = m * m - sq;
        } else out = (double)vx[c - K];
      }
    }
    a.out[(size_t)r * width + c] = out;
  }
}
void launch_tri_factors(const TriFactorArgs& a, hipStream_t st) {
  const int width = a.which == 0 ? a.L : (a.which == 1 ? 2 * a.L : a.K + a.L);
  const int blocks = (int)std::min<size_t>(2048, ((size_t)a.rows * width + 255) / 256);
  hipLaunchKernelGGL(tri_factors_kernel, dim3(blocks), dim3(256), 0, st, a);
}

// exp_square_diff's third term, sum_Omega varF . ((E[S] E[G]^T)^2 - E[S]^2 (E[G]^2)^T) (:238), as a sum over (j, k): the masked
// sums mv[j][k] = sum_{i in Omega_j} varF_ik are what the G step's covariance term has just used (masked_colsum_kernel), the
// bracket is a function of row j of E[G] and row k of E[S].  One thread per (j, k), fp64 partial sum per block.
__global__ __launch_bounds__(256) void tri_third_kernel(TriThirdArgs a) {
  __shared__ float Ss[32 * 32];
  __shared__ double red[4];
  for (int t = threadIdx.x; t < a.K * a.L; t += 256) Ss[t] = a.S[t];
  __syncthreads();
  const int j = blockIdx.x * 8 + (threadIdx.x >> 5), k = threadIdx.x & 31;
  double v = 0.0;
  if (j < a.rows && k < a.K) {
    const float* g = a.G + (size_t)j * 32;
    float m = 0.f, sq = 0.f;
    for (int l = 0; l < a.L; ++l) { const float t = g[l] * Ss[k * a.L + l]; m += t; sq = fmaf(t, t, sq); }
    v = (double)a.mv[(size_t)j * 32 + k] * ((double)m * (double)m - (double)sq);
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) a.part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
int tri_third_blocks(int rows) { return (rows + 7) / 8; }
void launch_tri_third(const TriThirdArgs& a, hipStream_t st) {
  if (a.rows > 0) hipLaunchKernelGGL(tri_third_kernel, dim3(tri_third_blocks(a.rows)), dim3(256), 0, st, a);
}

}  // namespace bnmtf
