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

#include "sweep_common.h"

namespace bnmtf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// mean and second moment of the effective factor:  Xe[r][c] = sum_t X[r][t] S(t,c),
// S2e[r][c] = sum_t (varX + X^2)[r][t] (varS + S^2)(t,c) - sum_t X^2[r][t] S^2(t,c) + Xe^2     (S(t,c) = S[t][c] or S[c][t])
__global__ __launch_bounds__(256) void small_product_vb_kernel(SmallProductVbArgs a) {
  __shared__ float Ss[32 * 32], Vs[32 * 32];
  for (int t = threadIdx.x; t < a.K * a.L; t += 256) { Ss[t] = a.S[t]; Vs[t] = a.varS[t]; }
  __syncthreads();
  const int inner = a.transposeS ? a.L : a.K, outw = a.transposeS ? a.K : a.L;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)a.rows * outw; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e / outw), c = (int)(e % outw);
    const float* x = a.X + (size_t)r * 32;
    const float* vx = a.varX + (size_t)r * 32;
    float m = 0.f, s2 = 0.f, sq = 0.f;
    for (int t = 0; t < inner; ++t) {
      const int si = a.transposeS ? c * a.L + t : t * a.L + c;
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
  constexpr int PF = 8;                                            // rows of A~ prefetched ahead of their step
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
      const float arow = pf[q];
      pf[q] = a.A[(uint32_t)ordl[i0 + q + PF] * (uint32_t)n2 + tcol];
      step(i0 + q, arow);
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
void launch_ssys_chain_vb(const SSysChainVbArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(ssys_chain_vb_kernel, dim3(1), dim3(1024), 0, st, a);
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
