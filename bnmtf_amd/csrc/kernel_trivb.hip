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
// muS = (-lambda + exptau (r_a + A~_aa E[S_a])) / tauS and the TN moments (fp64 routine), posts delta = E_new - E_old
// through LDS, and every thread folds delta A~[a][t] (a coalesced row: A~ is symmetric) into its residual.
// only_params: write mu/tau of the ordered entries, leave the moments alone (update_S without update_exp_S).
__global__ __launch_bounds__(1024) void ssys_chain_vb_kernel(SSysChainVbArgs a) {
  __shared__ float dl[2];
  const int n2 = a.K * a.L, t = threadIdx.x;
  const float tau = *a.tau;
  const bool mine = t < n2;
  float r = mine ? a.r0[t] : 0.f;
  float e = mine ? a.E[t] : 0.f;
  constexpr int PF = 4;                                            // rows of A~ prefetched ahead of their step
  float pf[PF];
#pragma unroll
  for (int q = 0; q < PF; ++q) pf[q] = (q < a.n_order && mine) ? a.A[(size_t)a.order[q] * n2 + t] : 0.f;
  for (int i0 = 0; i0 < a.n_order; i0 += PF) {
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int i = i0 + q;
      if (i >= a.n_order) break;
      const int ai = a.order[i];
      const float arow = pf[q];
      if (i + PF < a.n_order && mine) pf[q] = a.A[(size_t)a.order[i + PF] * n2 + t];
      if (t == ai) {
        const float aaa = arow;                                      // A~[a][a]
        const float tau_p = tau * aaa;
        const float numer = fmaf(tau, r + aaa * e, -a.lambdaS[t]);
        const float mu = numer / tau_p;
        a.mu[t] = mu; a.tauq[t] = tau_p;
        float enew = e;
        if (!a.only_params) {
          double ed, vd;
          tn_moments((double)mu, (double)tau_p, &ed, &vd);
          enew = (float)ed;
          a.var[t] = (float)vd;
          a.E[t] = enew;
        }
        dl[i & 1] = enew - e;
        e = enew;
      }
      __syncthreads();
      r = fmaf(-dl[i & 1], arow, r);
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

// update_tau + update_exp_tau (:231-233, 286-288) and the training-mask metrics from the three masked sums
__global__ void tri_vb_finish_kernel(const double* sums, double alpha, double beta, double* tau_d, float* tau_f, double* rec) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double n = sums[0], sr = sums[1], srr = sums[2], sp = sums[3], spp = sums[4], srp = sums[5];
  const double sse = srr - 2.0 * srp + spp;
  const double esd = sse + sums[8 + 3] + sums[16 + 3];
  const double alpha_s = alpha + 0.5 * n, beta_s = beta + 0.5 * esd;
  const double exptau = alpha_s / beta_s;
  *tau_d = exptau; *tau_f = (float)exptau;
  const double ss_tot = srr - sr * sr / n, cov = srp - sr * sp / n, vp = spp - sp * sp / n;
  rec[0] = exptau; rec[1] = sse / n;
  rec[2] = ss_tot != 0.0 ? 1.0 - sse / ss_tot : __longlong_as_double(0x7ff0000000000000LL);
  rec[3] = cov / (sqrt(ss_tot) * sqrt(vp));
  rec[4] = esd; rec[5] = beta_s;
}
void launch_tri_vb_finish(const double* sums, double alpha, double beta, double* tau_d, float* tau_f, double* rec, hipStream_t st) {
  hipLaunchKernelGGL(tri_vb_finish_kernel, dim3(1), dim3(64), 0, st, sums, alpha, beta, tau_d, tau_f, rec);
}

}  // namespace bnmtf
