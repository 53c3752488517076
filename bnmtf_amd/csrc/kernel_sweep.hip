// K3: one half sweep = the K sequential conditional updates of every row of one
// factor (bnmf_gibbs_optimised.py:134-142 and the tauU/muU algebra of :167-177),
// restated in the exact Gram + sparse-complement form ("form G"):
//
//   a_ik   = sum_j M_ij V_jk^2                = C_kk - sum_{j in miss(i)} V_jk^2
//   num_ik = sum_j M_ij (R_ij - sum_{l!=k} U_il V_jl) V_jk
//          = P_ik - sum_{l!=k} U_il C_lk + sum_{j in miss(i)} (q_ij - U_ik V_jk) V_jk
//   tauU_ik = tau a_ik ,  muU_ik = (-lambda_ik + tau num_ik) / tauU_ik
//
// with P = R~.V (kernel_gemm), C = V^T V (gram) and q_ij = U_i.V_j kept only on the
// missing entries and updated in place after each draw (q += delta * V_jk), so column
// k sees the new columns < k exactly as the reference's in-place loop does.
//
// Generic version: one 64-lane wave per row; the row's missing entries live in
// 64-wide slots (slot e*64+lane), q in global memory.  Correct for any mask; the
// register/LDS-resident fast path (sweep_chip.inc) handles the dense-mask
// shapes of the benchmark configs.
#include "kernels.h"
#include "device_rng.h"

namespace bnmtf {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void sweep_generic_kernel(SweepArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ui = blockIdx.x * 4 + wave;
  if (ui >= a.n) return;
  const int u = a.unit_list ? a.unit_list[ui] : ui;
  const int KP = a.KP, K = a.K;
  const size_t gi = (size_t)a.n0 + u;

  float x = 0.f, p = 0.f, lam = 0.f;
  if (lane < KP) {
    x = a.Xself[gi * KP + lane];
    for (int s = 0; s < a.split; ++s) p += a.slabs[((size_t)s * a.n_pad + u) * KP + lane];
    lam = a.lambda[(size_t)u * KP + lane];
  }
  const float tau = *a.tau;
  const uint32_t s0 = a.slot_ptr[u], s1 = a.slot_ptr[u + 1];
  const uint32_t* idx = a.idx + s0 + lane;
  float* q = a.q + s0 + lane;
  const int E = (int)((s1 - s0) >> 6);

  // q_ij = U_i . V_j on the missing entries (sentinel slots gather zeros)
  for (int e = 0; e < E; ++e) {
    const uint32_t j = idx[e * 64];
    float qv = 0.f;
    for (int k = 0; k < K; ++k) qv = fmaf(__shfl(x, k, 64), a.XoT[(size_t)k * a.ldT_o + j], qv);
    q[e * 64] = qv;
  }
  if (a.qinit_only) return;

  float mu_l = 0.f, tau_l = 0.f, var_l = 0.f;     // VB: per-lane (lane == k) outputs
  float var_in = 0.f;
  if (MODE == kSweepVB && lane < KP) { mu_l = a.mu_self[gi * KP + lane]; tau_l = a.tau_self[gi * KP + lane]; var_l = var_in = a.var_self[gi * KP + lane]; }
  const int only = a.cond_k >= 0 ? a.cond_k : a.only_k;
  const int kbeg = only >= 0 ? only : 0, kend = only >= 0 ? only + 1 : K;
  double e_quad = 0.0, e_lerfc = 0.0, e_ltau = 0.0, e_lamx = 0.0, e_q2 = 0.0, e_q3 = 0.0;   // VB: ELBO / exp_square_diff pieces
  // tri-factorisation VB: fs_t = sum_c x_c S(c,t) (lane = inner index t) and the masked variance sums of the other factor
  float fs = 0.f, mvl = 0.f;
  const bool cov_on = MODE == kSweepVB && a.cov_S != nullptr, cov_lane = cov_on && lane < a.cov_n;
  if (cov_on) {
    for (int c = 0; c < K; ++c) fs = fmaf(__shfl(x, c, 64), cov_lane ? a.cov_S[c * a.cov_sc + lane * a.cov_st] : 0.f, fs);
    mvl = cov_lane ? a.cov_mv[(size_t)u * 32 + lane] : 0.f;
  }
  for (int kk = kbeg; kk < kend; ++kk) {
    const int k = (MODE == kSweepVB && a.order && only < 0) ? a.order[kk] : kk;
    const float xk = __shfl(x, k, 64);
    const float* vcol = a.XoT + (size_t)k * a.ldT_o;
    float corr = 0.f, asq = 0.f, vsq = 0.f;
    for (int e = 0; e < E; ++e) {
      const uint32_t j = idx[e * 64];
      const float v = vcol[j];
      const float t = fmaf(-xk, v, q[e * 64]);
      corr = fmaf(t, v, corr);
      if (MODE == kSweepVB) { asq += a.S2oT[(size_t)k * a.ldT_o + j]; vsq = fmaf(v, v, vsq); }
      else asq = fmaf(v, v, asq);
    }
    if (MODE == kSweepVB) vsq = wave_sum(vsq);
    const float ckl = (lane < KP) ? a.C32[(size_t)k * KP + lane] : 0.f;
    if (lane < KP && lane != k) corr = fmaf(-x, ckl, corr);
    corr = wave_sum(corr);
    asq = wave_sum(asq);
    const float ckk = __shfl(ckl, k, 64);
    const float aik = (MODE == kSweepVB ? (float)a.colsum2_o[k] : ckk) - asq;
    float sc = 0.f;
    if (cov_on) {                                   // bnmtf_vb_optimised.py:246 / :269: sum_t S(k,t) mv_t (fs_t - x_k S(k,t))
      sc = cov_lane ? a.cov_S[k * a.cov_sc + lane * a.cov_st] : 0.f;
      corr -= wave_sum(sc * mvl * fmaf(-xk, sc, fs));
    }
    const float num = __shfl(p, k, 64) + corr;
    const float tau_p = tau * aik;
    const float numer = fmaf(tau, num, -__shfl(lam, k, 64));
    if (a.cond_k >= 0) {
      if (lane == 0) { a.numer_out[u] = (double)numer; a.tau_out[u] = (double)tau_p; }
      return;
    }
    const float mu = numer / tau_p;
    float xnew = 0.f;
    if (MODE == kSweepDraw) {
      const TnParams tp = tn_params(mu, tau_p);
      if (tp.live) {
        for (uint32_t round = 0; round < 64u; ++round) {
          float xc;
          const bool acc = tn_candidate(tp, (uint32_t)gi, a.col0 + (uint32_t)k, a.it, a.stream, round * 64u + lane,
                                        a.key0, a.key1, &xc);
          const unsigned long long m = __ballot(acc);
          if (m) { xnew = tn_guard(__shfl(xc, __ffsll((long long)m) - 1, 64)); break; }
        }
      }
    } else if (MODE == kSweepMode) {
      xnew = fmaxf((tau_p > 0.f && mu > 0.f) ? mu : 0.f, a.min_x);
    } else {
      double e_, v_;
      if (a.vb_moments) { float ef_, vf_; tn_moments_f32(mu, tau_p, &ef_, &vf_); e_ = (double)ef_; v_ = (double)vf_; }   // the sweeps' fp32 routine (device_rng.h)
      else { e_ = (double)xk; v_ = (double)__shfl(var_in, k, 64); }       // update_U(k) without update_exp_U(k)
      xnew = (float)e_;
      if (lane == k) { mu_l = mu; tau_l = tau_p; var_l = (float)v_; }
      // pieces of elbo() (:163-177) and of exp_square_diff (:185-187) for this (unit, k)
      const double dm = e_ - (double)mu;
      e_quad += 0.5 * (double)tau_p * (v_ + dm * dm);
      e_lerfc += log(0.5 * erfc(-(double)mu * sqrt((double)tau_p) * 0.7071067811865476));
      e_ltau += log((double)tau_p);
      e_lamx += (double)__shfl(lam, k, 64) * e_;
      e_q2 += (v_ + e_ * e_) * (double)asq;                       // sum_miss S2self_k S2other_k
      e_q3 += e_ * e_ * (double)vsq;                               // sum_miss exp_self_k^2 exp_other_k^2
    }
    const float delta = xnew - xk;
    if (lane == k) x = xnew;
    if (cov_on) fs = fmaf(delta, sc, fs);
    for (int e = 0; e < E; ++e) {
      const uint32_t j = idx[e * 64];
      q[e * 64] = fmaf(delta, vcol[j], q[e * 64]);
    }
  }

  if (lane < K) {
    a.Xself[gi * KP + lane] = x;
    if (MODE == kSweepVB) {
      a.mu_self[gi * KP + lane] = mu_l; a.tau_self[gi * KP + lane] = tau_l; a.var_self[gi * KP + lane] = var_l;
      const float s2 = var_l + x * x;
      a.S2self[gi * KP + lane] = s2;
    }
  }
  if (MODE == kSweepVB && a.vb_stats && lane == 0) {          // per-unit partial sums -> slab, summed by the VB finish kernel
    double* o = a.vb_stats + (size_t)u * 8;
    o[0] = e_quad; o[1] = e_lerfc; o[2] = e_ltau; o[3] = e_lamx; o[4] = e_q2; o[5] = e_q3;
  }
  if (a.acc) {
    double px = wave_sum_d((double)p * (double)x);
    double sq = 0.0, sq2 = 0.0;
    for (int e = 0; e < E; ++e) { const double qv = (double)q[e * 64]; sq += qv; sq2 += qv * qv; }
    sq = wave_sum_d(sq); sq2 = wave_sum_d(sq2);
    if (lane == 0) { atomicAdd(a.acc + 0, px); atomicAdd(a.acc + 1, sq); atomicAdd(a.acc + 2, sq2); }
  }
}

void launch_sweep(const SweepArgs& a, hipStream_t st) {
  dim3 grid((a.n + 3) / 4), block(256);
  switch (a.mode) {
    case kSweepDraw: hipLaunchKernelGGL(sweep_generic_kernel<kSweepDraw>, grid, block, 0, st, a); break;
    case kSweepMode: hipLaunchKernelGGL(sweep_generic_kernel<kSweepMode>, grid, block, 0, st, a); break;
    default:         hipLaunchKernelGGL(sweep_generic_kernel<kSweepVB>, grid, block, 0, st, a); break;
  }
}

}  // namespace bnmtf
