// BNMTF-only kernels (bnmtf_gibbs_optimised.py:152-167, 195-211).
//
// F columns and G columns are the BNMF half sweeps with an "effective" other factor:
//   F step:  R ~ F . (G S^T)^T   -> other = V_eff = G S^T  (J x K)
//   G step:  R ~ (F S) . G^T     -> other = U_eff = F S    (I x L)
// small_product_kernel forms those.  The S entries (K*L sequential scalars, :157-160) are
// restated per row k of S:
//   h_kj = sum_i M_ij (R_ij - P_ij) F_ik = Pv_jk - sum_l' G_jl' (Cf S)_kl' + sum_{i in miss(j)} q_ij F_ik
//   w_kj = sum_i M_ij F_ik^2            = Cf_kk - sum_{i in miss(j)} F_ik^2
//   eta_l = sum_j G_jl h_kj ,  Omega_ll' = sum_j w_kj G_jl G_jl'
//   for l = 0..L-1:  tauS_kl = tau Omega_ll ,
//                    muS_kl  = (-lambdaS_kl + tau (eta_l + S_kl Omega_ll - sum_{l''<l} delta_l'' Omega_l''l)) / tauS_kl
// (P = F S G^T, Pv = R~^T F, Cf = F^T F, q = P on the missing entries, delta = S_new - S_old).
// srow_gather_kernel makes the J-vectors and per-block partial eta/Omega for one k;
// srow_draw_kernel reduces them, runs the L sequential draws and propagates delta.
#include <algorithm>

#include "sweep_common.h"

namespace bnmtf {

// out[r][c] = sum_t X[r][t] * (transposeS ? S[c][t] : S[t][c]);  X [rows][KPin], S [K][L] row major, out [rows][KPout]
__global__ __launch_bounds__(256) void small_product_kernel(SmallProductArgs a) {
  __shared__ float Ss[64 * 64];
  for (int t = threadIdx.x; t < a.K * a.L; t += 256) Ss[t] = a.S[t];
  __syncthreads();
  const int inner = a.transposeS ? a.L : a.K, outw = a.transposeS ? a.K : a.L;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)a.rows * outw; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e / outw), c = (int)(e % outw);
    const float* x = a.X + (size_t)r * a.KPin;
    float s = 0.f;
    for (int t = 0; t < inner; ++t) s = fmaf(x[t], a.transposeS ? Ss[c * a.L + t] : Ss[t * a.L + c], s);
    a.out[(size_t)r * a.KPout + c] = s;
  }
}
void launch_small_product(const SmallProductArgs& a, hipStream_t st) {
  const int outw = a.transposeS ? a.K : a.L;
  const int blocks = (int)std::min<size_t>(2048, ((size_t)a.rows * outw + 255) / 256);
  hipLaunchKernelGGL(small_product_kernel, dim3(blocks), dim3(256), 0, st, a);
}

// CfS[k][l] = sum_k' Cf[k][k'] S[k'][l]
__global__ void cfs_kernel(const double* Cf64, int KPk, const float* S, int K, int L, float* CfS) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= K * L) return;
  const int k = t / L, l = t % L;
  double s = 0.0;
  for (int kk = 0; kk < K; ++kk) s += Cf64[(size_t)k * KPk + kk] * (double)S[kk * L + l];
  CfS[t] = (float)s;
}
void launch_cfs(const double* Cf64, int KPk, const float* S, int K, int L, float* CfS, hipStream_t st) {
  hipLaunchKernelGGL(cfs_kernel, dim3((K * L + 255) / 256), dim3(256), 0, st, Cf64, KPk, S, K, L, CfS);
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// One wave per column j (grid-strided).  Lane l (< L) holds G_jl.
__global__ __launch_bounds__(256) void srow_gather_kernel(SRowArgs a) {
  __shared__ float red[64 * 65 + 64];
  for (int t = threadIdx.x; t < 64 * 65 + 64; t += 256) red[t] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = a.L, k = a.k;
  float eta = 0.f;                    // lane l: sum_j G_jl h_kj
  float om[64];                       // lane l: Omega[l][l'] partial
#pragma unroll
  for (int t = 0; t < 64; ++t) om[t] = 0.f;
  const float* FTk = a.FT + (size_t)k * a.ldT;
  const float* FTp = a.FT + (size_t)(k > 0 ? k - 1 : 0) * a.ldT;
  const float cfs_l = (lane < L) ? a.CfS[k * L + lane] : 0.f;
  const float cfkk = *a.cfkk_ptr;
  const float dprev_l = (k > 0 && a.apply_prev && lane < L) ? a.delta_prev[lane] : 0.f;
  for (int u = blockIdx.x * 4 + wave; u < a.n; u += gridDim.x * 4) {
    const float g = (lane < L) ? a.G[((size_t)a.n0 + u) * a.KPl + lane] : 0.f;
    const float dj = wsum(g * dprev_l);                              // G_j . delta_{k-1}
    float pv = 0.f;                                                   // Pv_jk = sum of the contraction's partial slabs
    for (int s = lane; s < a.split; s += 64) pv += a.slabs[((size_t)s * a.n_pad + u) * a.KPk + k];
    const uint32_t s0 = a.slot_ptr[u], s1 = a.slot_ptr[u + 1];
    const uint32_t* __restrict__ idxp = a.idx;
    float* __restrict__ qp = a.q;
    const bool upd = k > 0 && a.apply_prev;
    float hm = 0.f, wm = 0.f;
    for (uint32_t e = s0 + lane; e < s1; e += 128) {                 // two independent slots per trip
      const uint32_t e2 = e + 64;
      const bool two = e2 < s1;
      const uint32_t i0 = idxp[e], i1 = two ? idxp[e2] : idxp[e];
      float q0 = qp[e], q1 = two ? qp[e2] : 0.f;
      const float f0 = FTk[i0], f1 = two ? FTk[i1] : 0.f;
      if (upd) {
        q0 = fmaf(FTp[i0], dj, q0); qp[e] = q0;
        if (two) { q1 = fmaf(FTp[i1], dj, q1); qp[e2] = q1; }
      }
      hm = fmaf(q0, f0, fmaf(q1, f1, hm));
      wm = fmaf(f0, f0, fmaf(f1, f1, wm));
    }
    // one combined butterfly for the four per-unit sums (hm, wm, pv, G.CfS)
    float r0 = hm, r1 = wm, r2 = pv, r3 = g * cfs_l;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { r0 += __shfl_xor(r0, m, 64); r1 += __shfl_xor(r1, m, 64); r2 += __shfl_xor(r2, m, 64); r3 += __shfl_xor(r3, m, 64); }
    hm = r0; wm = r1;
    const float h = r2 - r3 + hm;
    const float w = cfkk - wm;
    eta = fmaf(g, h, eta);
    const float wg = w * g;
#pragma unroll
    for (int lp = 0; lp < 64; ++lp)
      if (lp < L) om[lp] = fmaf(wg, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g), lp)), om[lp]);
  }
  // block partials -> slab [block][L + L*L]
  // the four waves add their partials in turn (LDS float atomics compile to a CAS loop here: 60k stall cycles)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      red[64 * 65 + lane] += eta;
#pragma unroll
      for (int lp = 0; lp < 64; ++lp)
        if (lp < L) red[lane * 65 + lp] += om[lp];
    }
    __syncthreads();
  }
  float* out = a.partial + (size_t)blockIdx.x * (L + L * L);
  for (int t = threadIdx.x; t < L + L * L; t += 256)
    out[t] = (t < L) ? red[64 * 65 + t] : red[((t - L) / L) * 65 + (t - L) % L];
}
void launch_srow_gather(const SRowArgs& a, int blocks, hipStream_t st) {
  hipLaunchKernelGGL(srow_gather_kernel, dim3(blocks), dim3(256), 0, st, a);
}

// sum the per-block partial (eta, Omega) slabs: 32 outputs per block, 32 partial strides per output
__global__ __launch_bounds__(1024) void srow_reduce_kernel(const float* partial, int nblocks, int nvals, float* out) {
  __shared__ float red[1024];
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + e;
  float s = 0.f;
  if (t < nvals)
    for (int b = g; b < nblocks; b += 32) s += partial[(size_t)b * nvals + t];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 16; w >= 1; w >>= 1) {
    if (g < w) red[threadIdx.x] += red[threadIdx.x + 32 * w];
    __syncthreads();
  }
  if (g == 0 && t < nvals) out[t] = red[e];
}

// Single block: run the L sequential conditionals of row k on the reduced (eta, Omega), write S[k][:],
// delta_k, and keep Cf.S current.  cond_l >= 0: only evaluate (k, cond_l) and write numer/tau.
__global__ __launch_bounds__(1024) void srow_draw_kernel(SDrawArgs a) {
  __shared__ float eta[64], Om[64 * 65], delta[64], srow[64];
  const int L = a.L, K = a.K, k = a.k, tid = threadIdx.x;
  for (int t = tid; t < L + L * L; t += 1024) {
    const float s = a.partial[t];
    if (t < L) eta[t] = s; else Om[((t - L) / L) * 65 + (t - L) % L] = s;
  }
  if (tid < L) { delta[tid] = 0.f; srow[tid] = a.S[k * L + tid]; }
  __syncthreads();
  if (tid < 64) {                                   // one wave: the sequential l loop
    const float tau = *a.tau;
    // hoisted Philox: lane l holds candidates 0 and 1 of entry (k, l)
    uint32_t c0a = 0, c0b = 0, c1a = 0, c1b = 0;
    if (a.update == 0 && a.cond_l < 0 && tid < L) {
      const U4 r0 = philox4x32_10(0u, (uint32_t)(k * L + tid), a.it, kStreamS, a.key0, a.key1);
      const U4 r1 = philox4x32_10(0u, (uint32_t)(k * L + tid), a.it, kStreamS + 16u, a.key0, a.key1);
      c0a = r0.x; c0b = r0.y; c1a = r1.x; c1b = r1.y;
    }
    const int lbeg = a.cond_l >= 0 ? a.cond_l : 0, lend = a.cond_l >= 0 ? a.cond_l + 1 : L;
    for (int l = lbeg; l < lend; ++l) {
      float corr = (tid < l && a.cond_l < 0) ? delta[tid] * Om[tid * 65 + l] : 0.f;
      corr = wsum(corr);
      const float oll = Om[l * 65 + l];
      const float sold = srow[l];
      const float tau_p = tau * oll;
      const float numer = fmaf(tau, eta[l] + sold * oll - corr, -a.lambdaS[k * L + l]);
      if (a.cond_l >= 0) { if (tid == 0) { a.numer_out[0] = (double)numer; a.tau_out[0] = (double)tau_p; } break; }
      const float mu = numer / tau_p;
      float snew = 0.f;
      if (a.update == 0) {
        const TnFast tp = tn_fast_params(numer, tau_p);           // the one-instruction forms the factor sweeps use
        if (tp.live) {
          // candidates 0, 1 from the hoisted words (lane l), then 64 fresh candidates per round
          const uint32_t w0a = (uint32_t)__builtin_amdgcn_readlane((int)c0a, l), w0b = (uint32_t)__builtin_amdgcn_readlane((int)c0b, l);
          const uint32_t w1a = (uint32_t)__builtin_amdgcn_readlane((int)c1a, l), w1b = (uint32_t)__builtin_amdgcn_readlane((int)c1b, l);
          float x0, x1;
          const bool a0 = tn_eval_fast(tp, w0a, w0b, &x0), a1 = tn_eval_fast(tp, w1a, w1b, &x1);
          if (a0) snew = tn_guard(x0);
          else if (a1) snew = tn_guard(x1);
          else {
            for (uint32_t round = 0; round < 64u; ++round) {
              const U4 r = philox4x32_10(0u, (uint32_t)(k * L + l), a.it, kStreamS + 16u * (2u + round * 64u + tid), a.key0, a.key1);
              float xc;
              const bool acc = tn_eval_fast(tp, r.x, r.y, &xc);
              const unsigned long long m = __ballot(acc);
              if (m) { snew = tn_guard(__shfl(xc, __ffsll((long long)m) - 1, 64)); break; }
            }
          }
        }
      } else {
        snew = fmaxf((tau_p > 0.f && mu > 0.f) ? mu : 0.f, a.min_x);
      }
      if (tid == 0) { delta[l] = snew - sold; srow[l] = snew; }
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  if (a.cond_l >= 0) return;
  if (tid < L) { a.S[k * L + tid] = srow[tid]; a.delta_out[tid] = delta[tid]; }
  for (int t = tid; t < K * L; t += 1024) {           // (Cf S)[k'][l] += Cf[k'][k] delta_l
    const int kp = t / L, l = t % L;
    a.CfS[t] += (float)a.Cf64[(size_t)kp * a.KPk + k] * delta[l];
  }
}
void launch_srow_draw(const SDrawArgs& a, hipStream_t st) {
  const int nvals = a.L + a.L * a.L;
  hipLaunchKernelGGL(srow_reduce_kernel, dim3((nvals + 31) / 32), dim3(1024), 0, st, a.partial, a.nblocks, nvals, a.reduced);
  SDrawArgs b = a;
  b.partial = a.reduced;
  hipLaunchKernelGGL(srow_draw_kernel, dim3(1), dim3(1024), 0, st, b);
}

}  // namespace bnmtf
