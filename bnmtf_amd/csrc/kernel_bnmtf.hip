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
// w_k and Omega^k do not depend on S: srow_w_kernel / srow_omega_kernel form them for every k once per iteration.
// Per row k, srow_gather_kernel makes h_k (carrying q forward by the previous row's delta) and per-block partial eta;
// srow_draw_kernel sums the partials, runs the L sequential draws and propagates delta.
#include <algorithm>

#include "sweep_common.h"

namespace bnmtf {

// out[r][c] = sum_t X[r][t] * (transposeS ? S[c][t] : S[t][c]);  X [rows][KPin], S [K][L] row major, out [rows][KPout]
// A block takes kSpRows rows: their X rows go through LDS (one coalesced pass), S sits there as M[t][c] (the inner index
// first: a lane's column c is its bank), thread (row, c) sums its `inner` products from LDS -- round 5: the version that read
// X[r][t] from global memory inside the loop was 7 us of load latency for 4 M products.
constexpr int kSpRows = 8;
__global__ __launch_bounds__(256) void small_product_kernel(SmallProductArgs a) {
  __shared__ float Ms[64 * 65], Xs[kSpRows][64];
  const int inner = a.transposeS ? a.L : a.K, outw = a.transposeS ? a.K : a.L, tid = threadIdx.x;
  for (int t = tid; t < a.K * a.L; t += 256) {
    const int k = t / a.L, l = t % a.L;
    if (a.transposeS) Ms[l * 65 + k] = a.S[t]; else Ms[k * 65 + l] = a.S[t];      // M[inner][out]
  }
  const int r0 = blockIdx.x * kSpRows;
  for (int t = tid; t < kSpRows * a.KPin; t += 256) {
    const int r = t / a.KPin, c = t % a.KPin;
    Xs[r][c] = r0 + r < a.rows ? a.X[(size_t)(r0 + r) * a.KPin + c] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < kSpRows * outw; e += 256) {
    const int r = e / outw, c = e % outw;
    float s = 0.f;
    for (int t = 0; t < inner; ++t) s = fmaf(Xs[r][t], Ms[t * 65 + c], s);
    if (r0 + r < a.rows) a.out[(size_t)(r0 + r) * a.KPout + c] = s;
  }
}
void launch_small_product(const SmallProductArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(small_product_kernel, dim3((a.rows + kSpRows - 1) / kSpRows), dim3(256), 0, st, a);
}

// out[j][c] = sum_t (sum_s slabs[s][j][t]) S[t][c]: the G sweep's contraction R~^T (F S) as (R~^T F) S -- R~^T F is there
// already (the S step's own contraction, F has not changed since), so the second pass over R~ shrinks to a K x L product per
// column.  Half wave per unit: lane c holds the summed slab entry t = c and column c of the product.
__global__ __launch_bounds__(256) void slab_product_kernel(SlabProductArgs a) {
  __shared__ float Ss[64 * 64];                 // [inner index][output column], row stride 64, either way round
  for (int t = threadIdx.x; t < a.K * a.L; t += 256) {
    const int k = t / a.L, l = t % a.L;
    Ss[a.transposeS ? l * 64 + k : k * 64 + l] = a.S[t];
  }
  __syncthreads();
  const int l5 = threadIdx.x & 31;
  const int u = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (u >= a.n) return;
  const size_t stride = (size_t)a.n_pad * a.KPin;
  const int inner = a.transposeS ? a.L : a.K, outw = a.transposeS ? a.K : a.L;      // (transposeS: out = slabs . S^T)
  for (int c0 = 0; c0 < a.KPout; c0 += 32) {
    float acc = 0.f;
    for (int t0 = 0; t0 < inner; t0 += 32) {
      const float tv = t0 + l5 < inner ? slab_sum_ordered(a.slabs, a.split, stride, (size_t)u * a.KPin + t0 + l5) : 0.f;
      for (int t = 0; t < 32 && t0 + t < inner; ++t) {
        const float x = __shfl(tv, t, 32);
        if (c0 + l5 < outw) acc = fmaf(x, Ss[(t0 + t) * 64 + c0 + l5], acc);
      }
    }
    a.out[(size_t)u * a.KPout + c0 + l5] = c0 + l5 < outw ? acc : 0.f;
  }
}
void launch_slab_product(const SlabProductArgs& a, hipStream_t st) {
  if (a.n > 0) hipLaunchKernelGGL(slab_product_kernel, dim3((a.n + 7) / 8), dim3(256), 0, st, a);
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

// w[j][k] = sum_i M_ij F_ik^2 = Cf_kk - sum_{i in miss(j)} F_ik^2 for every k: one wave per column j, lane k.  Does not
// depend on S, so it is formed once per iteration, ahead of the K sequential rows.
__global__ __launch_bounds__(256) void srow_w_kernel(SOmegaArgs a) {
  const int lane = threadIdx.x & 63;
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= a.n) return;
  const uint32_t s0 = a.slot_ptr[u], s1 = a.slot_ptr[u + 1];
  float tot;
  if (a.KPk == 32) {
    // a 32-lane half takes one entry per trip (lane = k): coalesced 128-byte rows of F, eight trips in flight
    const int half = lane >> 5, l5 = lane & 31;
    float acc = 0.f;
    for (uint32_t e0 = s0; e0 < s1; e0 += 16) {
      uint32_t ii[8]; float fv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) { const uint32_t e = e0 + 2u * t + half; ii[t] = e < s1 ? a.idx[e] : (uint32_t)a.zero_row; }
#pragma unroll
      for (int t = 0; t < 8; ++t) fv[t] = a.F[(size_t)ii[t] * 32 + l5];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc = fmaf(fv[t], fv[t], acc);
    }
    tot = acc + __shfl_xor(acc, 32, 64);
  } else {
    const int kk = lane < a.KPk ? lane : 0;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (uint32_t e0 = s0; e0 < s1; e0 += 64) {
      const uint32_t iv = (e0 + lane < s1) ? a.idx[e0 + lane] : (uint32_t)a.zero_row;     // padding slots point at the zero row too
      const int cnt = (int)min(64u, s1 - e0);
      int t = 0;
      for (; t + 4 <= cnt; t += 4) {
        const float f0 = a.F[(size_t)__builtin_amdgcn_readlane((int)iv, t) * a.KPk + kk];
        const float f1 = a.F[(size_t)__builtin_amdgcn_readlane((int)iv, t + 1) * a.KPk + kk];
        const float f2 = a.F[(size_t)__builtin_amdgcn_readlane((int)iv, t + 2) * a.KPk + kk];
        const float f3 = a.F[(size_t)__builtin_amdgcn_readlane((int)iv, t + 3) * a.KPk + kk];
        acc0 = fmaf(f0, f0, acc0); acc1 = fmaf(f1, f1, acc1); acc2 = fmaf(f2, f2, acc2); acc3 = fmaf(f3, f3, acc3);
      }
      for (; t < cnt; ++t) { const float f0 = a.F[(size_t)__builtin_amdgcn_readlane((int)iv, t) * a.KPk + kk]; acc0 = fmaf(f0, f0, acc0); }
    }
    tot = (acc0 + acc1) + (acc2 + acc3);
  }
  if (lane < a.KPk) a.w[(size_t)u * a.KPk + lane] = (lane < a.K ? a.Cf32[(size_t)lane * a.KPk + lane] : 0.f) - tot;
}

// q_ij = (F S)_i . G_j on the missing entries of every column j, in the order of the column's (64-wide) slots: what the S
// rows carry forward.  One wave per column; a 32-lane half takes one entry per trip (lane = l): a coalesced 128-byte
// row of U_eff = F S, times G_j held in registers, summed inside the half.  KPl = 32 (L <= 32) only.
__global__ __launch_bounds__(256) void srow_qinit_kernel(SOmegaArgs a, const float* Ueff, float* q) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l5 = lane & 31;
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= a.n) return;
  const float g = (l5 < a.L) ? a.G[((size_t)a.n0 + u) * a.KPl + l5] : 0.f;
  const uint32_t s0 = a.slot_ptr[u], s1 = a.slot_ptr[u + 1];
  for (uint32_t e0 = s0; e0 < s1; e0 += 16) {                          // eight trips of two entries in flight
    uint32_t ii[8]; float uv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { const uint32_t e = e0 + 2u * t + half; ii[t] = e < s1 ? a.idx[e] : (uint32_t)a.zero_row; }
#pragma unroll
    for (int t = 0; t < 8; ++t) uv[t] = Ueff[(size_t)ii[t] * a.KPl + l5];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float sm = half_sum_upper(uv[t] * g);                      // right in lanes 16-31 of the half
      const uint32_t e = e0 + 2u * t + half;
      if (l5 == 16 && e < s1) q[e] = sm;
    }
  }
}
void launch_srow_qinit(const SOmegaArgs& a, const float* Ueff, float* q, hipStream_t st) {
  if (a.n > 0) hipLaunchKernelGGL(srow_qinit_kernel, dim3((a.n + 3) / 4), dim3(256), 0, st, a, Ueff, q);
}

// Omega^k[l][l'] = sum_j w_kj G_jl G_jl' for every row k of S, as nch partial sums over column chunks: block (k, c).
// LP = padded L (32 or 64); thread t owns l = t mod LP and NACC consecutive l'.
template <int LP>
__global__ __launch_bounds__(256) void srow_omega_kernel(SOmegaArgs a) {
  constexpr int TJ = 64, NACC = LP * LP / 256;
  __shared__ float Gs[TJ][LP], ws[TJ];
  const int k = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
  const int chunk = (a.n + a.nch - 1) / a.nch;
  const int j0 = c * chunk, j1 = min(a.n, j0 + chunk);
  const int l = tid % LP, g = tid / LP;
  float acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = 0.f;
  for (int jb = j0; jb < j1; jb += TJ) {
    const int nj = min(TJ, j1 - jb);
    __syncthreads();
    for (int t = tid; t < TJ * LP; t += 256) {
      const int jj = t / LP, ll = t % LP;
      Gs[jj][ll] = (jj < nj && ll < a.L) ? a.G[((size_t)a.n0 + jb + jj) * a.KPl + ll] : 0.f;
    }
    if (tid < TJ) ws[tid] = tid < nj ? a.w[(size_t)(jb + tid) * a.KPk + k] : 0.f;
    __syncthreads();
#pragma unroll 4
    for (int jj = 0; jj < TJ; ++jj) {
      const float wg = ws[jj] * Gs[jj][l];
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = fmaf(wg, Gs[jj][g * NACC + t], acc[t]);
    }
  }
  float* out = a.omp + ((size_t)k * a.nch + c) * LP * LP + (size_t)l * LP + g * NACC;
#pragma unroll
  for (int t = 0; t < NACC; ++t) out[t] = acc[t];
}
void launch_srow_omega(const SOmegaArgs& a, hipStream_t st) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(srow_w_kernel, dim3((a.n + 3) / 4), dim3(256), 0, st, a);
  if (a.KPl == 32) hipLaunchKernelGGL(srow_omega_kernel<32>, dim3(a.K, a.nch), dim3(256), 0, st, a);
  else             hipLaunchKernelGGL(srow_omega_kernel<64>, dim3(a.K, a.nch), dim3(256), 0, st, a);
}

// Row k: one wave per column j, 16 columns per block.  Lane l (< L) holds G_jl; the block's share of eta_l = sum_j G_jl h_kj
// goes to partial[block][L] (summed by the draw kernel in a fixed order).
__global__ __launch_bounds__(1024) void srow_gather_kernel(SRowArgs a) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = a.L, k = a.k;
  const int u = blockIdx.x * 16 + wave;
  float contrib = 0.f;
  if (u < a.n) {
    const float* FTk = a.FT + (size_t)k * a.ldT;
    const float* FTp = a.FT + (size_t)(k > 0 ? k - 1 : 0) * a.ldT;
    const bool upd = k > 0 && a.apply_prev;
    const float cfs_l = (lane < L) ? a.CfS[k * L + lane] : 0.f;
    const float dprev_l = (upd && lane < L) ? a.delta_prev[lane] : 0.f;
    const float g = (lane < L) ? a.G[((size_t)a.n0 + u) * a.KPl + lane] : 0.f;
    float pv = 0.f;                                                   // Pv_jk = sum of the contraction's partial slabs
    for (int s = lane; s < a.split; s += 64) pv += a.slabs[((size_t)s * a.n_pad + u) * a.KPk + k];
    const uint32_t s0 = a.slot_ptr[u], s1 = a.slot_ptr[u + 1];
    const uint32_t* __restrict__ idxp = a.idx;
    float* __restrict__ qp = a.q;
    // one combined butterfly for the per-unit sums (G.delta, pv, G.CfS)
    float r0 = g * dprev_l, r2 = pv, r3 = g * cfs_l;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { r0 += __shfl_xor(r0, m, 64); r2 += __shfl_xor(r2, m, 64); r3 += __shfl_xor(r3, m, 64); }
    const float dj = r0;                                              // G_j . delta_{k-1}
    float hm = 0.f;
    for (uint32_t e = s0 + lane; e < s1; e += 256) {                 // four independent slots per trip: all index loads, then all gathers
      uint32_t ii[4]; float qq[4], ff[4], fp[4]; bool on[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { on[t] = e + 64u * t < s1; ii[t] = on[t] ? idxp[e + 64u * t] : idxp[e]; qq[t] = on[t] ? qp[e + 64u * t] : 0.f; }
#pragma unroll
      for (int t = 0; t < 4; ++t) { ff[t] = on[t] ? FTk[ii[t]] : 0.f; fp[t] = (upd && on[t]) ? FTp[ii[t]] : 0.f; }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (upd && on[t]) { qq[t] = fmaf(fp[t], dj, qq[t]); qp[e + 64u * t] = qq[t]; }
        hm = fmaf(qq[t], ff[t], hm);
      }
    }
    hm = wsum(hm);
    contrib = g * (r2 - r3 + hm);
  }
  red[wave][lane] = contrib;
  __syncthreads();
  if (threadIdx.x < L) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w][threadIdx.x];
    a.partial[(size_t)blockIdx.x * L + threadIdx.x] = s;
  }
}
void launch_srow_gather(const SRowArgs& a, int blocks, hipStream_t st) {
  if (blocks > 0) hipLaunchKernelGGL(srow_gather_kernel, dim3(blocks), dim3(1024), 0, st, a);
}

// Single block: sum the partials, run the L sequential conditionals of row k, write S[k][:], delta_k, and keep Cf.S
// current.  cond_l >= 0: only evaluate (k, cond_l) and write numer/tau.
// The sequential part is one wave with lane l' owning entry (k, l'): its running correction sum_{l''<l'} delta_l'' Omega_l''l'
// grows by one FMA per step (row l of Omega read from LDS ahead of the chain), so a step is readlanes + the sampler, with
// no reduction and no LDS round trip; the first four candidates of every entry are hoisted (lane l' holds those of
// (k, l')) and evaluated by lanes 0-3 at once, later ones 64 at a time.
__global__ __launch_bounds__(1024) void srow_draw_kernel(SDrawArgs a) {
  __shared__ float eta[64], Om[64 * 65], delta[64], red[32][64], cands[64 * 4 * 3];
  const int L = a.L, K = a.K, k = a.k, tid = threadIdx.x;
  {   // eta = the gather kernel's block partials (NG interleaved running sums, then those in order); Omega^k = its chunk partials
    const int NG = 1024 / a.LP;                        // groups of LP threads; group g takes blocks g, g + NG, ...
    const int l = tid % a.LP, grp = tid / a.LP;
    float sp[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) sp[t] = 0.f;
    if (l < L)
      for (int b0 = grp; b0 < a.nblocks; b0 += 8 * NG) {
#pragma unroll
        for (int t = 0; t < 8; ++t) { const int b = b0 + t * NG; if (b < a.nblocks) sp[t] += a.partial[(size_t)b * L + l]; }
      }
    red[grp][l] = ((sp[0] + sp[1]) + (sp[2] + sp[3])) + ((sp[4] + sp[5]) + (sp[6] + sp[7]));
    const float* op = a.omp + (size_t)k * kSOmegaChunks * a.LP * a.LP;
    for (int t = tid; t < L * L; t += 1024) {
      const int l1 = t / L, l2 = t % L;
      float o[kSOmegaChunks];
#pragma unroll
      for (int c = 0; c < kSOmegaChunks; ++c) o[c] = op[(size_t)c * a.LP * a.LP + l1 * a.LP + l2];
      float os = 0.f;
#pragma unroll
      for (int c = 0; c < kSOmegaChunks; ++c) os += o[c];
      Om[l1 * 65 + l2] = os;
    }
    if (tid < 64) delta[tid] = 0.f;
    __syncthreads();
    if (tid < L) { float e = 0.f; for (int w = 0; w < NG; ++w) e += red[w][tid]; eta[tid] = e; }
    __syncthreads();
  }
  if (tid < 64) {                                   // one wave: the sequential l loop
    const int lane = tid;
    const bool on = lane < L;
    const float tau = *a.tau;
    const float my_eta = on ? eta[lane] : 0.f, my_oll = on ? Om[lane * 65 + lane] : 0.f, my_lam = on ? a.lambdaS[k * L + lane] : 0.f;
    float my_s = on ? a.S[k * L + lane] : 0.f, my_delta = 0.f, corr = 0.f;
    const TnPre my_pre = tn_fast_pre(tau * my_oll);     // tau_p of every entry is known before the chain starts
    constexpr int NH = 4;                            // hoisted candidates per entry
    // lane l' makes the word-only half (tn_cand_pre) of the first NH candidates of entry (k, l') and parks it in LDS;
    // at step l lane c < NH reads candidate c of entry l back, ahead of the chain
    if (a.update == 0 && a.cond_l < 0 && on) {
#pragma unroll
      for (int c = 0; c < NH; ++c) {
        const U4 r = philox4x32_10(0u, a.word0 + (uint32_t)(k * a.ldword + lane), a.it, kStreamS + 16u * (uint32_t)c, a.key0, a.key1);
        const TnCand cd = tn_cand_pre(r.x, r.y);
        cands[(lane * NH + c) * 3 + 0] = cd.nl; cands[(lane * NH + c) * 3 + 1] = cd.z; cands[(lane * NH + c) * 3 + 2] = cd.sw;
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const int lbeg = a.cond_l >= 0 ? a.cond_l : 0, lend = a.cond_l >= 0 ? a.cond_l + 1 : L;
    for (int l = lbeg; l < lend; ++l) {
      const float row = on ? Om[l * 65 + lane] : 0.f;                  // Omega[l][lane]: off the chain
      TnCand cd0 = {0.f, 0.f, 0.f};                                     // hoisted candidate `lane` of entry l: off the chain too
      if (lane < NH) { cd0.nl = cands[(l * NH + lane) * 3 + 0]; cd0.z = cands[(l * NH + lane) * 3 + 1]; cd0.sw = cands[(l * NH + lane) * 3 + 2]; }
      const float numer_v = fmaf(tau, my_eta + my_s * my_oll - corr, -my_lam);
      const float numer = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, numer_v), l));
      const float oll = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_oll), l));
      const float sold = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_s), l));
      const float tau_p = tau * oll;
      if (a.cond_l >= 0) { if (tid == 0) { a.numer_out[0] = (double)numer; a.tau_out[0] = (double)tau_p; } break; }
      float snew = 0.f;
      if (a.update == 0) {
        TnPre pre;
        pre.irt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_pre.irt), l));
        pre.rcp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_pre.rcp), l));
        pre.tpirt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_pre.tpirt), l));
        pre.live = tau_p > 0.0f;
        const TnFast tp = tn_fast_post(pre, numer);              // the one-instruction forms the factor sweeps use
        if (tp.live) {
          float xc;
          bool acc = tn_cand_post(tp, cd0, &xc) && lane < NH;
          unsigned long long m = __ballot(acc);
          for (uint32_t round = 0; m == 0ull && round < 64u; ++round) {     // candidates NH + 64 round + lane
            const U4 r = philox4x32_10(0u, a.word0 + (uint32_t)(k * a.ldword + l), a.it, kStreamS + 16u * ((uint32_t)NH + round * 64u + (uint32_t)lane), a.key0, a.key1);
            acc = tn_eval_fast(tp, r.x, r.y, &xc);
            m = __ballot(acc);
          }
          if (m) snew = tn_guard(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xc), __ffsll((long long)m) - 1)));
        }
      } else {
        const float mu = numer / tau_p;
        snew = fmaxf((tau_p > 0.f && mu > 0.f) ? mu : 0.f, a.min_x);
      }
      const float dl = snew - sold;
      if (lane == l) { my_s = snew; my_delta = dl; }
      corr = fmaf(dl, row, corr);                     // what lanes l' > l subtract when their turn comes
    }
    if (a.cond_l < 0 && on) { a.S[k * L + lane] = my_s; a.delta_out[lane] = my_delta; delta[lane] = my_delta; }
  }
  __syncthreads();
  if (a.cond_l >= 0) return;
  for (int t = tid; t < K * L; t += 1024) {           // (Cf S)[k'][l] += Cf[k'][k] delta_l
    const int kp = t / L, l = t % L;
    a.CfS[t] += (float)a.Cf64[(size_t)kp * a.KPk + k] * delta[l];
  }
}
void launch_srow_draw(const SDrawArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(srow_draw_kernel, dim3(1), dim3(1024), 0, st, a);
}

}  // namespace bnmtf
