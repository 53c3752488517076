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
// So an iteration costs one masked Gram per column (scol_gram_kernel: f32 MFMA over the ~10 % missing entries), one
// K^2 x J x L^2 GEMM (ssys_gemm_kernel: f32 MFMA, partial slabs over column ranges, summed in a fixed order), and then the
// K.L sequential conditionals touch nothing but A: ssys_chain_kernel keeps the residual r = b - A S on chip and walks the
// entries row by row (a row of S = the lanes of one wave, the running correction of a lane grows by one FMA per step).
// The variational version (second moments) is the same system with E[F_ik F_ik'] = F_ik F_ik' + [k = k'] varF_ik and
// E[G_jl G_jl'] likewise: pass varF / varG.  Sharded over GPUs, W_j and the GEMM cover a rank's own columns and (A, b) is
// summed with ONE all-reduce -- the "K x L Gram" exchange -- after which every rank walks the same chain.
// K, L <= 32 (one 32 x 32 MFMA tile per column / per row of A); larger ranks keep the per-row path of kernel_bnmtf.hip.
#include <algorithm>

#include "sweep_common.h"

namespace bnmtf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// W~_j = C~f - sum_{i in miss(j)} (F_i F_i^T + diag(varF_i)) for every local column j: one wave per column, the 32 x 32
// tile in MFMA accumulators.  v_mfma_f32_32x32x2_f32 takes A[i][k] and B[k][j] from lane (i or j) + 32 k: for the Gram of
// two entries e, e+1 both operands are the same register, F[idx[e + (lane >> 5)]][lane & 31].
__global__ __launch_bounds__(256) void scol_gram_kernel(SColGramArgs a) {
  const int lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31;
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= a.n) return;
  const uint32_t s0 = a.slot_ptr[u], s1 = a.slot_ptr[u + 1];       // 64-wide slots, padded with the zero row
  f32x16 acc;
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
  float dv = 0.f;                                                   // sum_miss varF_i[c] (diagonal, VB)
  uint32_t in[8];                                                   // the NEXT batch's row indices: the row loads of a batch wait for its indices only once, ahead of the loop
#pragma unroll
  for (int t = 0; t < 8; ++t) in[t] = s0 + 2u * t + half < s1 ? a.idx[s0 + 2u * t + half] : 0u;
  for (uint32_t e0 = s0; e0 < s1; e0 += 16) {                       // eight MFMA steps (sixteen entries) per batch
    uint32_t ii[8]; float fv[8], vv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) ii[t] = in[t];
#pragma unroll
    for (int t = 0; t < 8; ++t) { fv[t] = a.F[(size_t)ii[t] * 32 + c]; vv[t] = a.varF ? a.varF[(size_t)ii[t] * 32 + c] : 0.f; }
    if (e0 + 16 < s1) {
#pragma unroll
      for (int t = 0; t < 8; ++t) in[t] = a.idx[e0 + 16 + 2u * t + half];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[t], fv[t], acc, 0, 0, 0); dv += vv[t]; }
  }
  dv += __shfl_xor(dv, 32, 64);
  float* w = a.Wt + (size_t)u * 1024;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int row = (t & 3) + 8 * (t >> 2) + 4 * half;             // C/D layout of the 32 x 32 tile: column on the lane
    float cf = (float)a.Cf64[(size_t)row * 32 + c];
    float m = acc[t];
    if (row == c) { if (a.cf_diag_extra) cf = (float)a.cf_diag_extra[c]; m += dv; }      // VB: C~f_kk = sum_i (E[F_ik]^2 + varF_ik) = the column sum of the second moments
    w[row * 32 + c] = (row < a.K && c < a.K) ? cf - m : 0.f;
  }
}
void launch_scol_gram(const SColGramArgs& a, hipStream_t st) {
  if (a.n > 0) hipLaunchKernelGGL(scol_gram_kernel, dim3((a.n + 3) / 4), dim3(256), 0, st, a);
}

// pair number p = 0 .. K(K+1)/2 - 1 of (k, k'), k <= k', row by row
__host__ __device__ inline void ssys_pair(int p, int K, int* k, int* kp) {
  int kk = 0;
  while (p >= K - kk) { p -= K - kk; ++kk; }
  *k = kk; *kp = kk + p;
}

// A-slab[s][(k,l)][(k',l')] = sum_{j in range s} W~_j[k][k'] G_jl G_jl'  for the block pairs k <= k' (A is symmetric:
// the lower blocks are mirrored by ssys_reduce_kernel).  One MFMA tile (rows l, columns l') is one K-pair (k, k'):
// the A operand is W~_j[k][k'] G_jl (a broadcast scalar times the lane's G), the B operand G_jl'.  Four tiles per wave,
// sixteen per block; grid (pair groups, column ranges).  (The [l = l'] varG_jl term of the variational version is not
// an outer product: ssys_vardiag_kernel adds it.)
__global__ __launch_bounds__(256) void ssys_gemm_kernel(SSysGemmArgs a) {
  const int lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31, wave = threadIdx.x >> 6;
  const int P = a.K * (a.K + 1) / 2, p0 = (blockIdx.x * 4 + wave) * 4, sp = blockIdx.y;
  const int per = ((a.n + a.nsplit - 1) / a.nsplit + 1) & ~1;      // columns per range (even: two per MFMA step)
  const int jbeg = sp * per, jend = min(a.n, jbeg + per);
  int wo[4];                                                        // offset of W~[k][k'] inside a column's 32 x 32 block (pairs past P: any valid one, not stored)
#pragma unroll
  for (int x = 0; x < 4; ++x) { int k, kp; ssys_pair(min(p0 + x, P - 1), a.K, &k, &kp); wo[x] = k * 32 + kp; }
  f32x16 acc[4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[x][t] = 0.f;
  constexpr int NS = 4;                                             // steps of two columns in flight
  for (int j0 = jbeg; j0 < jend; j0 += 2 * NS) {
    float w[NS][4], g[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      const int j = j0 + 2 * t + half;
      const bool on = j < jend;
      const float* wj = a.Wt + (size_t)(on ? j : 0) * 1024;
#pragma unroll
      for (int x = 0; x < 4; ++x) w[t][x] = wj[wo[x]];
      g[t] = on ? a.G[(size_t)(a.n0 + j) * 32 + c] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < NS; ++t)
#pragma unroll
      for (int x = 0; x < 4; ++x) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t][x] * g[t], g[t], acc[x], 0, 0, 0);
  }
  const int n2 = a.K * a.L;
  float* slab = a.slabs + (size_t)sp * n2 * n2;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    if (p0 + x >= P) continue;
    int k, kp; ssys_pair(p0 + x, a.K, &k, &kp);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int l = (t & 3) + 8 * (t >> 2) + 4 * half;              // l = tile row, l' = tile column = c
      if (l < a.L && c < a.L) slab[(size_t)(k * a.L + l) * n2 + kp * a.L + c] = acc[x][t];
    }
  }
}
void launch_ssys_gemm(const SSysGemmArgs& a, hipStream_t st) {
  const int P = a.K * (a.K + 1) / 2;
  hipLaunchKernelGGL(ssys_gemm_kernel, dim3((P + 15) / 16, a.nsplit), dim3(256), 0, st, a);
}

// A = sum of the column-range slabs (in range order) on the block pairs k <= k', mirrored into k > k': one block per pair
__global__ __launch_bounds__(1024) void ssys_reduce_kernel(const float* slabs, int nsplit, int K, int L, float* A) {
  __shared__ float tile[32][33];
  int k, kp; ssys_pair(blockIdx.x, K, &k, &kp);
  const int l = threadIdx.x >> 5, lp = threadIdx.x & 31, n2 = K * L;
  float v = 0.f;
  if (l < L && lp < L) {
    const size_t e = (size_t)(k * L + l) * n2 + kp * L + lp;
    for (int t = 0; t < nsplit; ++t) v += slabs[(size_t)t * n2 * n2 + e];
    A[e] = v;
  }
  tile[l][lp] = v;
  __syncthreads();
  if (k != kp && l < L && lp < L) A[(size_t)(kp * L + l) * n2 + k * L + lp] = tile[lp][l];
}
void launch_ssys_reduce(const float* slabs, int nsplit, int K, int L, float* A, hipStream_t st) {
  hipLaunchKernelGGL(ssys_reduce_kernel, dim3(K * (K + 1) / 2), dim3(1024), 0, st, slabs, nsplit, K, L, A);
}

// b[k][l] = sum_j Pv_jk G_jl over the local columns (Pv = the contraction's partial slabs, summed in slab order).
// Block = 64 columns: Pv and G tiles through LDS, thread (k, l) sums its 64 products; the per-block partials are summed
// in block order by ssys_reduce_kernel.
__global__ __launch_bounds__(1024) void ssys_b_kernel(SSysBArgs a) {
  __shared__ float pv[64][33], g[64][33];
  const int j0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * 32; e += 1024) {
    const int jj = e >> 5, cc = e & 31, j = j0 + jj;
    float p = 0.f, gg = 0.f;
    if (j < a.n) {
      for (int t = 0; t < a.split; ++t) p += a.slabs[((size_t)t * a.n_pad + j) * 32 + cc];
      gg = a.G[(size_t)(a.n0 + j) * 32 + cc];
    }
    pv[jj][cc] = p; g[jj][cc] = gg;
  }
  __syncthreads();
  const int k = threadIdx.x >> 5, l = threadIdx.x & 31;
  float s = 0.f;
#pragma unroll 16
  for (int jj = 0; jj < 64; ++jj) s = fmaf(pv[jj][k], g[jj][l], s);
  if (k < a.K && l < a.L) a.b[(size_t)blockIdx.x * a.K * a.L + k * a.L + l] = s;
}
void launch_ssys_b(const SSysBArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(ssys_b_kernel, dim3(ssys_b_blocks(a.n)), dim3(1024), 0, st, a);
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

// r = b - A S (fp64 accumulation: b and A S nearly cancel at convergence), one wave per row
__global__ __launch_bounds__(256) void ssys_residual_kernel(const float* A, const float* b, const float* S, int n2, float* r) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n2) return;
  double s = 0.0;
  for (int t = lane; t < n2; t += 64) s = fma((double)A[(size_t)row * n2 + t], (double)S[t], s);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if (lane == 0) r[row] = (float)((double)b[row] - s);
}
void launch_ssys_residual(const float* A, const float* b, const float* S, int n2, float* r, hipStream_t st) {
  hipLaunchKernelGGL(ssys_residual_kernel, dim3((n2 + 3) / 4), dim3(256), 0, st, A, b, S, n2, r);
}

// The K.L sequential conditionals, row-major (k, l) (bnmtf_gibbs_optimised.py:157-160), one block of 16 waves.
//   * the residual r = b - A S lives in LDS.  Row k of S is walked by wave 0, lane l owning entry (k, l): its numerator is
//     -lambda + tau (r_l + S_kl A_ll - sum_{l'' < l} delta_l'' A_l''l), the running sum one FMA per step against the row's
//     L x L diagonal block of A (LDS); a step is one readlane, the sampler on wave-uniform values, a ballot.
//   * everything that does not depend on the chain is off it: the first four candidates of EVERY entry are made by the
//     1024 threads before the first row (word-only halves: log, sqrt, cos), waves 1-15 stage the next row's blocks of A
//     while wave 0 walks the current one, and fold the previous row's deltas into the residual of the rows still to come
//     (A is symmetric: a column is read as a coalesced row); the one row that cannot wait for that -- the next one --
//     gets the last deltas from wave 0 itself, out of registers, against the staged off-diagonal block.
//   * the proposal is chosen by a branch (the parameters are wave-uniform): the common normal regime skips the
//     translated-exponential constants.  Same candidate sequence and acceptance rule as oracle/rng.py.
//   cond >= 0: only evaluate entry `cond` (numer, tau_p) and change nothing -- the tauS / muS hook.
template <int UPDATE>      // 0: draws, 1: mode updates (ICM / the deterministic harness)
__global__ __launch_bounds__(1024) void ssys_chain_kernel(SSysChainArgs a) {
#pragma clang fp contract(off)
  constexpr int NH = 4;                                            // hoisted candidates per entry
  // Od(k, k+1) is staged during row k-1 and read during row k+1: three buffers.  One spare row / entry behind Om and
  // cands: the one-step-ahead reads of the chain run one past the end.
  __shared__ float r[1024], Om[2][33 * 33], Od[3][32 * 33], delta[2][32];
  __shared__ float cands[(1024 + 1) * NH * 3];
  const int K = a.K, L = a.L, n2 = K * L, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float tau = *a.tau;
  auto stage_blocks = [&](int k, int t0, int nt) {                 // diagonal block of row k and the block (k, k+1) -> LDS
    for (int t = t0; t < L * L; t += nt) {
      const int l1 = t / L, l2 = t % L;
      Om[k & 1][l1 * 33 + l2] = a.A[(size_t)(k * L + l1) * n2 + k * L + l2];
      if (k + 1 < K) Od[k % 3][l1 * 33 + l2] = a.A[(size_t)(k * L + l1) * n2 + (k + 1) * L + l2];
    }
  };
  if (a.cond >= 0) {                                               // tauS(k,l) / muS(k,l) hook: the residual already holds everything
    if (tid == 0) {
      const float aaa = a.A[(size_t)a.cond * n2 + a.cond];
      a.numer_out[0] = (double)fmaf(tau, a.r0[a.cond] + a.S[a.cond] * aaa, -a.lambdaS[a.cond]);
      a.tau_out[0] = (double)(tau * aaa);
    }
    return;
  }
  if (tid < n2) {
    r[tid] = a.r0[tid];
    if (UPDATE == 0) {
#pragma unroll
      for (int c = 0; c < NH; ++c) {
        const U4 rr = philox4x32_10(0u, (uint32_t)tid, a.it, kStreamS + 16u * (uint32_t)c, a.key0, a.key1);
        const TnCand cd = tn_cand_pre(rr.x, rr.y);
        cands[(tid * NH + c) * 3 + 0] = cd.nl; cands[(tid * NH + c) * 3 + 1] = cd.z; cands[(tid * NH + c) * 3 + 2] = cd.u2;
      }
    }
  }
  if (tid < 64) { delta[0][tid & 31] = 0.f; delta[1][tid & 31] = 0.f; }
  for (int t = tid; t < 2 * 33 * 33; t += 1024) (&Om[0][0])[t] = 0.f;      // lanes beyond L read (and discard) these: keep them finite
  for (int t = tid; t < 3 * 32 * 33; t += 1024) (&Od[0][0])[t] = 0.f;
  __syncthreads();
  stage_blocks(0, tid, 1024);
  __syncthreads();
  float prev_delta = 0.f;                                           // wave 0: delta of the previous row, by lane
  for (int k = 0; k < K; ++k) {
    const int cur = k & 1;
    if (wave == 0) {
      const bool on = lane < L;
      const int l32 = lane & 31;                                    // lanes >= 32 mirror lanes 0-31: every LDS address below is valid, no exec juggling
      float my_eta = r[k * L + (on ? lane : 0)];
      if (k > 0) {                                                  // the previous row's deltas, which the background pass has not folded in yet
        const float* od = Od[(k - 1) % 3];
        for (int l0 = 0; l0 < L; l0 += 8) {                         // eight LDS reads in flight
          float ov[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) ov[j] = (l0 + j < L && on) ? od[((l0 + j) & 31) * 33 + l32] : 0.f;   // rows >= L were never staged: 0 * (stale Inf / NaN) is a NaN
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float dl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prev_delta), (l0 + j) & 63));
            my_eta = fmaf(-dl, ov[j], my_eta);
          }
        }
      }
      const float* om = Om[cur];
      const float my_oll = om[l32 * 33 + l32], my_lam = a.lambdaS[k * L + (on ? lane : 0)];
      float my_s = a.S[k * L + (on ? lane : 0)], my_delta = 0.f, corr = 0.f;
      const float c0 = fmaf(tau, my_eta + my_s * my_oll, -my_lam);  // numerator before the row's own deltas
      const float my_taup = tau * my_oll;
      const TnPre my_pre = tn_fast_pre(my_taup);                    // tau_p of every entry is known before the chain starts
      const float* cbase = &cands[(k * L * NH + (lane & (NH - 1))) * 3];
      float row_n = om[l32];                                        // A[(k,l)][(k,lane)] and candidate `lane` of entry l: one step ahead
      float cnl_n = cbase[0], cz_n = cbase[1], cu2_n = cbase[2];
      for (int l = 0; l < L; ++l) {
        const float row = row_n, cnl = cnl_n, cz = cz_n, cu2 = cu2_n;
        row_n = om[(l + 1) * 33 + l32];
        cnl_n = cbase[(l + 1) * NH * 3 + 0]; cz_n = cbase[(l + 1) * NH * 3 + 1]; cu2_n = cbase[(l + 1) * NH * 3 + 2];
        const float numer_v = fmaf(-tau, corr, c0);
        const float numer = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, numer_v), l));
        const float sold = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_s), l));
        const float tau_p = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_taup), l));
        float snew = 0.f;
        if (UPDATE == 0) {
          // (the entry's constants through LDS, read a step ahead with a uniform address, measured slower than these
          // readlanes: 267 vs 221 us -- the extra LDS waits land on the chain)
          const float irt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_pre.irt), l));
          const float rcp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_pre.rcp), l));
          const float tpirt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_pre.tpirt), l));
          // both proposals evaluated, the regime and the live test are selects (every value here is wave-uniform except the
          // candidate words): one branch per step, for the rare whole-batch rejection
          const float mu = numer * rcp;
          const float av = -mu * tpirt;                             // -mu sqrt(tau_p)
          const bool live = tau_p > 0.0f && isfinite(av);
          const bool tail = av >= kTnA0;
          float d = 0.f, ilam = 0.f, xc = fmaf(cz, irt, mu);
          bool acc = cz >= av;
          if (__builtin_amdgcn_readfirstlane((int)tail)) {         // wave-uniform: the normal regime skips the translated-exponential constants
            d = 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(av, av, 4.0f)) + av);
            ilam = __builtin_amdgcn_rcpf(av + d);
            const float e = cnl * ilam, t = e - d;
            acc = cu2 <= __builtin_amdgcn_exp2f(-0.72134752f * t * t);
            xc = e * irt;
          }
          unsigned long long m = __ballot(acc && live) & ((1ull << NH) - 1ull);
          if (__builtin_expect(m == 0ull && __ballot(live) != 0ull, 0)) {   // candidates NH + 64 round + lane
            TnFast tp; tp.mu = mu; tp.irt = irt; tp.a = av; tp.live = true; tp.tail = tail;
            tp.d = 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(av, av, 4.0f)) + av); tp.ilam = __builtin_amdgcn_rcpf(av + tp.d);
            for (uint32_t round = 0; m == 0ull && round < 64u; ++round) {
              const U4 rr = philox4x32_10(0u, (uint32_t)(k * L + l), a.it, kStreamS + 16u * ((uint32_t)NH + round * 64u + (uint32_t)lane), a.key0, a.key1);
              m = __ballot(tn_eval_fast(tp, rr.x, rr.y, &xc));
            }
          }
          const float xs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xc), m ? __ffsll((long long)m) - 1 : 0));
          snew = m ? tn_guard(xs) : 0.f;
        } else {
          const float mu = numer / tau_p;
          snew = fmaxf((tau_p > 0.f && mu > 0.f) ? mu : 0.f, a.min_x);
        }
        const float dl = snew - sold;
        if (lane == l) { my_s = snew; my_delta = dl; }
        corr = fmaf(dl, row, corr);                                  // what lanes l' > l subtract when their turn comes
      }
      if (on) { a.S[k * L + lane] = my_s; delta[cur][lane] = my_delta; }
      prev_delta = my_delta;
    } else {
      // waves 1-15, beside the chain: the next row's blocks of A, and the deltas of row k-1 folded into rows >= k+1
      if (k + 1 < K) stage_blocks(k + 1, tid - 64, 960);
      if (k > 0) {
        const float* dp = delta[cur ^ 1];
        for (int t = (k + 1) * L + (tid - 64); t < n2; t += 960) {
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
    __syncthreads();
  }
}
void launch_ssys_chain(const SSysChainArgs& a, hipStream_t st) {
  if (a.update == 0) hipLaunchKernelGGL(ssys_chain_kernel<0>, dim3(1), dim3(1024), 0, st, a);
  else               hipLaunchKernelGGL(ssys_chain_kernel<1>, dim3(1), dim3(1024), 0, st, a);
}

}  // namespace bnmtf
