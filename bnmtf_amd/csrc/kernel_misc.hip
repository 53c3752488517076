// Small kernels around the two hot ones: Gram matrices, end-of-iteration scalar
// work (masked SSE from Gram identities, tau draw, metrics), direct fp64 metric
// sums for predict(), layout helpers and the stand-alone distribution hooks.
#include <algorithm>

#include "kernels.h"
#include "many.h"
#include "device_rng.h"

namespace bnmtf {

// ---------------------------------------------------------------------------
// Gram: C = X^T X in fp64.  Block = 256 threads, 128 rows per block staged in LDS;
// thread t owns entries (a, b) with a*KP+b = t + 256*m.  fp64 atomics merge blocks.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_kernel(GramArgs a) {
  __shared__ float tile[128 * 64];
  const int KP = a.KP;
  const int r0 = blockIdx.x * 128;
  const int nr = min(128, a.rows - r0);
  for (int t = threadIdx.x; t < nr * KP; t += 256) tile[t] = a.X[(size_t)r0 * KP + t];
  __syncthreads();
  const int npairs = KP * KP;
  for (int pidx = threadIdx.x; pidx < npairs; pidx += 256) {
    const int ia = pidx / KP, ib = pidx % KP;
    if (ib < ia) continue;                       // symmetric: compute upper triangle
    double s = 0.0;
    for (int r = 0; r < nr; ++r) s = fma((double)tile[r * KP + ia], (double)tile[r * KP + ib], s);
    atomicAdd(a.C64 + pidx, s);
    if (ib != ia) atomicAdd(a.C64 + ib * KP + ia, s);
  }
  if ((int)threadIdx.x < KP) {
    double s = 0.0;
    for (int r = 0; r < nr; ++r) s += (double)tile[r * KP + threadIdx.x];
    atomicAdd(a.colsum + threadIdx.x, s);
    if (a.S2) {
      double s2 = 0.0;
      for (int r = 0; r < nr; ++r) s2 += (double)a.S2[(size_t)(r0 + r) * KP + threadIdx.x];
      atomicAdd(a.colsum2 + threadIdx.x, s2);
    }
  }
}
__global__ void gram_finish_kernel(const double* C64, float* C32, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) C32[t] = (float)C64[t];
}
void launch_gram(const GramArgs& a, hipStream_t st) {
  (void)hipMemsetAsync(a.C64, 0, sizeof(double) * a.KP * a.KP, st);
  (void)hipMemsetAsync(a.colsum, 0, sizeof(double) * a.KP, st);
  if (a.colsum2) (void)hipMemsetAsync(a.colsum2, 0, sizeof(double) * a.KP, st);
  hipLaunchKernelGGL(gram_kernel, dim3((a.rows + 127) / 128), dim3(256), 0, st, a);
  hipLaunchKernelGGL(gram_finish_kernel, dim3((a.KP * a.KP + 255) / 256), dim3(256), 0, st, a.C64, a.C32, a.KP * a.KP);
}

// ---------------------------------------------------------------------------
// end of iteration (bnmf_gibbs_optimised.py:144 tau draw, :199-223 metrics):
//   sum_Omega Rp^2 = <U^T U, V^T V> - sum_miss q^2 ;  sum_Omega Rp = (1^T U).(1^T V) - sum_miss q
//   sum_Omega R Rp = sum (Pv o V)   (Pv = R~^T U only holds observed entries)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void finish_kernel(FinishArgs a) {
  __shared__ double red[16];
  const int KP = a.KP, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (a.copy_dst)      // (round 6: S's sample was a device-to-device copy of 4 KB on the compute stream, 4.8 us of every iteration)
    for (int t = threadIdx.x; t < a.copy_n; t += 1024) a.copy_dst[t] = a.copy_src[t];
  // five sums, one per WAVE (round 5: every wave used to reduce four values over its lanes): waves 0-3 a quarter of <Cr, Cc> each,
  // waves 4-6 one column of the per-block sweep statistics, wave 7 the column-sum product; the other waves have nothing to do
  double s = 0.0;
  if (wave < 4) {
    const int q4 = KP * KP / 4;                                    // (KP = 32 or 64: 4 or 16 terms per lane, all loads in flight)
    double p[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int t = wave * q4 + lane + 64 * u; p[u] = 64 * u < q4 ? a.Cr64[t] * a.Cc64[t] : 0.0; }
#pragma unroll
    for (int u = 0; u < 16; ++u) s += p[u];
  } else if (wave < 7) {
    for (int b = lane; b < a.nstats; b += 64) s += a.stats[(size_t)b * 4 + (wave - 4)];
  } else if (wave == 7) {
    s = lane < KP ? a.sr[lane] * a.sc[lane] : 0.0;
  }
  if (wave < 8) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) red[wave] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot[4];
    tot[0] = (red[0] + red[1]) + (red[2] + red[3]);
    tot[1] = red[4]; tot[2] = red[5]; tot[3] = red[6];
    const double sp1 = red[7];
    const double dot = tot[0];
    double acc[3] = {a.acc[0] + tot[1], a.acc[1] + tot[2], a.acc[2] + tot[3]};
    const double srp = acc[0], sp = sp1 - acc[1], spp = dot - acc[2];
    const double n = a.n_obs;
    const double sse = a.sumR2 - 2.0 * srp + spp;
    const double alpha_s = a.alpha + 0.5 * n, beta_s = a.beta + 0.5 * sse;
    double tau;
    if (a.update == 2) tau = (alpha_s - 1.0) / beta_s;            // gamma_mode (distributions/gamma.py:27-29)
    else if (a.update != 0) tau = alpha_s / beta_s;
    else if (a.gunit) tau = *a.gunit / beta_s;
    else tau = gamma_draw_serial(alpha_s, beta_s, a.it, kStreamTau, a.key0, a.key1);
    *a.tau_d = tau;
    *a.tau_f = (float)tau;
    const double ss_tot = a.sumR2 - a.sumR * a.sumR / n;
    const double cov = srp - a.sumR * sp / n;
    const double vp = spp - sp * sp / n;
    a.rec[0] = tau;
    a.rec[1] = sse / n;
    a.rec[2] = ss_tot != 0.0 ? 1.0 - sse / ss_tot : __longlong_as_double(0x7ff0000000000000LL);
    a.rec[3] = cov / (sqrt(ss_tot) * sqrt(vp));
    a.rec[4] = sse;
  }
}
// VB end of iteration (bnmf_vb_optimised.py:181-187, 213-215): exp_square_diff from Gram identities,
// exptau = alpha_s / beta_s, training-mask metrics, and the O((I+J)K) sums elbo() needs.
__device__ __forceinline__ void vb_finish_body(const VbFinishArgs& a) {
  __shared__ double red[16][16];
  const int KP = a.KP, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // 16 sums: <Cr, Cc>, six columns of the rows-sweep pieces, six of the cols-sweep pieces, and the three K-term sums (colsum .
  // colsum, colsum2 . colsum2, diag(Cr) . diag(Cc)).  Round 5: a sum per WAVE -- waves 0-3 a quarter of <Cr, Cc> each (and, waves
  // 1-3, one K-term sum), waves 4-15 one column of the pieces -- so a wave reduces one or two values over its lanes, not sixteen
  // (16 x 6 fp64 shuffles per wave were most of this kernel's 16 us).  red[w][0 / 1] = the wave's sums.
  double s0 = 0.0, s1 = 0.0;
  if (wave < 4) {
    const int q4 = KP * KP / 4;
    double p[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int t = wave * q4 + lane + 64 * u; p[u] = 64 * u < q4 ? a.Cr64[t] * a.Cc64[t] : 0.0; }
#pragma unroll
    for (int u = 0; u < 16; ++u) s0 += p[u];
    if (wave == 0 && a.extra) for (int b = lane; b < a.n_extra; b += 64) s1 += a.extra[b];
    if (wave >= 1 && lane < KP)
      s1 = wave == 1 ? a.sr[lane] * a.sc[lane] : wave == 2 ? a.s2r[lane] * a.s2c[lane] : a.Cr64[lane * KP + lane] * a.Cc64[lane * KP + lane];
  } else {
    const int c = wave - 4, col = c % 6;
    const double* src = c < 6 ? a.stats_r : a.stats_c;
    const int n = c < 6 ? a.nr : a.nc;
    for (int b = lane; b < n; b += 64) s0 += src[(size_t)b * 8 + col];
    // (round 6: the on-chip cols sweep's per-block sums -- sum P.X', sum_miss q, sum_miss q^2 -- are folded here by waves 4-6;
    // they used to be a launch of their own behind a memset of acc: ~10 us of an iteration that is 125 us for a GDSC-shaped model)
    if (a.sweep_stats && c < 3)
      for (int b = lane; b < a.n_sweep_stats; b += 64) s1 += a.sweep_stats[(size_t)b * 4 + c];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { s0 += __shfl_xor(s0, m, 64); s1 += __shfl_xor(s1, m, 64); }
  if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot[16];
    tot[0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    for (int c = 0; c < 12; ++c) tot[1 + c] = red[4 + c][0];
    tot[13] = red[1][1]; tot[14] = red[2][1]; tot[15] = red[3][1];
    const double dot = tot[0];
    const double* su = &tot[1];
    const double* sv = &tot[7];
    const double sp1 = tot[13], s22 = tot[14], sdd = tot[15];
    const double acc0 = a.acc[0] + (a.sweep_stats ? red[4][1] : 0.0), acc1 = a.acc[1] + (a.sweep_stats ? red[5][1] : 0.0), acc2 = a.acc[2] + (a.sweep_stats ? red[6][1] : 0.0);
    const double srp = acc0, sp = sp1 - acc1, spp = dot - acc2;
    const double n = a.n_obs;
    const double sse = a.sumR2 - 2.0 * srp + spp;
    const double esd = sse + (s22 - sv[4]) - (sdd - sv[5]) + red[0][1];
    const double alpha_s = a.alpha + 0.5 * n, beta_s = a.beta + 0.5 * esd;
    const double exptau = alpha_s / beta_s;
    *a.tau_d = exptau; *a.tau_f = (float)exptau;
    const double ss_tot = a.sumR2 - a.sumR * a.sumR / n;
    const double cov = srp - a.sumR * sp / n, vp = spp - sp * sp / n;
    a.rec[0] = exptau; a.rec[1] = sse / n;
    a.rec[2] = ss_tot != 0.0 ? 1.0 - sse / ss_tot : __longlong_as_double(0x7ff0000000000000LL);
    a.rec[3] = cov / (sqrt(ss_tot) * sqrt(vp));
    a.rec[4] = esd; a.rec[5] = beta_s;
    for (int c = 0; c < 4; ++c) { a.rec[6 + c] = su[c]; a.rec[10 + c] = sv[c]; }
  }
}
__global__ __launch_bounds__(1024) void vb_finish_kernel(VbFinishArgs a) { vb_finish_body(a); }
// list form (many.h): blockIdx.z = model; rec = the run's first record, the iteration's one is `it` records behind it
__global__ __launch_bounds__(1024) void vb_finish_many(const VbFinishArgs* list, int it) {
  VbFinishArgs a = load_pack(list, blockIdx.z);
  a.rec += (size_t)it * 16;
  vb_finish_body(a);
}
void launch_vb_finish(const VbFinishArgs& a, hipStream_t st) {
  if (record_launch((const void*)vb_finish_many, dim3(1), dim3(1024), 0, a)) return;     // (recording: the caller passes the run's FIRST record)
  hipLaunchKernelGGL(vb_finish_kernel, dim3(1), dim3(1024), 0, st, a);
}

void launch_finish(const FinishArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(1024), 0, st, a);
}

// ---------------------------------------------------------------------------
// direct masked metric sums, fp64 arithmetic (predict(), beta_s(), validation).
// Block (32 x 8) covers a 32 x 32 tile of R; A/B tiles in LDS as double.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void metric_kernel(MetricArgs a) {
  __shared__ double At[32 * 65], Bt[32 * 65];
  __shared__ double red[7][256];
  const int K = a.K;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tid = threadIdx.y * 32 + threadIdx.x;
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  const int j = j0 + threadIdx.x;
  // the contraction in chunks of 64 columns (round 6: ranks above 64 -- the tiles hold 64): the element's product and the
  // product of squares accumulate over the chunks, the sums follow behind the last one
  double pr4[4] = {0, 0, 0, 0}, sq4[4] = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 64) {
    const int kc = min(64, K - k0);
    if (k0) __syncthreads();
    for (int t = tid; t < 32 * kc; t += 256) {
      const int r = t / kc, k = t % kc;
      At[r * 65 + k] = (i0 + r < a.I) ? a.A[(size_t)(i0 + r) * K + k0 + k] : 0.0;
      Bt[r * 65 + k] = (j0 + r < a.J) ? a.B[(size_t)(j0 + r) * K + k0 + k] : 0.0;
    }
    __syncthreads();
    for (int rr = 0; rr < 4; ++rr) {
      const int il = threadIdx.y + 8 * rr, i = i0 + il;
      if (i < a.I && j < a.J && a.Mp[(size_t)i * a.J + j]) {
        double pr = pr4[rr], sq = sq4[rr];
        for (int k = 0; k < kc; ++k) {
          const double av = At[il * 65 + k], bv = Bt[threadIdx.x * 65 + k];
          pr = fma(av, bv, pr);
          sq = fma(av * av, bv * bv, sq);
        }
        pr4[rr] = pr; sq4[rr] = sq;
      }
    }
  }
  for (int rr = 0; rr < 4; ++rr) {
    const int il = threadIdx.y + 8 * rr, i = i0 + il;
    if (i < a.I && j < a.J && a.Mp[(size_t)i * a.J + j]) {
      const double pr = pr4[rr], sq = sq4[rr];
      const double r = (double)a.R[(size_t)i * a.J + j];
      s[0] += 1.0; s[1] += r; s[2] += r * r; s[3] += pr; s[4] += pr * pr; s[5] += r * pr; s[6] -= sq;
    }
  }
  if (a.A2) {                                  // second-moment product A2_i . B2_j (VB exp_square_diff)
    __syncthreads();
    for (int t = tid; t < 32 * K; t += 256) {
      const int r = t / K, k = t % K;
      At[r * 65 + k] = (i0 + r < a.I) ? a.A2[(size_t)(i0 + r) * K + k] : 0.0;
      Bt[r * 65 + k] = (j0 + r < a.J) ? a.B2[(size_t)(j0 + r) * K + k] : 0.0;
    }
    __syncthreads();
    for (int rr = 0; rr < 4; ++rr) {
      const int il = threadIdx.y + 8 * rr, i = i0 + il;
      if (i < a.I && j < a.J && a.Mp[(size_t)i * a.J + j]) {
        double pr = 0.0;
        for (int k = 0; k < K; ++k) pr = fma(At[il * 65 + k], Bt[threadIdx.x * 65 + k], pr);
        s[6] += pr;
      }
    }
  }
  for (int m = 0; m < 7; ++m) red[m][tid] = s[m];
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (tid < w) for (int m = 0; m < 7; ++m) red[m][tid] += red[m][tid + w];
    __syncthreads();
  }
  if (tid < 7) atomicAdd(a.out6 + tid, red[tid][0]);
}
void launch_metric_sums(const MetricArgs& a, hipStream_t st) {
  (void)hipMemsetAsync(a.out6, 0, 8 * sizeof(double), st);
  dim3 grid((a.J + 31) / 32, (a.I + 31) / 32), block(32, 8);
  hipLaunchKernelGGL(metric_kernel, grid, block, 0, st, a);
}

__global__ __launch_bounds__(256) void sum_stats_kernel(const double* stats, int nblocks, double* acc) {
  __shared__ double red[4][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double v[3] = {0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nblocks; b += 256)
    for (int t = 0; t < 3; ++t) v[t] += stats[(size_t)b * 4 + t];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v[t] += __shfl_xor(v[t], m, 64);
    if (lane == 0) red[wave][t] = v[t];
  }
  __syncthreads();
  if (threadIdx.x < 3) acc[threadIdx.x] += (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
void launch_sum_stats(const double* stats, int nblocks, double* acc, hipStream_t st) {
  hipLaunchKernelGGL(sum_stats_kernel, dim3(1), dim3(256), 0, st, stats, nblocks, acc);
}

// out[c] += sum_r stats[r * ld + c] for c < ncols (<= 8): the per-unit / per-block partial sums of a sweep folded into one
// short vector, so that a multi-GPU run exchanges a handful of doubles
__global__ __launch_bounds__(256) void sum_cols_kernel(const double* stats, int nrows, int ld, int ncols, double* out) {
  __shared__ double red[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double v[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nrows; b += 256)
#pragma unroll
    for (int t = 0; t < 8; ++t) if (t < ncols) v[t] += stats[(size_t)b * ld + t];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v[t] += __shfl_xor(v[t], m, 64);
    if (lane == 0) red[wave][t] = v[t];
  }
  __syncthreads();
  if ((int)threadIdx.x < ncols) out[threadIdx.x] += (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
void launch_sum_cols(const double* stats, int nrows, int ld, int ncols, double* out, hipStream_t st) {
  hipLaunchKernelGGL(sum_cols_kernel, dim3(1), dim3(256), 0, st, stats, nrows, ld, ncols, out);
}

// posterior-mean accumulation on the device (approx_expectation, bnmf_gibbs_optimised.py:182-187): sum[e] += X[e] in fp64;
// the last element (e == n) takes the scalar *tau.  One pass over a factor: 2-4 MB.
__global__ __launch_bounds__(256) void accumulate_kernel(const float* X, size_t n, double* sum, const double* tau, double* tausum) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n) sum[e] += (double)X[e];
  if (e == 0 && tau) *tausum += *tau;
}
void launch_accumulate(const float* X, size_t n, double* sum, const double* tau, double* tausum, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, n, sum, tau, tausum);
}

// sample hand-off: the W true columns of X [rows][KP] packed into dst [rows][W] (what all_U[it] holds)
__global__ __launch_bounds__(256) void compact_rows_kernel(const float* X, int rows, int W, int KP, float* dst) {
  const size_t n = (size_t)rows * W;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const size_t r = e / W; const int k = (int)(e - r * W);
    dst[e] = X[r * KP + k];
  }
}
void launch_compact_rows(const float* X, int rows, int W, int KP, float* dst, hipStream_t st) {
  if (rows <= 0) return;
  if (W == KP) { (void)hipMemcpyAsync(dst, X, (size_t)rows * W * sizeof(float), hipMemcpyDeviceToDevice, st); return; }
  const size_t n = (size_t)rows * W;
  hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, st, X, rows, W, KP, dst);
}

// ---------------------------------------------------------------------------
__global__ void transpose_kernel(const float* X, int rows, int KP, float* XT, int ldT) {
  __shared__ float t[64][65];
  const int r0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * KP; e += 256) {
    const int r = e / KP, k = e % KP;
    t[r][k] = (r0 + r < rows) ? X[(size_t)(r0 + r) * KP + k] : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * KP; e += 256) {
    const int k = e / 64, r = e % 64;
    if (r0 + r < rows) XT[(size_t)k * ldT + r0 + r] = t[r][k];
  }
}
void launch_transpose(const float* X, int rows, int KP, float* XT, int ldT, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3((rows + 63) / 64), dim3(256), 0, st, X, rows, KP, XT, ldT);
}

// ---------------------------------------------------------------------------
__global__ void tn_sample_kernel(const double* mu, const double* tau, size_t n, uint32_t k0, uint32_t k1,
                                 uint32_t it, uint32_t col, uint32_t elem0, double* out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  out[e] = (double)tn_draw_serial((float)mu[e], (float)tau[e], elem0 + (uint32_t)e, col, it, kStreamHook, k0, k1);
}
void launch_tn_sample(const double* mu, const double* tau, size_t n, uint64_t seed, uint32_t it, uint32_t col,
                      uint32_t elem0, double* out, hipStream_t st) {
  hipLaunchKernelGGL(tn_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mu, tau, n,
                     (uint32_t)seed, (uint32_t)(seed >> 32), it, col, elem0, out);
}
__global__ void tn_moments_kernel(const double* mu, const double* tau, size_t n, double* e, double* v) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  tn_moments(mu[t], tau[t], e + t, v + t);
}
void launch_tn_moments(const double* mu, const double* tau, size_t n, double* e, double* v, hipStream_t st) {
  hipLaunchKernelGGL(tn_moments_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mu, tau, n, e, v);
}
__global__ void gamma_sample_kernel(double alpha, double beta, uint32_t k0, uint32_t k1, uint32_t it, double* out) {
  *out = gamma_draw_serial(alpha, beta, it, kStreamTau, k0, k1);
}
void launch_gamma_sample(double alpha, double beta, uint64_t seed, uint32_t it, double* out, hipStream_t st) {
  hipLaunchKernelGGL(gamma_sample_kernel, dim3(1), dim3(1), 0, st, alpha, beta, (uint32_t)seed, (uint32_t)(seed >> 32), it, out);
}

}  // namespace bnmtf
