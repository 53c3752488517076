// Device-side statement of oracle/rng.py: Philox-4x32-10 and the
// first-accepted-candidate TN(mu,tau) / Gamma samplers.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "tn_moments_coeffs.h"
#include "tn_moments_f32_coeffs.h"

namespace bnmtf {

constexpr float kTnA0 = 0.25f;
constexpr float kTwoPi = 6.283185307179586f;

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
__host__ __device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                             uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// uint32 -> (0,1) from the top 23 bits: (n + 0.5) * 2^-23 with n < 2^23 needs 24 significant bits, so the value is exact
// in fp32 (and the same number in the oracle's fp64), lies strictly inside (0,1) and never rounds to 1.0
__device__ __forceinline__ float u23(uint32_t r) { return ((float)(r >> 9) + 0.5f) * (1.0f / 8388608.0f); }
__host__ __device__ __forceinline__ double u32d(uint32_t r) { return ((double)r + 0.5) * (1.0 / 4294967296.0); }

// Per-draw constants of TN(mu, tau_p) on [0,inf): a = -mu*sqrt(tau_p).
struct TnParams {
  float mu, rt;      // mean, sqrt(precision)
  float a, d, lam;   // lower bound (standardised), lam - a, Robert rate
  bool live, tail;
};

__device__ __forceinline__ TnParams tn_params(float mu, float tau_p) {
  TnParams p;
  p.live = tau_p > 0.0f;
  p.rt = sqrtf(p.live ? tau_p : 1.0f);
  p.mu = mu;
  p.a = -mu * p.rt;
  p.live = p.live && isfinite(p.a);
  p.d = 2.0f / (sqrtf(p.a * p.a + 4.0f) + p.a);
  p.lam = p.a + p.d;
  p.tail = p.a >= kTnA0;
  return p;
}

// Candidate number `cand` of the draw identified by (elem, col, it, stream).
// Returns acceptance; *x receives the candidate value.
__device__ __forceinline__ bool tn_candidate(const TnParams& p, uint32_t elem, uint32_t col, uint32_t it,
                                             uint32_t stream, uint32_t cand, uint32_t k0, uint32_t k1, float* x) {
  const U4 r = philox4x32_10(elem, col, it, stream + 16u * cand, k0, k1);
  const float u1 = u23(r.x), u2 = u23(r.y);
  const float nl = -logf(u1);
  bool acc;
  if (p.tail) {
    const float e = nl / p.lam;
    const float t = e - p.d;
    acc = u2 <= expf(-0.5f * t * t);
    *x = e / p.rt;
  } else {
    const float z = sqrtf(2.0f * nl) * cosf(kTwoPi * u2);
    acc = z >= p.a;
    *x = p.mu + z / p.rt;
  }
  return acc;
}

// acceptance / value of a candidate from two raw 32-bit words (Philox done elsewhere)
__device__ __forceinline__ bool tn_eval_words(const TnParams& p, uint32_t r0, uint32_t r1, float* x) {
  const float u1 = u23(r0), u2 = u23(r1);
  const float nl = -logf(u1);
  if (p.tail) {
    const float e = nl / p.lam, t = e - p.d;
    *x = e / p.rt;
    return u2 <= expf(-0.5f * t * t);
  }
  const float z = sqrtf(2.0f * nl) * cosf(kTwoPi * u2);
  *x = p.mu + z / p.rt;
  return z >= p.a;
}

__device__ __forceinline__ float tn_guard(float x) { return (isfinite(x) && x >= 0.0f) ? x : 0.0f; }

// Serial form (one thread per draw): used by the stand-alone hook and the S sweep.
__device__ __forceinline__ float tn_draw_serial(float mu, float tau_p, uint32_t elem, uint32_t col, uint32_t it,
                                                uint32_t stream, uint32_t k0, uint32_t k1) {
  const TnParams p = tn_params(mu, tau_p);
  if (!p.live) return 0.0f;
  float x = 0.0f;
  for (uint32_t c = 0; c < 4096u; ++c)
    if (tn_candidate(p, elem, col, it, stream, c, k0, k1, &x)) return tn_guard(x);
  return 0.0f;
}

// Gamma(shape, 1) by Marsaglia-Tsang in fp64 (one thread).  The variate depends on (shape, seed, iteration) only --
// not on the rate -- so run() computes it on the host ahead of the iteration and the device divides by the rate.
__host__ __device__ inline double gamma_unit_draw(double shape, uint32_t it, uint32_t stream, uint32_t k0, uint32_t k1) {
  const bool boost = shape < 1.0;
  const double a = boost ? shape + 1.0 : shape;
  const double d = a - 1.0 / 3.0;
  const double c = 1.0 / sqrt(9.0 * d);
  for (uint32_t cand = 0; cand < 4096u; ++cand) {
    const U4 r = philox4x32_10(0u, 0u, it, stream + 16u * cand, k0, k1);
    const double u1 = u32d(r.x), u2 = u32d(r.y), u3 = u32d(r.z), u4 = u32d(r.w);
    const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    double v = 1.0 + c * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    if (log(u3) < 0.5 * x * x + d - d * v + d * log(v)) {
      double g = d * v;
      if (boost) g *= pow(u4, 1.0 / shape);
      return g;
    }
  }
  return shape;
}
__host__ __device__ inline double gamma_draw_serial(double shape, double rate, uint32_t it, uint32_t stream,
                                                    uint32_t k0, uint32_t k1) {
  return gamma_unit_draw(shape, it, stream, k0, k1) / rate;
}

// TN moments in fp64, as truncated_normal_vector.py:53-73 (incl. the exponential
// fall-back for mu < -30 sigma and the negative / non-finite -> 0 guard).
// fp64 helpers of tn_moments: hardware seed + two Newton steps (the seeds are good to ~2^-26), no denormal/overflow fix-ups --
// the callers' arguments are ordinary positive numbers, and a non-finite result is caught by the guards at the end
__device__ __forceinline__ double fast_rcp(double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fast_div(double a, double b) {
  const double r = fast_rcp(b), q = a * r;
  return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double fast_rsqrt(double a) {
  double r = __builtin_amdgcn_rsq(a);
  r = fma(r * fma(-a * r, r, 1.0), 0.5, r);
  r = fma(r * fma(-a * r, r, 1.0), 0.5, r);
  return r;
}

// Expectation and variance of TN(mu, tau_p) on [0, inf) (truncated_normal_vector.py:53-73; oracle tn_expectation /
// tn_variance), in fp64 and branch-free: every VB column update waits for one evaluation of this per unit.
//   lam = pdf(x) / (1 - cdf(x)),  x = -mu sqrt(tau_p)
//   x >= 0:  lam = sqrt(2/pi) / erfcx(x / sqrt 2)
//   x <  0:  lam = pdf(x) / (1 - exp(-x^2/2) erfcx(|x| / sqrt 2) / 2)
// with erfcx from one degree-26 polynomial in s = (t - 3)/(t + 3) (tn_moments_coeffs.h, generated and checked against
// 50-digit values by tools/gen_tn_moments_coeffs.py: 4e-16) and exp from Cody-Waite reduction + degree-13 Taylor.  ~120
// fp64 operations and ~25 VGPRs, against ~400 for exp() + erfc() + three IEEE divisions.  The reference's own fp64
// value has the same cancellation (x^2 in E, x^4 in Var for mu << 0); agreement with it: tests/test_distributions_gpu.py.
__device__ inline void tn_moments(double mu, double tau_p, double* e_out, double* v_out) {
  const double sig = fast_rsqrt(tau_p);
  const double rt = tau_p * sig;
  const double x = -mu * rt;
  const double t = fabs(x) * 0.7071067811865476;
  const double s = fast_div(t - kErfcxA, t + kErfcxA);
  double f = kErfcxP[kErfcxDeg];
#pragma unroll
  for (int k = kErfcxDeg - 1; k >= 0; --k) f = fma(f, s, kErfcxP[k]);       // (1 + 2t) erfcx(t)
  const double w = fma(2.0, t, 1.0);
  const double rf = fast_rcp(f);
  const double lam_pos = 0.7978845608028654 * w * rf;
  // exp(-y), y = x^2/2 (only the x < 0 side uses it)
  const double y = fmin(0.5 * x * x, 745.0);
  const double n = rint(y * 1.4426950408889634);
  const double r = fma(n, 1.9082149292705877e-10, fma(n, 0.6931471803691238, -y));
  double e13 = 1.6059043836821613e-10;                                     // 1/13!
  e13 = fma(e13, r, 2.08767569878681e-09);                                 // 1/12!
  e13 = fma(e13, r, 2.505210838544172e-08);
  e13 = fma(e13, r, 2.755731922398589e-07);
  e13 = fma(e13, r, 2.7557319223985893e-06);
  e13 = fma(e13, r, 2.48015873015873e-05);
  e13 = fma(e13, r, 0.0001984126984126984);
  e13 = fma(e13, r, 0.001388888888888889);
  e13 = fma(e13, r, 0.008333333333333333);
  e13 = fma(e13, r, 0.041666666666666664);
  e13 = fma(e13, r, 0.16666666666666666);
  e13 = fma(e13, r, 0.5);
  e13 = fma(e13, r, 1.0);
  e13 = fma(e13, r, 1.0);
  const double p = ldexp(e13, -(int)n);
  const double ecx = f * fast_rcp(w);
  const double lam_neg = fast_div(0.3989422804014327 * p, fma(-0.5 * p, ecx, 1.0));
  const double lam = x >= 0.0 ? lam_pos : lam_neg;
  double e = fma(sig, lam, mu);
  double v = sig * sig * (1.0 - lam * (lam - x));
  if (mu < -30.0 * sig) {
    e = fast_rcp(fabs(mu) * tau_p);
    v = e * e;
  }
  *e_out = (isfinite(e) && e >= 0.0) ? e : 0.0;
  *v_out = (isfinite(v) && v >= 0.0) ? v : 0.0;
}

// The sweeps' version: fp32 in, fp32 out, ~70 fp32 instructions instead of ~120 fp64 ones.  E = sigma r(x), Var = sigma^2
// h(x) with r = lam - x and h = 1 - lam (lam - x) approximated directly (polynomials in s = (x - 3)/(x + 3) for x > 0;
// exp + an erfcx polynomial for x <= 0, where nothing cancels), so the x^2 / x^4 cancellation that makes the textbook
// formula need fp64 for mu << 0 never happens.  Against 40-digit values: 2.5e-7 (E), 1e-6 (Var) over x in [-40, 30]
// (tools/gen_tn_moments_f32_coeffs.py; tests/test_tn_moments_coeffs_cpu.py) -- the results are stored as fp32 anyway.
__device__ __forceinline__ void tn_moments_f32(float mu, float tau_p, float* e_out, float* v_out) {
  const float sig = __builtin_amdgcn_rsqf(tau_p);
  const float x = -mu * (tau_p * sig);
  const float ax = fabsf(x);
  // x > 0
  const float s = (ax - kTnF32A) * __builtin_amdgcn_rcpf(ax + kTnF32A);
  float pr = kTnF32R[kTnF32RDeg], ph = kTnF32H[kTnF32HDeg];
#pragma unroll
  for (int k = kTnF32RDeg - 1; k >= 0; --k) pr = fmaf(pr, s, kTnF32R[k]);
#pragma unroll
  for (int k = kTnF32HDeg - 1; k >= 0; --k) ph = fmaf(ph, s, kTnF32H[k]);
  const float r_pos = pr * __builtin_amdgcn_rcpf(1.0f + ax);
  const float h_pos = ph * __builtin_amdgcn_rcpf(fmaf(ax, ax, 1.0f));
  // x <= 0
  const float t = ax * 0.70710678f;
  const float st = (t - kTnF32A) * __builtin_amdgcn_rcpf(t + kTnF32A);
  float pe = kTnF32E[kTnF32EDeg];
#pragma unroll
  for (int k = kTnF32EDeg - 1; k >= 0; --k) pe = fmaf(pe, st, kTnF32E[k]);
  const float ecx = pe * __builtin_amdgcn_rcpf(fmaf(2.0f, t, 1.0f));
  const float p = __builtin_amdgcn_exp2f(-0.72134752f * ax * ax);          // exp(-x^2/2)
  const float lam = 0.39894228f * p * __builtin_amdgcn_rcpf(fmaf(-0.5f * p, ecx, 1.0f));
  const float r_neg = lam + ax;
  const float h_neg = fmaf(-lam, r_neg, 1.0f);
  const float r = x > 0.0f ? r_pos : r_neg, h = x > 0.0f ? h_pos : h_neg;
  float e = sig * r;
  float v = sig * sig * h;
  if (mu < -30.0f * sig) {
    e = __builtin_amdgcn_rcpf(fabsf(mu) * tau_p);
    v = e * e;
  }
  *e_out = (isfinite(e) && e >= 0.0f) ? e : 0.0f;
  *v_out = (isfinite(v) && v >= 0.0f) ? v : 0.0f;
}

// The mean alone, one regime at a time (the same operations as tn_moments_f32 above, in the same order): for callers that know
// the sign of x = -mu sqrt(tau) wave-uniformly and have no use for the variance on their critical path (the S chain of the
// variational tri-factorisation, kernel_trivb.hip).  XPOS: x > 0 (mu < 0).
template <bool XPOS>
__device__ __forceinline__ float tn_mean_f32_regime(float mu, float tau_p) {
  const float sig = __builtin_amdgcn_rsqf(tau_p);
  const float x = -mu * (tau_p * sig);
  const float ax = fabsf(x);
  float e;
  if (XPOS) {
    const float s = (ax - kTnF32A) * __builtin_amdgcn_rcpf(ax + kTnF32A);
    float pr = kTnF32R[kTnF32RDeg];
#pragma unroll
    for (int k = kTnF32RDeg - 1; k >= 0; --k) pr = fmaf(pr, s, kTnF32R[k]);
    e = sig * (pr * __builtin_amdgcn_rcpf(1.0f + ax));
    if (mu < -30.0f * sig) e = __builtin_amdgcn_rcpf(fabsf(mu) * tau_p);
  } else {
    const float t = ax * 0.70710678f;
    const float st = (t - kTnF32A) * __builtin_amdgcn_rcpf(t + kTnF32A);
    float pe = kTnF32E[kTnF32EDeg];
#pragma unroll
    for (int k = kTnF32EDeg - 1; k >= 0; --k) pe = fmaf(pe, st, kTnF32E[k]);
    const float ecx = pe * __builtin_amdgcn_rcpf(fmaf(2.0f, t, 1.0f));
    const float p = __builtin_amdgcn_exp2f(-0.72134752f * ax * ax);          // exp(-x^2/2)
    const float lam = 0.39894228f * p * __builtin_amdgcn_rcpf(fmaf(-0.5f * p, ecx, 1.0f));
    e = sig * (lam + ax);
  }
  return (isfinite(e) && e >= 0.0f) ? e : 0.0f;
}

}  // namespace bnmtf
