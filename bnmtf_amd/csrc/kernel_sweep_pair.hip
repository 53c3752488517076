// K3 "pair-step" fast path: the half sweep advances TWO columns (k0 = 2p, k1 = 2p+1) per block barrier.
//
// Per unit the only per-entry register state is q (= U_i . V_j on the missing entries) plus the slot
// addresses packed 2 x 16 bit; the other factor's columns arrive as PAIR panels
// [j] -> (V_j,k0 , V_j,k1) staged in LDS by LDS-DMA, and one conflict-free ds_read_b64 per entry
// feeds both columns:
//   pass 1 (before either draw):   c0 = sum (q - x0 v0) v0 ,  s0 = sum v0^2
//                                  A  = sum (q - x1 v1) v1 ,  B  = sum v0 v1 ,  s1 = sum v1^2
//   column k0:  corr0 = c0 - g0                        -> draw -> d0 = x0' - x0
//   column k1:  corr1 = (A - g1) + d0 (B - C[k0][k1])  -> draw -> d1        (g: Gram terms with the OLD x_k0)
//   pass 2:     q += d0 v0 + d1 v1                     (pair re-read from the still resident panel)
// so nothing but q survives a step, a 1024-thread block (32 units) fits the 128-VGPR budget and every
// CU hosts its 32 units of an 8192-row factor in ONE round; barriers, reductions and staged bytes per
// column are halved with respect to kernel_sweep_fast.hip.  The pre-pass (q from scratch) uses the same
// pair panels.  RNG, candidate order and results are identical to the other sweep kernels.
#include <cstdlib>

#include "kernels.h"
#include "device_rng.h"

namespace bnmtf {

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ float row_sum16(float v) {   // all-reduce inside each 16-lane DPP row
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}
__device__ __forceinline__ float hsum(float v) {        // all-reduce inside each 32-lane half
  v = row_sum16(v);
  return v + __shfl_xor(v, 16, 64);
}
__device__ __forceinline__ double hsum_d(double v) {
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float hbcast(float v, int src, int half) {
  const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
  const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src + 32));
  return half ? a1 : a0;
}
__device__ __forceinline__ uint32_t hbcast_u(uint32_t v, int src, int half) {
  const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)v, src);
  const uint32_t a1 = (uint32_t)__builtin_amdgcn_readlane((int)v, src + 32);
  return half ? a1 : a0;
}

struct TnF { float mu, irt, a, d, ilam; bool live, tail; };
__device__ __forceinline__ TnF tnf_params(float numer, float tau_p) {
  TnF p;
  p.live = tau_p > 0.0f;
  const float tp = p.live ? tau_p : 1.0f;
  p.irt = __builtin_amdgcn_rsqf(tp);             // v_rsq_f32 / v_rcp_f32 / v_sqrt_f32: ~1 ulp, one instruction each
  p.mu = numer * __builtin_amdgcn_rcpf(tp);
  p.a = -p.mu * (tp * p.irt);
  p.live = p.live && isfinite(p.a);
  p.d = 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(p.a, p.a, 4.0f)) + p.a);
  p.ilam = __builtin_amdgcn_rcpf(p.a + p.d);
  p.tail = p.a >= kTnA0;
  return p;
}
__device__ __forceinline__ bool tnf_eval(const TnF& p, uint32_t r0, uint32_t r1, float* x) {
  const float u1 = u24(r0), u2 = u24(r1);
  const float nl = -0.69314718f * __builtin_amdgcn_logf(u1);            // v_log_f32 is log2
  const float e = nl * p.ilam;
  const float t = e - p.d;
  const bool acc_t = u2 <= __builtin_amdgcn_exp2f(-0.72134752f * t * t);   // exp(-t^2/2) via v_exp_f32 (2^x)
  const float z = __builtin_amdgcn_sqrtf(2.0f * nl) * __builtin_amdgcn_cosf(u2);   // v_cos_f32 takes revolutions
  const bool acc_n = z >= p.a;
  *x = p.tail ? e * p.irt : fmaf(z, p.irt, p.mu);
  return p.tail ? acc_t : acc_n;
}

// LDS byte address of pair-panel element `idx` (16-bit, low or high half of `packed`): base + 8*idx in ONE
// v_mad_u32_u16.  Done in asm so that the compiler cannot split off and hoist the loop-invariant 8*idx
// (which would keep two address registers per slot pair alive and defeat the packing).
__device__ __forceinline__ uint32_t pair_addr_lo(uint32_t packed, uint32_t base) {
  uint32_t r;
  asm("v_mad_u32_u16 %0, %1, 8, %2" : "=v"(r) : "v"(packed), "v"(base));
  return r;
}
__device__ __forceinline__ uint32_t pair_addr_hi(uint32_t packed, uint32_t base) {
  uint32_t r;
  asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(packed), "v"(base));
  return r;
}
typedef __attribute__((address_space(3))) const f32x2 lds_cf2;
__device__ __forceinline__ f32x2 lds_read2(uint32_t addr) { return *reinterpret_cast<lds_cf2*>(addr); }
__device__ __forceinline__ uint32_t lds_byte_addr(const float* p) {
  return (uint32_t)(size_t)(__attribute__((address_space(3))) const float*)p;
}

template <int NW>
__device__ __forceinline__ void stage_pair(const float* src, float* dst, int chunks, int wave, int lane) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  for (int c = wave; c < chunks; c += NW)
    __builtin_amdgcn_global_load_lds(src + (size_t)c * 256 + lane * 4, (lds_ptr)(dst + (size_t)c * 256), 16, 0, 0);
}

// One conditional update of column k for this wave's two units (one per half).  Returns the new value.
template <int NX, int MODE, int KH>
__device__ __forceinline__ float draw_column(const SweepArgs& a, int k, float numer, float tau_p, bool valid, size_t gi,
                                             int half, int l5, const uint32_t (&ca)[KH][NX], const uint32_t (&cb)[KH][NX]) {
  float xnew = 0.f;
  if (MODE == kSweepDraw) {
    const TnF tf = tnf_params(numer, tau_p);
    bool done = !tf.live || !valid;
    const bool hi32 = (NX == 2) && k >= 32;             // select, never a runtime array index (that would demote the arrays to scratch)
#pragma unroll
    for (int c = 0; c < KH; ++c) {
      if (c == 0 || __ballot(!done)) {
        float xc;
        const uint32_t wa = hi32 ? ca[c][NX - 1] : ca[c][0], wb = hi32 ? cb[c][NX - 1] : cb[c][0];
        const bool acc = tnf_eval(tf, hbcast_u(wa, k & 31, half), hbcast_u(wb, k & 31, half), &xc);
        if (!done && acc) { xnew = tn_guard(xc); done = true; }
      }
    }
    if (__ballot(!done)) {                               // rare: fresh candidates KH.., 32 per round
      TnParams tp;
      tp.mu = tf.mu; tp.rt = 1.0f / tf.irt; tp.a = tf.a; tp.d = tf.d; tp.lam = tf.a + tf.d; tp.live = tf.live; tp.tail = tf.tail;
      for (uint32_t round = 0; round < 128u && __ballot(!done); ++round) {
        float xr;
        const bool ar = tn_candidate(tp, (uint32_t)gi, (uint32_t)k, a.it, a.stream, (uint32_t)KH + round * 32u + (uint32_t)l5,
                                     a.key0, a.key1, &xr);
        const unsigned long long m = __ballot(ar);
        const uint32_t mh = half ? (uint32_t)(m >> 32) : (uint32_t)m;
        const int first = mh ? __ffs((int)mh) - 1 : 0;
        const float xf = __shfl(xr, half * 32 + first, 64);
        if (!done && mh) { xnew = tn_guard(xf); done = true; }
      }
    }
  } else {
    const float mu = numer / tau_p;
    xnew = fmaxf((valid && tau_p > 0.f && mu > 0.f) ? mu : 0.f, a.min_x);
  }
  return xnew;
}

template <int EM, int NX, int MODE, int NW>
__device__ __forceinline__ void sweep_pair_body(const SweepArgs& a, const FastArgs& f, float* lds) {
  constexpr int KP = NX * 32;
  constexpr int EH = EM / 2;
  constexpr int KH = (NW == 16) ? 2 : 4;              // hoisted candidates per column
  constexpr int CH = (NW == 16) ? 4 : 8;              // slot pairs between scheduling fences (bounds the gathered values in flight)
  const int PW = f.pw;
  float* Cs = lds;
  float* pan = lds + KP * KP;                          // 2 x (2*PW) floats: pair panels, double buffered

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l5 = lane & 31;
  const int pair = blockIdx.x * NW + wave;
  const bool wave_on = pair < f.npairs;
  const uint32_t base = wave_on ? f.pair_base[pair] : 0u;
  const int E = wave_on ? (int)f.pair_E[pair] : 0;
  const int u = wave_on ? f.unit_map[2 * pair + half] : -1;
  const bool valid = u >= 0;
  const size_t gi = (size_t)a.n0 + (valid ? u : 0);
  const int K = a.K;

  float x[NX], p[NX], lam[NX];
#pragma unroll
  for (int nx = 0; nx < NX; ++nx) {
    const int kk = l5 + 32 * nx;
    x[nx] = 0.f; p[nx] = 0.f; lam[nx] = 0.f;
    if (valid) {
      x[nx] = a.Xself[gi * KP + kk];
      for (int s = 0; s < a.split; ++s) p[nx] += a.slabs[((size_t)s * a.n_pad + u) * KP + kk];
      lam[nx] = a.lambda[(size_t)u * KP + kk];
    }
  }
  // slot addresses: two 16-bit inner indices per register (sentinel mz + l5 -> a zero pair on this lane's banks)
  uint32_t off2[EH];
  f32x2 q2[EH];
  const uint32_t sent = (uint32_t)(f.mz + l5);
#pragma unroll
  for (int h = 0; h < EH; ++h) {
    off2[h] = (2 * h < E) ? f.off16[((size_t)(base >> 1) + h) * 64 + lane] : (sent | (sent << 16));
    q2[h] = f32x2{0.f, 0.f};
  }
  for (int t = tid; t < KP * KP; t += NW * 64) Cs[t] = a.C32[t];

  uint32_t ca[KH][NX], cb[KH][NX];
  if (MODE == kSweepDraw) {
#pragma unroll
    for (int c = 0; c < KH; ++c)
#pragma unroll
      for (int nx = 0; nx < NX; ++nx) {
        const U4 r = philox4x32_10((uint32_t)gi, (uint32_t)(l5 + 32 * nx), a.it, a.stream + 16u * c, a.key0, a.key1);
        ca[c][nx] = r.x; cb[c][nx] = r.y;
      }
  }

  const int chunks2 = (2 * PW) / 256;
  const size_t stride = (size_t)f.ld2_o * 2;
  const int npair = (K + 1) / 2;
  auto pairbuf = [&](int i) { return pan + (size_t)(i & 1) * 2 * PW; };

  // ------------------------------------------------------------ pre-pass: q = U_i . V_j
  stage_pair<NW>(f.XoT2, pairbuf(0), chunks2, wave, lane);
  __syncthreads();
  for (int kp = 0; kp < npair; ++kp) {
    // next panel: pair kp+1 of the pre-pass, or pair 0 of the main loop after the last pre-pass step
    const int nxt = (kp + 1 < npair) ? kp + 1 : 0;
    stage_pair<NW>(f.XoT2 + (size_t)nxt * stride, pairbuf(kp + 1), chunks2, wave, lane);
    const uint32_t cur = lds_byte_addr(pairbuf(kp));
    const int k0 = 2 * kp, k1 = 2 * kp + 1;
    const float xs = (NX == 2 && k0 >= 32) ? x[NX - 1] : x[0];
    const float x0 = hbcast(xs, k0 & 31, half), x1 = hbcast(xs, k1 & 31, half);
    const f32x2 x00 = {x0, x0}, x11 = {x1, x1};
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      const f32x2 va = lds_read2(pair_addr_lo(off2[h], cur)), vb = lds_read2(pair_addr_hi(off2[h], cur));
      q2[h] = pfma(f32x2{va.x, vb.x}, x00, q2[h]);
      q2[h] = pfma(f32x2{va.y, vb.y}, x11, q2[h]);
      if ((h % CH) == CH - 1) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // ------------------------------------------------------------ main loop: two columns per step
  const float tau = *a.tau;
  const int pb0 = npair & 1;                           // buffer that received main-loop pair 0
  for (int kp = 0; kp < npair; ++kp) {
    if (kp + 1 < npair) stage_pair<NW>(f.XoT2 + (size_t)(kp + 1) * stride, pairbuf(pb0 + kp + 1), chunks2, wave, lane);
    const uint32_t cur = lds_byte_addr(pairbuf(pb0 + kp));
    const int k0 = 2 * kp, k1 = 2 * kp + 1;
    const bool hi32 = (NX == 2) && k0 >= 32;
    const float xsel = hi32 ? x[NX - 1] : x[0], psel = hi32 ? p[NX - 1] : p[0], lsel = hi32 ? lam[NX - 1] : lam[0];
    const float x0 = hbcast(xsel, k0 & 31, half), x1 = hbcast(xsel, k1 & 31, half);
    f32x2 c0 = {0.f, 0.f}, s0 = {0.f, 0.f}, A = {0.f, 0.f}, B = {0.f, 0.f}, s1 = {0.f, 0.f};
    const f32x2 nx0 = {-x0, -x0}, nx1 = {-x1, -x1};
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      const f32x2 va = lds_read2(pair_addr_lo(off2[h], cur)), vb = lds_read2(pair_addr_hi(off2[h], cur));
      const f32x2 v0 = {va.x, vb.x}, v1 = {va.y, vb.y};
      const f32x2 t0 = pfma(nx0, v0, q2[h]);
      const f32x2 t1 = pfma(nx1, v1, q2[h]);
      c0 = pfma(t0, v0, c0);
      s0 = pfma(v0, v0, s0);
      A = pfma(t1, v1, A);
      B = pfma(v0, v1, B);
      s1 = pfma(v1, v1, s1);
      if ((h % CH) == CH - 1) __builtin_amdgcn_sched_barrier(0);
    }
    float g0 = 0.f, g1 = 0.f;
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) {
      const int kk = l5 + 32 * nx;
      if (kk != k0) g0 = fmaf(x[nx], Cs[k0 * KP + kk], g0);
      if (kk != k1) g1 = fmaf(x[nx], Cs[k1 * KP + kk], g1);
    }
    const float corr0 = hsum((c0.x + c0.y) - g0);
    const float s0r = hsum(s0.x + s0.y);
    const float A1 = hsum((A.x + A.y) - g1);
    const float Br = hsum(B.x + B.y);
    const float s1r = hsum(s1.x + s1.y);
    const float c00 = Cs[k0 * KP + k0], c11 = Cs[k1 * KP + k1], c01 = Cs[k0 * KP + k1];
    const float pk0 = hbcast(psel, k0 & 31, half), pk1 = hbcast(psel, k1 & 31, half);
    const float lk0 = hbcast(lsel, k0 & 31, half), lk1 = hbcast(lsel, k1 & 31, half);
    // ---- column k0
    const float tau_p0 = tau * (c00 - s0r);
    const float numer0 = fmaf(tau, pk0 + corr0, -lk0);
    const float xn0 = draw_column<NX, MODE, KH>(a, k0, numer0, tau_p0, valid, gi, half, l5, ca, cb);
    const float d0 = xn0 - x0;
    // ---- column k1 (absent when K is odd and this is the last pair)
    float d1 = 0.f, xn1 = x1;
    if (k1 < K) {
      const float corr1 = fmaf(d0, Br - c01, A1);
      const float tau_p1 = tau * (c11 - s1r);
      const float numer1 = fmaf(tau, pk1 + corr1, -lk1);
      xn1 = draw_column<NX, MODE, KH>(a, k1, numer1, tau_p1, valid, gi, half, l5, ca, cb);
      d1 = xn1 - x1;
    }
    // ---- pass 2: q += d0 v0 + d1 v1
    const f32x2 d00 = {d0, d0}, d11 = {d1, d1};
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      const f32x2 va = lds_read2(pair_addr_lo(off2[h], cur)), vb = lds_read2(pair_addr_hi(off2[h], cur));
      q2[h] = pfma(d00, f32x2{va.x, vb.x}, q2[h]);
      q2[h] = pfma(d11, f32x2{va.y, vb.y}, q2[h]);
      if ((h % CH) == CH - 1) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) {
      if (l5 + 32 * nx == k0) x[nx] = xn0;
      if (l5 + 32 * nx == k1) x[nx] = xn1;
    }
    __syncthreads();
  }

  // ------------------------------------------------------------ results
#pragma unroll
  for (int nx = 0; nx < NX; ++nx) {
    const int kk = l5 + 32 * nx;
    if (valid && kk < K) a.Xself[gi * KP + kk] = x[nx];
  }
  if (f.stats) {
    double px = 0.0, sq = 0.0, sq2 = 0.0;
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) px += (double)p[nx] * (double)x[nx];
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      const double qa = (double)q2[h].x, qb = (double)q2[h].y;
      sq += qa + qb; sq2 += qa * qa + qb * qb;
    }
    px = hsum_d(px); sq = hsum_d(sq); sq2 = hsum_d(sq2);
    double* red = reinterpret_cast<double*>(pan);
    if (l5 == 0) { red[(wave * 2 + half) * 3 + 0] = valid ? px : 0.0; red[(wave * 2 + half) * 3 + 1] = sq; red[(wave * 2 + half) * 3 + 2] = sq2; }
    __syncthreads();
    if (tid < 3) {
      double s = 0.0;
      for (int w = 0; w < 2 * NW; ++w) s += red[w * 3 + tid];
      f.stats[(size_t)blockIdx.x * 4 + tid] = s;
    }
  }
}

template <int NX, int MODE, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void sweep_pair_kernel(SweepArgs a, FastArgs f) {
  extern __shared__ float lds[];
  const int e0 = (int)f.pair_E[blockIdx.x * NW];      // descending order: first pair of the block is its fullest
  if (e0 <= 8) sweep_pair_body<8, NX, MODE, NW>(a, f, lds);
  else if (e0 <= 16) sweep_pair_body<16, NX, MODE, NW>(a, f, lds);
  else if (e0 <= 24) sweep_pair_body<24, NX, MODE, NW>(a, f, lds);
  else if (e0 <= 32) sweep_pair_body<32, NX, MODE, NW>(a, f, lds);
  else if (e0 <= 40) sweep_pair_body<40, NX, MODE, NW>(a, f, lds);
  else if (NW == 8 && e0 <= 48) sweep_pair_body<(NW == 8 ? 48 : 40), NX, MODE, NW>(a, f, lds);
  else if (NW == 8 && e0 <= 56) sweep_pair_body<(NW == 8 ? 56 : 40), NX, MODE, NW>(a, f, lds);
  else if (f.stats && threadIdx.x < 3) f.stats[(size_t)blockIdx.x * 4 + threadIdx.x] = 0.0;
}

template <int NX, int MODE, int NW>
static void launch_pair_inst(const SweepArgs& a, const FastArgs& f, int nblocks, size_t lds_bytes, hipStream_t st) {
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)sweep_pair_kernel<NX, MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  if (nblocks > 0) hipLaunchKernelGGL((sweep_pair_kernel<NX, MODE, NW>), dim3(nblocks), dim3(NW * 64), lds_bytes, st, a, f);
}

// pairs [0, npairs_hi): slot count > 40 -> 8-wave blocks; the rest -> 16-wave blocks (nw16 = false: all 8-wave)
void launch_sweep_pair(const SweepArgs& a, const FastArgs& f, bool nw16, hipStream_t st) {
  const size_t lds_bytes = sizeof(float) * ((size_t)a.KP * a.KP + 4 * (size_t)f.pw);
  const int nx = a.KP / 32;
  FastArgs lo = f, hi = f;
  int nb_hi, nb_lo;
  if (nw16) {
    hi.npairs = f.npairs_hi;
    lo.pair_E += f.npairs_hi; lo.pair_base += f.npairs_hi; lo.unit_map += 2 * f.npairs_hi; lo.npairs = f.npairs - f.npairs_hi;
    if (lo.stats) lo.stats += (size_t)(f.npairs_hi / 8) * 4;
    nb_hi = hi.npairs / 8; nb_lo = (lo.npairs + 15) / 16;
  } else { nb_hi = (f.npairs + 7) / 8; nb_lo = 0; }
#define BNMTF_L(NXV, MODEV)                                                        \
  do {                                                                             \
    launch_pair_inst<NXV, MODEV, 8>(a, hi, nb_hi, lds_bytes, st);                  \
    launch_pair_inst<NXV, MODEV, 16>(a, lo, nb_lo, lds_bytes, st);                 \
  } while (0)
  if (a.mode == kSweepDraw) { if (nx == 1) BNMTF_L(1, kSweepDraw); else BNMTF_L(2, kSweepDraw); }
  else                      { if (nx == 1) BNMTF_L(1, kSweepMode); else BNMTF_L(2, kSweepMode); }
#undef BNMTF_L
}

}  // namespace bnmtf
