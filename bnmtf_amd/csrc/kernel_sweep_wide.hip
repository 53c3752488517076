// K3 wide path: the register-resident half sweep of kernel_sweep_fast.hip reshaped for FOUR waves per SIMD.
//
// Measured on the 8-wave kernel (make timing): a column step is bound by instruction issue (4 cycles per wave
// instruction, two waves per SIMD) plus phases no second wave is there to fill -- the LDS-DMA issue of the next
// panel (the texture path takes 1 KiB per ~16 cycles per CU and a wave's VMEM issue blocks while it is busy),
// the serial reduction -> sampler chain, the barrier.  Panels cost the same per block however many units share
// them.  So: one 16-wave block per CU (32 units), <= 128 VGPRs per lane:
//  * slots are "balanced" (host: E = ceil(cnt/32) per lane, overflow entries parked in free lanes at the price
//    of a 2-way LDS bank conflict in a few rows) so that <= 28 slots per lane cover a 10 %-missing 8192 row;
//  * the hoisted Philox candidates live in an LDS table instead of registers, and the first kCands candidates
//    of a draw are evaluated by different lanes at once (lane c takes candidate c): one evaluation per column,
//    no data-dependent branches, no skew between waves ahead of the barrier;
//  * sum (q - x_k v) v is taken as sum q v - x_k sum v^2: two packed FMAs per slot pair instead of three.
// Everything else (LDS-DMA double-buffered panels at a compile-time stride, pair-panel pre-pass, three-pass
// column step, RNG stream layout, outputs) is the 8-wave kernel's, so the two produce the same chain.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "sweep_common.h"

namespace bnmtf {

constexpr int kWideNW = 16;                     // waves per block
constexpr int kWideCands = 4;                   // candidates per draw held in the LDS table
constexpr int kWidePanelStride = 9216;          // floats between the two single-column panel buffers (>= pw)

template <int EM, int NX, int MODE>
__device__ __forceinline__ void sweep_wide_body(const SweepArgs& a, const FastArgs& f, float* lds) {
  constexpr int NW = kWideNW;
  constexpr int KP = NX * 32;
  constexpr int EH = EM / 2;
  constexpr int NC = kWideCands;
  static_assert(EM % 2 == 0, "slots are processed in pairs");
  const int PW = f.pw;                      // floats per single-column panel (multiple of 256, <= kWidePanelStride)
  float* Cs = lds;                          // [KP][KP]
  float* pan = lds + KP * KP;               // main loop: buffers at 0 and kWidePanelStride ; pre-pass: 2 x 2*PW
  uint2* tab = reinterpret_cast<uint2*>(pan + kWidePanelStride + PW);   // [2*NW units][KP][NC] raw words; written after the pre-pass
  const uint32_t pan_b = (uint32_t)(uintptr_t)(lds_fp)pan;   // LDS byte address of `pan`

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l5 = lane & 31;
  const int pair = blockIdx.x * NW + wave;
  const bool wave_on = pair < f.npairs;
  const uint32_t base = wave_on ? f.pair_base[pair] : 0u;
  const int E = wave_on ? (int)f.pair_E[pair] : 0;
  const int u = wave_on ? f.unit_map[2 * pair + half] : -1;
  const bool valid = u >= 0;
  const bool valid0 = wave_on && f.unit_map[2 * pair] >= 0, valid1 = wave_on && f.unit_map[2 * pair + 1] >= 0;   // per half, scalar
  const uint32_t gi = (uint32_t)a.n0 + (uint32_t)(valid ? u : 0);
  const int K = a.K;
  const float tau = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.tau)));

#ifdef BNMTF_PHASE_TIMING
  const unsigned long long t_start = tick(0.f);
#endif
  // x = the unit's row of the factor; pl = tau * P - lambda, the part of the conditional's numerator that does not
  // change during the sweep (P = the contraction slabs summed)
  float x[NX], pl[NX];
  auto slab_sum = [&](int nx) {
    float s = 0.f;
    if (valid) s = slab_sum_ordered(a.slabs, a.split, (size_t)a.n_pad * KP, (size_t)u * KP + l5 + 32 * nx);
    return s;
  };
#pragma unroll
  for (int nx = 0; nx < NX; ++nx) {
    const int kk = l5 + 32 * nx;
    x[nx] = valid ? a.Xself[(size_t)gi * KP + kk] : 0.f;
    pl[nx] = valid ? fmaf(tau, slab_sum(nx), -a.lambda[(size_t)u * KP + kk]) : 0.f;
  }
  // slot addresses as LDS BYTE addresses inside panel buffer 0 (sentinel: a zero word on bank l5)
  uint32_t addr[EM];
  f32x2 q2[EH], vp2[EH];                   // slots (2h, 2h+1) share a register pair: packed f32 FMAs
  if (f.off16) {                            // two 16-bit inner indices per word: half the bytes of the slot table
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      const uint32_t sent = (uint32_t)(f.mz + l5);
      const uint32_t w = (2 * h < E) ? f.off16[((size_t)(base >> 1) + h) * 64 + lane] : (sent | (sent << 16));
      addr[2 * h] = pan_b + 4u * (w & 0xFFFFu);
      addr[2 * h + 1] = pan_b + 4u * (w >> 16);
    }
  } else {
#pragma unroll
    for (int s = 0; s < EM; ++s) {
      const uint32_t j = (s < E) ? f.off[((size_t)base + s) * 64 + lane] : (uint32_t)(f.mz + l5);
      addr[s] = pan_b + 4u * j;
    }
  }
#pragma unroll
  for (int h = 0; h < EH; ++h) { q2[h] = f32x2{0.f, 0.f}; vp2[h] = f32x2{0.f, 0.f}; }
  for (int t = tid; t < KP * KP; t += NW * 64) Cs[t] = a.C32[t];

#ifdef BNMTF_PHASE_TIMING
  const unsigned long long t_pre = tick(__builtin_bit_cast(float, addr[EM - 1]) + x[0] + pl[0]);
#endif
  // ------------------------------------------------------------ pre-pass: q = U_i . V_j  (pair panels)
  {
    const int chunks2 = (2 * PW) / 256;
    const uint32_t stride_b = (uint32_t)f.ld2_o * 8u;
    const __amdgpu_buffer_rsrc_t rs2 = panel_rsrc(f.XoT2, (size_t)(KP / 2) * f.ld2_o * 8);
    stage_panel_buf<NW>(rs2, 0u, pan, chunks2, wave, lane * 16);
    __syncthreads();
    const int npair = KP / 2;
    for (int kp = 0; kp < npair; ++kp) {
      if (kp + 1 < npair) stage_panel_buf<NW>(rs2, (uint32_t)(kp + 1) * stride_b, pan + (size_t)((kp + 1) & 1) * 2 * PW, chunks2, wave, lane * 16);
      // element j of a pair panel sits 8 j bytes in: 2 * addr - pan_b (+ the buffer's offset)
      const uint32_t boff = (uint32_t)((kp & 1) * 2 * PW) * 4u - pan_b;
      const int k0 = 2 * kp, k1 = 2 * kp + 1;
      const float xs0 = (NX == 2 && k0 >= 32) ? x[NX - 1] : x[0];
      const float x0 = half_bcast(xs0, k0 & 31, half), x1 = half_bcast(xs0, k1 & 31, half);
      const f32x2 x01 = {x0, x1};
#pragma unroll
      for (int h = 0; h < EH; ++h) {          // (q2[h], vp2[h]) = (even col, odd col) partial sums of slots 2h, 2h+1
        const f32x2 va = *(lds_cf2*)(uintptr_t)(2u * addr[2 * h] + boff);
        const f32x2 vb = *(lds_cf2*)(uintptr_t)(2u * addr[2 * h + 1] + boff);
        q2[h] = pk_fma(va, x01, q2[h]);
        vp2[h] = pk_fma(vb, x01, vp2[h]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int h = 0; h < EH; ++h) { q2[h] = f32x2{q2[h].x + q2[h].y, vp2[h].x + vp2[h].y}; vp2[h] = f32x2{0.f, 0.f}; }
  }

#ifdef BNMTF_PHASE_TIMING
  const unsigned long long t_tab = tick(q2[0].x);
#endif
  // ------------------------------------------------------------ candidate table + first panel
  const int chunks1 = PW / 256;
  const __amdgpu_buffer_rsrc_t rs1 = panel_rsrc(f.XoT, (size_t)KP * f.ldT_o * 4);
  stage_panel_buf<NW>(rs1, 0u, pan, chunks1, wave, lane * 16);
  if (MODE == kSweepDraw) {
    // entry ((unit * KP + col) * NC + c) = words (x, y) of Philox(row, col, it, stream | c << 4): the RNG streams of
    // oracle/rng.py, produced by whichever thread the flat index falls on
    constexpr int kEntries = 2 * NW * KP * NC;
    for (int e = tid; e < kEntries; e += NW * 64) {
      const int c = e % NC, col = (e / NC) % KP, un = e / (NC * KP);
      const int pr = blockIdx.x * NW + (un >> 1);
      const int uu = pr < f.npairs ? f.unit_map[2 * pr + (un & 1)] : -1;
      const U4 r = philox4x32_10((uint32_t)(a.n0 + (uu >= 0 ? uu : 0)), (uint32_t)col, a.it, a.stream + 16u * (uint32_t)c, a.key0, a.key1);
      tab[e] = uint2{r.x, r.y};
    }
  }
  __syncthreads();
  float dprev = 0.f;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) const u32x2 lds_cu2;
  const uint32_t mytab_b = (uint32_t)(uintptr_t)(lds_fp)(pan + kWidePanelStride + PW) + 8u * (uint32_t)((2 * wave + half) * KP * NC + (l5 & (NC - 1)));

  // One column.  BUF (which panel buffer holds column k) and HI (k >= 32: which register of x/p/lam owns column k)
  // are compile-time, so the buffer offset is a ds_read immediate.
#ifdef BNMTF_PHASE_TIMING
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = tick(q2[0].x);
  const unsigned long long t_main = tlast;
#endif
  auto column = [&](auto buf_c, auto hi_c, int k) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr int HI = decltype(hi_c)::value;
    if (k + 1 < K) stage_panel_buf<NW>(rs1, (uint32_t)(k + 1) * (uint32_t)f.ldT_o * 4u, pan + (size_t)(1 - BUF) * kWidePanelStride, chunks1, wave, lane * 16);
    const float xk = half_bcast(x[HI], k & 31, half);
    // (A) column k-1's update, from the registers that still hold v_{k-1}
    const f32x2 dp2 = {dprev, dprev};
#pragma unroll
    for (int h = 0; h < EH; ++h) q2[h] = pk_fma(dp2, vp2[h], q2[h]);
    __builtin_amdgcn_sched_barrier(0);
    TICK(0, q2[0].x);
    // (B) gather v_k into those registers: address register + immediate, no VALU
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      asm volatile("" : "+v"(addr[2 * h]), "+v"(addr[2 * h + 1]));   // opaque: keeps addr + const from being hoisted into registers
      vp2[h].x = *(lds_cf*)(uintptr_t)(addr[2 * h] + (uint32_t)(BUF * kWidePanelStride * 4));
      vp2[h].y = *(lds_cf*)(uintptr_t)(addr[2 * h + 1] + (uint32_t)(BUF * kWidePanelStride * 4));
    }
    // (C) sum q v  and  sum v^2 ;  sum (q - x_k v) v = sum q v - x_k sum v^2
    f32x2 qv2[2] = {{0.f, 0.f}, {0.f, 0.f}}, vv2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      qv2[h & 1] = pk_fma(q2[h], vp2[h], qv2[h & 1]);
      vv2[h & 1] = pk_fma(vp2[h], vp2[h], vv2[h & 1]);
    }
    float asq_t = (vv2[0].x + vv2[0].y) + (vv2[1].x + vv2[1].y);
    float corr_t = fmaf(-xk, asq_t, (qv2[0].x + qv2[0].y) + (qv2[1].x + qv2[1].y));
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) corr_t = fmaf(-x[nx], Cs[k * KP + l5 + 32 * nx], corr_t);   // all l: the l = k term is put back below
    TICK(1, corr_t + asq_t);
    u32x2 cw = {0u, 0u};
    if (MODE == kSweepDraw) cw = *(lds_cu2*)(uintptr_t)(mytab_b + (uint32_t)(k * NC * 8));
    corr_t = half_sum_upper(corr_t);     // from here on the unit's scalars are right in lanes 16-31 of its half only
    asq_t = half_sum_upper(asq_t);
    const float ckk = Cs[k * KP + k];
    const float tau_p = tau * (ckk - asq_t);
    const float numer = fmaf(tau, fmaf(xk, ckk, corr_t), half_bcast(pl[HI], k & 31, half));
    float xnew = 0.f;
    TICK(2, numer + tau_p);
    if (MODE == kSweepDraw) {
      const TnFast tf = tn_fast_params(numer, tau_p);
      // lanes 16 .. 16+NC-1 of each half evaluate candidates 0 .. NC-1 of the batch; the first accepted one is the draw.
      // Batch 0 comes from the table; a wave in which some unit rejected a whole batch (rare) computes the next NC
      // candidates (numbers NC, NC+1, ... : the oracle's candidate sequence) and runs the same code again.
      // Everything that is one value per unit is kept per HALF in scalar registers (ballots, s_ff1, v_readlane), not as
      // per-lane booleans: the selection costs scalar instructions instead of vector issue slots.
      constexpr unsigned long long kCandMask = (unsigned long long)((1u << NC) - 1u) << 16 | (unsigned long long)((1u << NC) - 1u) << 48;
      const unsigned long long mlive = __ballot(tf.live);                 // lanes 16 / 48 speak for the two units
      bool need0 = valid0 && ((mlive >> 16) & 1ull), need1 = valid1 && ((mlive >> 48) & 1ull);
      float xs0 = 0.f, xs1 = 0.f;
      for (uint32_t cbase = 0;;) {
        float xc;
        const bool acc = tn_eval_fast(tf, cw.x, cw.y, &xc);
        xc = tn_guard(xc);
        const unsigned long long m = __ballot(acc) & kCandMask;
        const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
        const float c0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xc), m0 ? __ffs((int)m0) - 1 : 0));
        const float c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xc), 32 + (m1 ? __ffs((int)m1) - 1 : 0)));
        if (need0 && m0) { xs0 = c0; need0 = false; }
        if (need1 && m1) { xs1 = c1; need1 = false; }
        cbase += NC;
        if (!(need0 || need1) || cbase >= 4096u) break;
        uint32_t row = gi;
        asm volatile("" : "+v"(row));                            // opaque: nothing of this Philox call is hoisted out of the column loop
        const U4 r = philox4x32_10(row, (uint32_t)k, a.it, a.stream + 16u * (cbase + (uint32_t)(l5 & (NC - 1))), a.key0, a.key1);
        cw = u32x2{r.x, r.y};
      }
      xnew = half ? xs1 : xs0;
    } else {
      const float mu = numer / tau_p;
      const float xm = fmaxf((valid && tau_p > 0.f && mu > 0.f) ? mu : 0.f, a.min_x);
      xnew = half_bcast(xm, 16, half);
    }
    dprev = xnew - xk;
    TICK(3, dprev);
    if (l5 + 32 * HI == k) x[HI] = xnew;
#ifdef BNMTF_EXPERIMENT_NO_BARRIER
    __builtin_amdgcn_s_waitcnt(0);    // experiment only (wrong results): how fast would the column loop run without the per-column barrier?
#else
    __syncthreads();
#endif
    TICK(4, dprev);
  };
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  using chi = std::integral_constant<int, NX - 1>;
  const int K0 = K < 32 ? K : 32;
  for (int k = 0; k < K0; k += 2) {
    column(c0{}, c0{}, k);
    if (k + 1 < K0) column(c1{}, c0{}, k + 1);
  }
  if (NX == 2) {                            // here K0 == 32: column 32 is in buffer 0 again
    for (int k = 32; k < K; k += 2) {
      column(c0{}, chi{}, k);
      if (k + 1 < K) column(c1{}, chi{}, k + 1);
    }
  }

#ifdef BNMTF_PHASE_TIMING
  if (blockIdx.x % 61 == 0 && (tid & 255) == 0)
    printf("block %d wave %d EM %d: prologue %llu prepass %llu table %llu | A %llu  BC %llu  reduce %llu  sampler %llu  barrier %llu  (cycles, %d columns)\n",
           (int)blockIdx.x, wave, EM, t_pre - t_start, t_tab - t_pre, t_main - t_tab, ph[0], ph[1], ph[2], ph[3], ph[4], K);
#endif
  // ------------------------------------------------------------ results
#pragma unroll
  for (int nx = 0; nx < NX; ++nx) {
    const int kk = l5 + 32 * nx;
    if (valid && kk < K) a.Xself[(size_t)gi * KP + kk] = x[nx];
  }
  if (f.stats) {                      // per-block partial sums -> slab, summed by finish_kernel
    double px = 0.0, sq = 0.0, sq2 = 0.0;
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) px += (double)slab_sum(nx) * (double)x[nx];
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      const double qa = (double)fmaf(dprev, vp2[h].x, q2[h].x), qb = (double)fmaf(dprev, vp2[h].y, q2[h].y);
      sq += qa + qb; sq2 += qa * qa + qb * qb;
    }
    px = half_sum_d(px); sq = half_sum_d(sq); sq2 = half_sum_d(sq2);
    double* red = reinterpret_cast<double*>(pan);      // panels are dead: reuse
    if (l5 == 0) { red[(wave * 2 + half) * 3 + 0] = valid ? px : 0.0; red[(wave * 2 + half) * 3 + 1] = sq; red[(wave * 2 + half) * 3 + 2] = sq2; }
    __syncthreads();
    if (tid < 3) {
      double s = 0.0;
      for (int w = 0; w < 2 * NW; ++w) s += red[w * 3 + tid];
      f.stats[(size_t)blockIdx.x * 4 + tid] = s;
    }
  }
}

// One launch covers every block.
template <int NX, int MODE>
__global__ __launch_bounds__(kWideNW * 64, 1) void sweep_wide_kernel(SweepArgs a, FastArgs f) {
  extern __shared__ float lds[];
  // the slot class is picked per WAVE (every body runs the same barrier sequence), so a block mixes full and light pairs
  const int pr = blockIdx.x * kWideNW + (int)(threadIdx.x >> 6);
  const int e0 = __builtin_amdgcn_readfirstlane(pr < f.npairs ? (int)f.pair_E[pr] : 0);
  if (e0 <= 8) sweep_wide_body<8, NX, MODE>(a, f, lds);
  else if (e0 <= 14) sweep_wide_body<14, NX, MODE>(a, f, lds);
  else if (e0 <= 20) sweep_wide_body<20, NX, MODE>(a, f, lds);
  else if (e0 <= 24) sweep_wide_body<24, NX, MODE>(a, f, lds);
  else if (e0 <= 26) sweep_wide_body<26, NX, MODE>(a, f, lds);
  else if (e0 <= 28) sweep_wide_body<28, NX, MODE>(a, f, lds);
  else sweep_wide_body<kWideMaxSlots, NX, MODE>(a, f, lds);      // host guarantees e0 <= kWideMaxSlots
}

// C | max(pre-pass: two pair panels = 4 pw , main loop: two single panels + the candidate table)
size_t sweep_wide_lds_bytes(int KP, int pw) {
  const size_t main_loop = (size_t)kWidePanelStride + pw + (size_t)2 * kWideNW * KP * kWideCands * 2;
  return sizeof(float) * ((size_t)KP * KP + std::max<size_t>(4 * (size_t)pw, main_loop));
}

bool sweep_wide_supported(int KP, int pw) { return pw <= kWidePanelStride && sweep_wide_lds_bytes(KP, pw) <= 160 * 1024; }

template <int NX, int MODE>
static void launch_inst(const SweepArgs& a, const FastArgs& f, int nblocks, size_t lds_bytes, hipStream_t st) {
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)sweep_wide_kernel<NX, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  if (nblocks > 0) hipLaunchKernelGGL((sweep_wide_kernel<NX, MODE>), dim3(nblocks), dim3(kWideNW * 64), lds_bytes, st, a, f);
}

// f describes the pairs this launch owns (f.npairs of them, at most kWideMaxSlots slots each), 16 pairs per block.
void launch_sweep_wide(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  const size_t lds_bytes = sweep_wide_lds_bytes(a.KP, f.pw);
  const int nblocks = (f.npairs + kWideNW - 1) / kWideNW;
  const int nx = a.KP / 32;
  if (a.mode == kSweepDraw) { if (nx == 1) launch_inst<1, kSweepDraw>(a, f, nblocks, lds_bytes, st); else launch_inst<2, kSweepDraw>(a, f, nblocks, lds_bytes, st); }
  else                      { if (nx == 1) launch_inst<1, kSweepMode>(a, f, nblocks, lds_bytes, st); else launch_inst<2, kSweepMode>(a, f, nblocks, lds_bytes, st); }
}

}  // namespace bnmtf
