// K3, 2/4/8-wave blocks: the half sweep of kernel_sweep.hip with everything that is touched K times per row kept on
// chip.  This is the shape for shards of a multi-GPU run, small problems and masks with more than 32 slots per lane (up
// to 56: two waves per SIMD, 256 VGPRs); large single-GPU problems take the 16-wave kernel (kernel_sweep_wide.hip), whose
// operation order this kernel follows so that the two draw exactly the same chain.
//
//  * a unit (row of the factor being updated) owns one 32-lane half wave; its missing entries sit in "slots": slot s of
//    lane l holds an entry whose inner index j has j mod 32 == l (bank-conflict-free ds_read_b32 gathers; the host parks
//    the entries of over-full residue classes in free lanes, api.hip build_dir); sentinel slots read a per-lane zero word.
//  * q_ij (= U_i . V_j on the missing entries), the slot byte addresses and the previous column's gathered values live
//    in registers (EM slots per lane, template), two slots per register pair for packed-f32 FMAs.
//  * the other factor's column k ("panel", m floats) is staged in LDS once per block and k by LDS-DMA through a buffer
//    descriptor, double buffered at a compile-time stride (a gather is one ds_read_b32 offset:imm); C = V^T V sits in LDS.
//  * q is rebuilt each sweep by a pre-pass over pair panels (ds_read_b64: two columns per gather, same slots), which is
//    what makes the kernel independent of how the other direction ordered its entries (and of the GPU count).
//  * the K draws per unit are sequential, so the Philox work is hoisted: lane l pre-computes the first four candidates
//    of columns l and l+32; step k broadcasts them.  Four rejections in a row (rare) fall back to 32 fresh candidates
//    per round, evaluated with the same one-instruction transcendental forms.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "sweep_common.h"

namespace bnmtf {

constexpr int kPanelStride = 9216;              // floats between the two single-column panel buffers (>= pw)

template <int EM, int NX, int MODE, int NW, int DW>
__device__ __forceinline__ void sweep_fast_body(const SweepArgs& a, const FastArgs& f, float* lds) {
  constexpr int KP = NX * 32;
  constexpr int EH = EM / 2;
  constexpr int kHoist = 4;
  static_assert(EM % 2 == 0, "slots are processed in pairs");
  const int PW = f.pw;                      // floats per single-column panel (multiple of 256, <= kPanelStride)
  float* Cs = lds;                          // [KP][KP]
  float* pan = lds + KP * KP;               // main loop: buffers at 0 and kPanelStride ; pre-pass: 2 x 2*PW
  const uint32_t pan_b = (uint32_t)(uintptr_t)(lds_fp)pan;   // LDS byte address of `pan`

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l5 = lane & 31;
  if (DW && wave == NW) {
    // The staging wave of a small block (DW = 1: 2- or 4-wave blocks, i.e. few units per CU: shards of a multi-GPU run).
    // It owns no units and issues every LDS-DMA piece of every panel; the unit waves never stall on VMEM issue (the
    // texture path takes 1 KiB per ~16 cycles, and with one unit wave per SIMD nothing hides that stall).  Same barrier
    // sequence as the unit waves.
    const int chunks2 = (2 * PW) / 256, chunks1 = PW / 256;
    const uint32_t stride_b = (uint32_t)f.ld2_o * 8u;
    const __amdgpu_buffer_rsrc_t rs2 = panel_rsrc(f.XoT2, (size_t)(KP / 2) * f.ld2_o * 8);
    const __amdgpu_buffer_rsrc_t rs1 = panel_rsrc(f.XoT, (size_t)KP * f.ldT_o * 4);
    stage_panel_buf<1>(rs2, 0u, pan, chunks2, 0, lane * 16);
    __syncthreads();
    for (int kp = 0; kp < KP / 2; ++kp) {
      if (kp + 1 < KP / 2) stage_panel_buf<1>(rs2, (uint32_t)(kp + 1) * stride_b, pan + (size_t)((kp + 1) & 1) * 2 * PW, chunks2, 0, lane * 16);
      __syncthreads();
    }
    stage_panel_buf<1>(rs1, 0u, pan, chunks1, 0, lane * 16);
    __syncthreads();
    for (int k = 0; k < a.K; ++k) {
      if (k + 1 < a.K) stage_panel_buf<1>(rs1, (uint32_t)(k + 1) * (uint32_t)f.ldT_o * 4u, pan + (size_t)((k + 1) & 1) * kPanelStride, chunks1, 0, lane * 16);
      __syncthreads();
    }
    if (f.stats) __syncthreads();
    return;
  }
  const int pair = blockIdx.x * NW + wave;
  const bool wave_on = pair < f.npairs;
  const uint32_t base = wave_on ? f.pair_base[pair] : 0u;
  const int E = wave_on ? (int)f.pair_E[pair] : 0;
  const int u = wave_on ? f.unit_map[2 * pair + half] : -1;
  const bool valid = u >= 0;
  const size_t gi = (size_t)a.n0 + (valid ? u : 0);
  const int K = a.K;

#ifdef BNMTF_PHASE_TIMING
  const unsigned long long t_start = tick(0.f);
#endif
  const float tau = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.tau)));
  // x = the unit's row of the factor; p = the contraction slabs summed; pl = tau * p - lambda (constant during the sweep)
  float x[NX], p[NX], pl[NX];
#pragma unroll
  for (int nx = 0; nx < NX; ++nx) {
    const int kk = l5 + 32 * nx;
    x[nx] = 0.f; p[nx] = 0.f; pl[nx] = 0.f;
    if (valid) {
      x[nx] = a.Xself[gi * KP + kk];
      p[nx] = slab_sum_ordered(a.slabs, a.split, (size_t)a.n_pad * KP, (size_t)u * KP + kk);
      pl[nx] = fmaf(tau, p[nx], -a.lambda[(size_t)u * KP + kk]);
    }
  }
  // slot addresses as LDS BYTE addresses inside panel buffer 0 (sentinel: a zero word on bank l5)
  uint32_t addr[EM];
  f32x2 q2[EH], vp2[EH];                   // slots (2h, 2h+1) share a register pair: packed f32 FMAs
#pragma unroll
  for (int s = 0; s < EM; ++s) {
    const uint32_t j = (s < E) ? f.off[((size_t)base + s) * 64 + lane] : (uint32_t)(f.mz + l5);
    addr[s] = pan_b + 4u * j;
  }
#pragma unroll
  for (int h = 0; h < EH; ++h) { q2[h] = f32x2{0.f, 0.f}; vp2[h] = f32x2{0.f, 0.f}; }
  for (int t = tid; t < KP * KP; t += NW * 64) Cs[t] = a.C32[t];

  // hoisted Philox: candidates 0..kHoist-1 of columns l5 (+32)
  uint32_t ca[kHoist][NX], cb[kHoist][NX];
  if (MODE == kSweepDraw) {
#pragma unroll
    for (int c = 0; c < kHoist; ++c)
#pragma unroll
      for (int nx = 0; nx < NX; ++nx) {
        const U4 r = philox4x32_10((uint32_t)gi, (uint32_t)(l5 + 32 * nx), a.it, a.stream + 16u * c, a.key0, a.key1);
        ca[c][nx] = r.x; cb[c][nx] = r.y;
      }
  }

#ifdef BNMTF_PHASE_TIMING
  const unsigned long long t_pre = tick(__builtin_bit_cast(float, ca[kHoist - 1][0] ^ addr[EM - 1]) + x[0] + p[0] + pl[0]);
#endif
  // ------------------------------------------------------------ pre-pass: q = U_i . V_j  (pair panels)
  {
    const int chunks2 = (2 * PW) / 256;
    const uint32_t stride_b = (uint32_t)f.ld2_o * 8u;
    const __amdgpu_buffer_rsrc_t rs2 = panel_rsrc(f.XoT2, (size_t)(KP / 2) * f.ld2_o * 8);
    if (!DW) stage_panel_buf<NW>(rs2, 0u, pan, chunks2, wave, lane * 16);
    __syncthreads();
    const int npair = KP / 2;
    for (int kp = 0; kp < npair; ++kp) {
      if (!DW && kp + 1 < npair) stage_panel_buf<NW>(rs2, (uint32_t)(kp + 1) * stride_b, pan + (size_t)((kp + 1) & 1) * 2 * PW, chunks2, wave, lane * 16);
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

  // ------------------------------------------------------------ the K sequential columns
  const int chunks1 = PW / 256;
  const __amdgpu_buffer_rsrc_t rs1 = panel_rsrc(f.XoT, (size_t)KP * f.ldT_o * 4);
  if (!DW) stage_panel_buf<NW>(rs1, 0u, pan, chunks1, wave, lane * 16);
  __syncthreads();
  float dprev = 0.f;

  // One column.  BUF (which panel buffer holds column k) and HI (k >= 32: which register of x/p/lam/ca/cb
  // owns column k) are compile-time, so the buffer offset is a ds_read immediate.
#ifdef BNMTF_PHASE_TIMING
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = tick(q2[0].x);
  const unsigned long long t_main = tlast;
#endif
  auto column = [&](auto buf_c, auto hi_c, int k) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr int HI = decltype(hi_c)::value;
    if (!DW && k + 1 < K) stage_panel_buf<NW>(rs1, (uint32_t)(k + 1) * (uint32_t)f.ldT_o * 4u, pan + (size_t)(1 - BUF) * kPanelStride, chunks1, wave, lane * 16);
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
      vp2[h].x = *(lds_cf*)(uintptr_t)(addr[2 * h] + (uint32_t)(BUF * kPanelStride * 4));
      vp2[h].y = *(lds_cf*)(uintptr_t)(addr[2 * h + 1] + (uint32_t)(BUF * kPanelStride * 4));
    }
    // (C) sum q v  and  sum v^2 ;  sum (q - x_k v) v = sum q v - x_k sum v^2.  Same operation order as the 16-wave
    // kernel (kernel_sweep_wide.hip), so a shard of a multi-GPU run draws exactly what the single-GPU run draws.
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
    corr_t = half_sum(corr_t);
    asq_t = half_sum(asq_t);
    const float ckk = Cs[k * KP + k];
    const float tau_p = tau * (ckk - asq_t);
    const float numer = fmaf(tau, fmaf(xk, ckk, corr_t), half_bcast(pl[HI], k & 31, half));
    float xnew = 0.f;
    TICK(2, numer + tau_p);
    if (MODE == kSweepDraw) {
      const TnFast tf = tn_fast_params(numer, tau_p);
      bool done = !tf.live || !valid;
#pragma unroll
      for (int c = 0; c < kHoist; ++c) {
        if (c == 0 || __ballot(!done)) {                      // wave-uniform: later candidates only when someone still needs one
          float xc;
          const bool acc = tn_eval_fast(tf, half_bcast_u(ca[c][HI], k & 31, half), half_bcast_u(cb[c][HI], k & 31, half), &xc);
          if (!done && acc) { xnew = tn_guard(xc); done = true; }
        }
      }
      for (uint32_t round = 0; round < 128u && __ballot(!done); ++round) {   // rare: fresh candidates kHoist.. : 32 per round
        uint32_t row = (uint32_t)gi;
        asm volatile("" : "+v"(row));                            // opaque: nothing of this Philox call is hoisted out of the column loop
        const U4 r = philox4x32_10(row, (uint32_t)k, a.it, a.stream + 16u * ((uint32_t)kHoist + round * 32u + (uint32_t)l5), a.key0, a.key1);
        float xr;
        const bool ar = tn_eval_fast(tf, r.x, r.y, &xr);
        const unsigned long long m = __ballot(ar);
        const uint32_t mh = half ? (uint32_t)(m >> 32) : (uint32_t)m;
        const int first = mh ? __ffs((int)mh) - 1 : 0;
        const float xf = __shfl(xr, half * 32 + first, 64);
        if (!done && mh) { xnew = tn_guard(xf); done = true; }
      }
    } else {
      const float mu = numer / tau_p;
      xnew = fmaxf((valid && tau_p > 0.f && mu > 0.f) ? mu : 0.f, a.min_x);
    }
    dprev = xnew - xk;
    TICK(3, dprev);
    if (l5 + 32 * HI == k) x[HI] = xnew;
    __syncthreads();
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
  if (blockIdx.x % 97 == 0 && tid == 0)
    printf("block %d EM %d: prologue %llu prepass %llu | A %llu  BC %llu  reduce %llu  sampler %llu  barrier %llu  (cycles, %d columns)\n",
           (int)blockIdx.x, EM, t_pre - t_start, t_main - t_pre, ph[0], ph[1], ph[2], ph[3], ph[4], K);
#endif
  // ------------------------------------------------------------ results
#pragma unroll
  for (int nx = 0; nx < NX; ++nx) {
    const int kk = l5 + 32 * nx;
    if (valid && kk < K) a.Xself[gi * KP + kk] = x[nx];
  }
  if (f.stats) {                      // per-block partial sums -> slab, summed by finish_kernel
    double px = 0.0, sq = 0.0, sq2 = 0.0;
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) px += (double)p[nx] * (double)x[nx];
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

// One launch covers every block; a block's slot class (template EM) is the smallest class that
// holds its fullest pair (units are sorted by slot count, so blocks are homogeneous).
template <int NX, int MODE, int NW, int DW>
__global__ __launch_bounds__((NW + DW) * 64, (DW ? 1 : 2)) void sweep_fast_kernel(SweepArgs a, FastArgs f) {
  extern __shared__ float lds[];
  const int e0 = (int)f.pair_E[blockIdx.x * NW];      // descending order: first pair of the block is its fullest
  if (e0 <= 8) sweep_fast_body<8, NX, MODE, NW, DW>(a, f, lds);
  else if (e0 <= 16) sweep_fast_body<16, NX, MODE, NW, DW>(a, f, lds);
  else if (e0 <= 24) sweep_fast_body<24, NX, MODE, NW, DW>(a, f, lds);
  else if (e0 <= 32) sweep_fast_body<32, NX, MODE, NW, DW>(a, f, lds);
  else if (e0 <= 40) sweep_fast_body<40, NX, MODE, NW, DW>(a, f, lds);
  else if (e0 <= 48) sweep_fast_body<48, NX, MODE, NW, DW>(a, f, lds);
  else if (e0 <= kFastMaxSlots) sweep_fast_body<kFastMaxSlots, NX, MODE, NW, DW>(a, f, lds);
  else if (f.stats && threadIdx.x < 3) f.stats[(size_t)blockIdx.x * 4 + threadIdx.x] = 0.0;   // generic kernel owns these units
}

// C | max(pre-pass: two pair panels = 4 pw , main loop: two single panels kPanelStride apart)
size_t sweep_fast_lds_bytes(int KP, int pw) {
  const size_t panels = std::max<size_t>(4 * (size_t)pw, (size_t)kPanelStride + pw);
  return sizeof(float) * ((size_t)KP * KP + panels);
}

bool sweep_fast_supported(int KP, int pw) { return pw <= kPanelStride && sweep_fast_lds_bytes(KP, pw) <= 160 * 1024; }

template <int NX, int MODE, int NW, int DW>
static void launch_inst(const SweepArgs& a, const FastArgs& f, int nblocks, size_t lds_bytes, hipStream_t st) {
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)sweep_fast_kernel<NX, MODE, NW, DW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  if (nblocks > 0) hipLaunchKernelGGL((sweep_fast_kernel<NX, MODE, NW, DW>), dim3(nblocks), dim3((NW + DW) * 64), lds_bytes, st, a, f);
}

// Blocks of NW waves (2*NW units).  8 waves is the throughput shape; when a rank owns few units (row/column
// shards of a multi-GPU run, small problems) 4- or 2-wave blocks keep every CU busy instead of a few.
void launch_sweep_fast(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  const size_t lds_bytes = sweep_fast_lds_bytes(a.KP, f.pw);
  const int nx = a.KP / 32;
  const int nw = f.nw;
  static const bool dw = getenv("BNMTF_NO_STAGING_WAVE") == nullptr;    // small blocks get a staging wave (see sweep_fast_body)
#define BNMTF_L(NXV, MODEV)                                                                       \
  do {                                                                                            \
    if (nw == 2 && dw) launch_inst<NXV, MODEV, 2, 1>(a, f, (f.npairs + 1) / 2, lds_bytes, st);     \
    else if (nw == 2) launch_inst<NXV, MODEV, 2, 0>(a, f, (f.npairs + 1) / 2, lds_bytes, st);      \
    else if (nw == 4 && dw) launch_inst<NXV, MODEV, 4, 1>(a, f, (f.npairs + 3) / 4, lds_bytes, st); \
    else if (nw == 4) launch_inst<NXV, MODEV, 4, 0>(a, f, (f.npairs + 3) / 4, lds_bytes, st);      \
    else launch_inst<NXV, MODEV, 8, 0>(a, f, (f.npairs + 7) / 8, lds_bytes, st);                   \
  } while (0)
  if (a.mode == kSweepDraw) { if (nx == 1) BNMTF_L(1, kSweepDraw); else BNMTF_L(2, kSweepDraw); }
  else                      { if (nx == 1) BNMTF_L(1, kSweepMode); else BNMTF_L(2, kSweepMode); }
#undef BNMTF_L
}

}  // namespace bnmtf
