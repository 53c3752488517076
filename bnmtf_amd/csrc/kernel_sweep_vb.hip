// K3-VB: the variational half sweep (bnmf_vb_optimised.py:189-211, update_U(k) + update_exp_U(k) for k = 0..K-1)
// in the register/LDS-resident shape of sweep_chip.inc (block shapes: below).
//
// Per unit i and column k (E = expectations, S2 = var + E^2 of the OTHER factor):
//   tau_ik = exptau * sum_j M_ij S2_jk                = exptau * (colsum2_k - sum_{j in miss(i)} S2_jk)
//   mu_ik  = (-lambda_ik + exptau * num_ik) / tau_ik ,  num as in the Gibbs sweep with U, V -> E[U], E[V]
//   E[U_ik], Var[U_ik] = moments of TN(mu_ik, tau_ik)  (fp64: exp, erfc)
// Differences from the Gibbs kernels:
//  * the panel of column k is the PAIR (E_jk, S2_jk) interleaved per j (PostArgs::XS, written by post_kernel), one
//    ds_read_b64 per slot, so the main loop shares the pre-pass's 8-byte slot addressing and its two 66 KiB buffers;
//  * the fp64 moments are NOT evaluated by every wave (8 waves x ~300 fp64 instructions per column would swamp the
//    SIMDs): each wave posts (mu, tau) of its two units to LDS, wave 0 evaluates the units of the block one per
//    lane, writes mu/tau/E/Var/S2 of column k to global memory and posts E back.  Two block barriers per column;
//  * the ELBO / exp_square_diff pieces (more fp64 erfc/log) do not feed the recurrence: the sweep stores
//    sum_miss S2 and sum_miss E^2 per (unit, column) and vb_pieces_kernel evaluates them afterwards, all units in parallel.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "sweep_common.h"
#include "many.h"

namespace bnmtf {

// Two block shapes (the panels a block stages are L2/fabric traffic: 6.3 MB per block and sweep at cfg5, so units per
// block is what the large sizes pay for):
//   16 unit waves, 32 units per block (<= 128 VGPRs; wave 0 also evaluates the moments)    -- when that fills the chip;
//    8 unit waves + 2 service waves (staging of every panel, the first one the moments)     -- smaller problems.

// COV = 1 (round 6): the F and G half sweeps of the variational TRI-factorisation (bnmtf_vb_optimised.py:241-250, 264-273) -- the
// same update against the effective factor (mean and second moment in the pair panels), with
//   * the columns walked in the order the host hands over (SweepArgs::order: the reference shuffles them, :178, :184) -- the
//     panel buffers alternate with the STEP, the column of a step sits in a lane of one register (v_readlane);
//   * the covariance term (:246, :269)  sum_t S(k,t) mv_t (fs_t - x_k S(k,t)),  t = the inner index of S (lane t of the unit's
//     half wave holds mv_t = the masked variance sum of the other observed-side factor and fs_t = sum_c x_c S(c,t), kept current
//     with one FMA per column): it rides in the numerator's half-wave sum, S(k, .) comes from an LDS copy laid out [column][t].
template <int EM, int NX, int NW, int NS, int COV = 0>
__device__ __forceinline__ void sweep_vb_body(const SweepArgs& a, const FastArgs& f, float* lds) {
  constexpr int KP = NX * 32;
  static_assert(!COV || NX == 1, "the tri-factorisation's sweeps: K, L <= 32");
  constexpr int EH = EM / 2;
  static_assert(EM % 2 == 0, "slots are processed in pairs");
  const int PW = f.pw;
  float* Cs = lds;                          // [KP][KP]
  float* c2s = lds + KP * KP;               // [KP] colsum2 of the other factor
  float* xch = c2s + KP;                    // [2*NW][4] (mu, tau, sum_miss S2, sum_miss E^2) posted by the owners
  float* ret = xch + 2 * NW * 4;            // [2*NW] E back from wave 0
  float* Ssl = ret + 2 * NW;                // COV: [32][32] S(column, t), zero beyond (K, cov_n)
  float* pan = Ssl + (COV ? 1024 : 0);      // two pair-panel buffers of 2*PW floats
  const uint32_t pan_b = (uint32_t)(uintptr_t)(lds_fp)pan;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l5 = lane & 31;
  const int chunks2 = (2 * PW) / 256;
  const uint32_t buf_b = (uint32_t)(2 * PW) * 4u;           // bytes between the two buffers
  // the column of step kk: lane kk of ordv
  const int ordv = (COV && a.order && lane < a.K) ? a.order[lane] : lane;
  auto col_of = [&](int kk) -> int { return COV ? __builtin_amdgcn_readlane(ordv, kk) : kk; };
  if (NS > 0 && wave >= NW) {
    const int sid = wave - NW;               // NS service waves share the staging; the first one also does the moments
    // The service waves: they own no units, so it has the registers for the fp64 moments and the time for the LDS-DMA.
    // It issues every piece of every panel (a (E, S2) pair panel is 66 KiB: issued by the unit waves that was eight
    // pieces per wave and column, each blocking its wave for 150-250 cycles) and, between the two barriers of a column,
    // evaluates the block's moments, one unit per lane.  Same barrier sequence as the unit waves.
    const int K = a.K;
    int mgi = -1;
    if (sid == 0 && lane < 2 * NW) {
      const int pr = blockIdx.x * NW + (lane >> 1);
      const int uu = pr < f.npairs ? f.unit_map[2 * pr + (lane & 1)] : -1;
      mgi = uu >= 0 ? a.n0 + uu : -1;
    }
    const uint32_t stride_b = (uint32_t)f.ld2_o * 8u;
    const __amdgpu_buffer_rsrc_t rs2 = panel_rsrc(f.XoT2, (size_t)(KP / 2) * f.ld2_o * 8);
    const __amdgpu_buffer_rsrc_t rsx = panel_rsrc(f.XoS, (size_t)KP * f.ld2_o * 8);
    stage_panel_buf<(NS > 0 ? NS : 1)>(rs2, 0u, pan, chunks2, sid, lane * 16);
    sync_with_dma();
    for (int kp = 0; kp < KP / 2; ++kp) {
      if (kp + 1 < KP / 2) stage_panel_buf<(NS > 0 ? NS : 1)>(rs2, (uint32_t)(kp + 1) * stride_b, pan + (size_t)((kp + 1) & 1) * 2 * PW, chunks2, sid, lane * 16);
      sync_with_dma();
    }
    stage_panel_buf<(NS > 0 ? NS : 1)>(rsx, (uint32_t)col_of(0) * stride_b, pan, chunks2, sid, lane * 16);
    sync_with_dma();
#ifdef BNMTF_PHASE_TIMING
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = tick(0.f);
#endif
    for (int kk = 0; kk < K; ++kk) {
      const int k = col_of(kk);
      if (kk + 1 < K) stage_panel_buf<(NS > 0 ? NS : 1)>(rsx, (uint32_t)col_of(kk + 1) * stride_b, pan + (size_t)((kk + 1) & 1) * 2 * PW, chunks2, sid, lane * 16);
      // first barrier of the column without a vmcnt wait: the pieces just issued land while the moments are evaluated
      // (the second barrier carries the vmcnt(0))
      TICK(0, 0.f);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      TICK(1, 0.f);
      float4 o = {0.f, 0.f, 0.f, 0.f};
      float ef = 0.f, vf = 0.f;
      if (sid == 0 && lane < 2 * NW) {
        o = *reinterpret_cast<const float4*>(&xch[lane * 4]);
        if (mgi >= 0) tn_moments_f32(o.x, o.y, &ef, &vf);
        ret[lane] = ef;
      }
      TICK(2, ef);
      sync_with_dma();                   // lands the next panel (vmcnt) and releases the unit waves
      TICK(3, ef);
      if (sid == 0 && lane < 2 * NW && mgi >= 0) {   // the seven stores per unit go out behind the barrier: nobody waits for them
        const size_t p = (size_t)mgi * KP + k;
        a.Xself[p] = ef; a.mu_self[p] = o.x; a.tau_self[p] = o.y; a.var_self[p] = vf; a.S2self[p] = vf + ef * ef;
        f.vb_asq[p] = o.z; f.vb_vsq[p] = o.w;
      }
    }
#ifdef BNMTF_PHASE_TIMING
    if (blockIdx.x % 101 == 0 && lane == 0 && sid == 0)
      printf("vb block %d service wave: dma issue %llu  wait1 %llu  moments %llu  wait2 %llu (cycles, %d columns)\n", (int)blockIdx.x, ph[0], ph[1], ph[2], ph[3], K);
#endif
    if (f.stats) sync_with_dma();
    return;
  }
  // q hand-over between the half sweeps (FastArgs::ho_*; the 16-wave shape on one GPU): q = E[U_i] . E[V_j] of the missing
  // entries comes from the region the other direction's sweep filled, not from a pre-pass, and goes on at the end
  const bool ho_read = NS == 0 && f.ho_read, ho_write = NS == 0 && f.ho_write;
  if (ho_read) ho_issue_region<NW>(f, pan, wave, lane);
  const int pair = blockIdx.x * NW + wave;
  const bool wave_on = pair < f.npairs;
  const uint32_t base = wave_on ? f.pair_base[pair] : 0u;
  const int E = wave_on ? (int)f.pair_E[pair] : 0;
  const int u = wave_on ? f.unit_map[2 * pair + half] : -1;
  const bool valid = u >= 0;
  const uint32_t gi = (uint32_t)a.n0 + (uint32_t)(valid ? u : 0);
  const int K = a.K;
  const float tau = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.tau)));

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
  // slot addresses as LDS byte addresses of the 8-byte element j in pair-panel buffer 0 (sentinel: zeros on bank pair l5)
  uint32_t addr[EM];
  f32x2 q2[EH], vp2[EH];
#pragma unroll
  for (int h = 0; h < EH; ++h) {
    const uint32_t sent = (uint32_t)(f.mz + l5);
    const uint32_t w = (2 * h < E) ? f.off16[((size_t)(base >> 1) + h) * 64 + lane] : (sent | (sent << 16));
    addr[2 * h] = pan_b + 8u * (w & 0xFFFFu);
    addr[2 * h + 1] = pan_b + 8u * (w >> 16);
    q2[h] = f32x2{0.f, 0.f}; vp2[h] = f32x2{0.f, 0.f};
  }
  uint32_t win[EH];
  if (ho_read) {
#pragma unroll
    for (int h = 0; h < EH; ++h) win[h] = (2 * h < E) ? f.ho_in[((size_t)(base >> 1) + h) * 64 + lane] : 0u;
  }
  for (int t = tid; t < KP * KP; t += NW * 64) Cs[t] = a.C32[t];
  if (tid < KP) c2s[tid] = (float)a.colsum2_o[tid];
  if (COV)
    for (int t = tid; t < 1024; t += NW * 64) {
      const int c = t >> 5, tt = t & 31;
      Ssl[t] = (c < a.K && tt < a.cov_n) ? a.cov_S[c * a.cov_sc + tt * a.cov_st] : 0.f;
    }
  // NS == 0: wave 0, lane un: the unit it evaluates the moments for
  int mgi = -1;
  if (NS == 0 && wave == 0 && lane < 2 * NW) {
    const int pr = blockIdx.x * NW + (lane >> 1);
    const int uu = pr < f.npairs ? f.unit_map[2 * pr + (lane & 1)] : -1;
    mgi = uu >= 0 ? a.n0 + uu : -1;
  }
  // ------------------------------------------------------------ pre-pass: q = E[U_i] . E[V_j]  (pair panels of E)
  if (ho_read) {
    sync_with_dma();
#pragma unroll
    for (int h = 0; h < EH; ++h)
      if (2 * h < E) q2[h] = f32x2{pan[win[h] & 0xFFFFu], pan[win[h] >> 16]};
    sync_with_dma();
  } else {
    const uint32_t stride_b = (uint32_t)f.ld2_o * 8u;
    const __amdgpu_buffer_rsrc_t rs2 = panel_rsrc(f.XoT2, (size_t)(KP / 2) * f.ld2_o * 8);
    if (NS == 0) stage_panel_buf<NW>(rs2, 0u, pan, chunks2, wave, lane * 16);
    sync_with_dma();
    const int npair = KP / 2;
#pragma nounroll      // (KP = 32: all 16 steps unrolled keep every step's partial sums live -- 1 149 spilled registers in the 16-wave K <= 32 instantiation, found in round 6)
    for (int kp = 0; kp < npair; ++kp) {
      if (NS == 0 && kp + 1 < npair) stage_panel_buf<NW>(rs2, (uint32_t)(kp + 1) * stride_b, pan + (size_t)((kp + 1) & 1) * 2 * PW, chunks2, wave, lane * 16);
      const uint32_t boff = (uint32_t)(kp & 1) * buf_b;
      const int k0 = 2 * kp, k1 = 2 * kp + 1;
      const float xs0 = (NX == 2 && k0 >= 32) ? x[NX - 1] : x[0];
      const float x0 = half_bcast(xs0, k0 & 31, half), x1 = half_bcast(xs0, k1 & 31, half);
      const f32x2 x01 = {x0, x1};
#pragma unroll
      for (int h = 0; h < EH; ++h) {
        const f32x2 va = *(lds_cf2*)(uintptr_t)(addr[2 * h] + boff);
        const f32x2 vb = *(lds_cf2*)(uintptr_t)(addr[2 * h + 1] + boff);
        q2[h] = pk_fma(va, x01, q2[h]);
        vp2[h] = pk_fma(vb, x01, vp2[h]);
      }
      sync_with_dma();
    }
#pragma unroll
    for (int h = 0; h < EH; ++h) { q2[h] = f32x2{q2[h].x + q2[h].y, vp2[h].x + vp2[h].y}; vp2[h] = f32x2{0.f, 0.f}; }
  }

  // ------------------------------------------------------------ the K sequential columns, panels of (E_k, S2_k)
  const __amdgpu_buffer_rsrc_t rsx = panel_rsrc(f.XoS, (size_t)KP * f.ld2_o * 8);
  const uint32_t cstride_b = (uint32_t)f.ld2_o * 8u;
  // NS == 0: the panel of column k + 2 is issued inside column k's moments window by the fifteen waves that idle there,
  // into the column's own buffer (every gather of column k is behind the first barrier).  A wave's LDS-DMA issue blocks
  // for ~250 cycles per 1 KiB piece: at the top of a column that was ~1 000 cycles of every wave's slot work (66 KiB per
  // column here), in the window it costs nothing.  So the first two panels are on their way before the first column.
  if (NS == 0) {
    stage_panel_buf<NW>(rsx, (uint32_t)col_of(0) * cstride_b, pan, chunks2, wave, lane * 16);
    if (K > 1) stage_panel_buf<NW>(rsx, (uint32_t)col_of(1) * cstride_b, pan + (size_t)2 * PW, chunks2, wave, lane * 16);
  }
  sync_with_dma();
  float fs = 0.f, mvl = 0.f;
  if (COV) {
    for (int c = 0; c < K; ++c) fs = fmaf(half_bcast(x[0], c, half), Ssl[c * 32 + l5], fs);
    mvl = (valid && l5 < a.cov_n) ? a.cov_mv[(size_t)u * 32 + l5] : 0.f;
  }
#ifdef BNMTF_PHASE_TIMING
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = tick(q2[0].x);
#endif
  // hand-over epilogue's offset table through LDS: the block's slice of ho_out (first word ho_w0, ho_np pieces of 256 words =
  // 4 slot-pair rows) lands in column K - 2's panel buffer during that column's moments window, when no panel is left to stage
  const uint32_t ho_row0 = ho_write ? f.pair_base[blockIdx.x * NW] >> 1 : 0u;
  const uint32_t ho_rows = ho_write ? ((blockIdx.x * NW + NW < (uint32_t)f.npairs ? f.pair_base[blockIdx.x * NW + NW] : (uint32_t)f.ho_rows_total) >> 1) - ho_row0 : 0u;
  const int ho_np = (int)((ho_rows + 3u) / 4u);
  const bool ho_lds = ho_write && K >= 2 && ho_np <= chunks2;
  float dprev = 0.f;
  for (int kk = 0; kk < K; ++kk) {
    const int k = col_of(kk);
    const uint32_t boff = (uint32_t)(kk & 1) * buf_b;
    const float xsel = (NX == 2 && k >= 32) ? x[NX - 1] : x[0];
    const float xk = half_bcast(xsel, k & 31, half);
    // (A) column k-1's update of q
    const f32x2 dp2 = {dprev, dprev};
#pragma unroll
    for (int h = 0; h < EH; ++h) q2[h] = pk_fma(dp2, vp2[h], q2[h]);
    TICK(0, q2[0].x);
    // (B, C) gather (E_jk, S2_jk); sum q E, sum E^2, sum S2
    f32x2 qv2 = {0.f, 0.f}, vv2 = {0.f, 0.f}, ss2 = {0.f, 0.f};
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      const f32x2 ta = *(lds_cf2*)(uintptr_t)(addr[2 * h] + boff);
      const f32x2 tb = *(lds_cf2*)(uintptr_t)(addr[2 * h + 1] + boff);
      vp2[h] = f32x2{ta.x, tb.x};
      qv2 = pk_fma(q2[h], vp2[h], qv2);
      vv2 = pk_fma(vp2[h], vp2[h], vv2);
      ss2.x += ta.y; ss2.y += tb.y;
    }
    float vv_t = vv2.x + vv2.y, ss_t = ss2.x + ss2.y;
    float corr_t = fmaf(-xk, vv_t, qv2.x + qv2.y);
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) corr_t = fmaf(-x[nx], Cs[k * KP + l5 + 32 * nx], corr_t);   // all l: the l = k term is put back below
    float sc = 0.f;
    if (COV) {
      sc = Ssl[k * 32 + l5];
      corr_t = fmaf(-sc * mvl, fmaf(-xk, sc, fs), corr_t);
    }
    TICK(1, corr_t);
    corr_t = half_sum_upper(corr_t);     // right in lanes 16-31 of the half
    vv_t = half_sum_upper(vv_t);
    ss_t = half_sum_upper(ss_t);
    const float ckk = Cs[k * KP + k];
    const float psel = (NX == 2 && k >= 32) ? pl[NX - 1] : pl[0];
    const float tau_p = tau * (c2s[k] - ss_t);
    const float numer = fmaf(tau, fmaf(xk, ckk, corr_t), half_bcast(psel, k & 31, half));
    if (l5 == 16) {
      float4 o; o.x = numer / tau_p; o.y = tau_p; o.z = ss_t; o.w = vv_t;
      *reinterpret_cast<float4*>(&xch[(2 * wave + half) * 4]) = o;
    }
    TICK(2, numer);
    sync_with_dma();
    TICK(3, numer);
    // between the two barriers of a column the block's moments are evaluated, one unit per lane: by the first service
    // wave, or (NS == 0) by wave 0.  The seven global stores per unit go out behind the second barrier, whose vmcnt(0)
    // would otherwise make the whole block wait for them.
    float4 o = {0.f, 0.f, 0.f, 0.f};
    float ef = 0.f, vf = 0.f;
    const bool mom = NS == 0 && wave == 0 && lane < 2 * NW;
    if (mom) {
      o = *reinterpret_cast<const float4*>(&xch[lane * 4]);
      if (mgi >= 0) tn_moments_f32(o.x, o.y, &ef, &vf);
      ret[lane] = ef;
    } else if (NS == 0 && wave >= 1 && kk + 2 < K) {
      typedef __attribute__((address_space(3))) void* lds_ptr;
      float* dst = pan + (size_t)(kk & 1) * 2 * PW;
      const uint32_t cb = (uint32_t)col_of(kk + 2) * cstride_b;
      for (int c = wave - 1; c < chunks2; c += NW - 1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(dst + (size_t)c * 256), 16, lane * 16,
                                                  (int)(cb + (uint32_t)c * 1024u), 0, 0);
    } else if (NS == 0 && wave >= 1 && ho_lds && kk + 2 == K) {
      typedef __attribute__((address_space(3))) void* lds_ptr;
      float* dst = pan + (size_t)(kk & 1) * 2 * PW;
      const __amdgpu_buffer_rsrc_t rso = panel_rsrc(reinterpret_cast<const float*>(f.ho_out) + (size_t)ho_row0 * 64, (size_t)ho_rows * 256);
      for (int c = wave - 1; c < ho_np; c += NW - 1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rso, (lds_ptr)(dst + (size_t)c * 256), 16, lane * 16, (int)((uint32_t)c * 1024u), 0, 0);
    }
    if (NS == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS traffic only: the pieces just issued have all of column k + 1 to land (its first barrier waits for them)
    else sync_with_dma();                // service-wave shape: also lands the next panel (vmcnt) and retires this one
    if (mom && mgi >= 0) {
      const size_t p = (size_t)mgi * KP + k;
      a.Xself[p] = ef; a.mu_self[p] = o.x; a.tau_self[p] = o.y; a.var_self[p] = vf; a.S2self[p] = vf + ef * ef;
      f.vb_asq[p] = o.z; f.vb_vsq[p] = o.w;
    }
    TICK(4, numer);
    const float xnew = ret[2 * wave + half];
    dprev = xnew - xk;
    if (COV) fs = fmaf(dprev, sc, fs);
#pragma unroll
    for (int nx = 0; nx < NX; ++nx)
      if (l5 + 32 * nx == k) x[nx] = xnew;
  }

#ifdef BNMTF_PHASE_TIMING
  if (blockIdx.x % 101 == 0 && (tid & 127) == 0)
    printf("vb block %d wave %d EM %d: A %llu  BC %llu  reduce+post %llu  wait1 %llu  window %llu (cycles, %d columns)\n", (int)blockIdx.x, wave, EM, ph[0], ph[1], ph[2], ph[3], ph[4], K);
#endif
  // ------------------------------------------------------------ hand-over: the final q, sorted by the other direction's blocks
  float* stage = pan + 256;            // (the first 256 floats: the statistics' partial sums below)
  if (ho_write) {
    uint32_t wo[EH];
    if (ho_lds) {
      sync_with_dma();                 // the slice staged in column K - 2's window has landed
      const float* src = pan + (size_t)((K - 2) & 1) * 2 * PW + (size_t)((base >> 1) - ho_row0) * 64 + lane;
#pragma unroll
      for (int h = 0; h < EH; ++h) wo[h] = (2 * h < E) ? __builtin_bit_cast(uint32_t, src[h * 64]) : 0u;
    } else {
#pragma unroll
      for (int h = 0; h < EH; ++h) wo[h] = (2 * h < E) ? f.ho_out[((size_t)(base >> 1) + h) * 64 + lane] : 0u;
    }
    sync_with_dma();                   // the staging area takes the panels' bytes: everybody is done with them (and with `ret`)
#pragma unroll
    for (int h = 0; h < EH; ++h)
      if (2 * h < E) {
        stage[wo[h] & 0xFFFFu] = fmaf(dprev, vp2[h].x, q2[h].x);
        stage[wo[h] >> 16] = fmaf(dprev, vp2[h].y, q2[h].y);
      }
    if (!f.stats) sync_with_dma();
  }
  // ------------------------------------------------------------ the three sums of the SSE identity (cols sweep)
  if (f.stats) {
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
    sync_with_dma();
    if (tid < 3) {
      double s = 0.0;
      for (int w = 0; w < 2 * NW; ++w) s += red[w * 3 + tid];
      f.stats[(size_t)blockIdx.x * 4 + tid] = s;
    }
  }
  if (ho_write) ho_send_packets<NW>(f, stage, wave, lane);
}

template <int NX, int NW, int NS, int COV = 0>
__global__ __launch_bounds__((NW + NS) * 64, 1) void sweep_vb_kernel(SweepArgs a, FastArgs f) {
  extern __shared__ float lds[];
  const int wv = (int)(threadIdx.x >> 6);
  const int pr = blockIdx.x * NW + wv;
  const int e0 = __builtin_amdgcn_readfirstlane((wv < NW && pr < f.npairs) ? (int)f.pair_E[pr] : 0);   // service waves run in the smallest class
  if (e0 <= 8) sweep_vb_body<8, NX, NW, NS, COV>(a, f, lds);
  else if (e0 <= 16) sweep_vb_body<16, NX, NW, NS, COV>(a, f, lds);
  else if (e0 <= 24) sweep_vb_body<24, NX, NW, NS, COV>(a, f, lds);
  else if (e0 <= 28) sweep_vb_body<28, NX, NW, NS, COV>(a, f, lds);
  else sweep_vb_body<kWideMaxSlots, NX, NW, NS, COV>(a, f, lds);      // host guarantees e0 <= kWideMaxSlots
}

struct SweepVbPack { SweepArgs a; FastArgs f; };
template <int NX, int NW, int NS, int COV = 0>       // list form (many.h): blockIdx.z = model; a model with fewer blocks than the launch leaves
__global__ __launch_bounds__((NW + NS) * 64, 1) void sweep_vb_many(const SweepVbPack* list, int) {
  extern __shared__ float lds[];
  const SweepVbPack p = load_pack(list, blockIdx.z);
  if ((int)blockIdx.x >= (p.f.npairs + NW - 1) / NW) return;
  const int wv = (int)(threadIdx.x >> 6);
  const int pr = blockIdx.x * NW + wv;
  const int e0 = __builtin_amdgcn_readfirstlane((wv < NW && pr < p.f.npairs) ? (int)p.f.pair_E[pr] : 0);
  if (e0 <= 8) sweep_vb_body<8, NX, NW, NS, COV>(p.a, p.f, lds);
  else if (e0 <= 16) sweep_vb_body<16, NX, NW, NS, COV>(p.a, p.f, lds);
  else if (e0 <= 24) sweep_vb_body<24, NX, NW, NS, COV>(p.a, p.f, lds);
  else if (e0 <= 28) sweep_vb_body<28, NX, NW, NS, COV>(p.a, p.f, lds);
  else sweep_vb_body<kWideMaxSlots, NX, NW, NS, COV>(p.a, p.f, lds);
}

int sweep_vb_blocks(int npairs, int nw) { return (npairs + nw - 1) / nw; }
static size_t sweep_vb_lds_bytes(int KP, int pw, bool cov = false) { return sizeof(float) * ((size_t)KP * KP + KP + 2 * 16 * 5 + (cov ? 1024 : 0) + 4 * (size_t)pw); }

bool sweep_vb_supported(int KP, int pw) { return sweep_vb_lds_bytes(KP, pw) <= 160 * 1024; }
bool sweep_vb_cov_supported(int KP, int pw) { return KP == 32 && sweep_vb_lds_bytes(KP, pw, true) <= 160 * 1024; }

template <int NX, int NW, int NS, int COV = 0>
static void launch_vb_inst(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  static std::atomic<uint64_t> lds_ok{0};
  const int nblocks = sweep_vb_blocks(f.npairs, NW);
  if (g_recorder) {
    if constexpr (COV == 0) {            // (the tri-factorisation's sweeps have no list form)
    static std::atomic<uint64_t> lds_ok_many{0};
    if (nblocks <= 0 || !allow_full_lds((const void*)sweep_vb_many<NX, NW, NS, COV>, lds_ok_many)) return;
    SweepVbPack p; memset(&p, 0, sizeof(p)); p.a = a; p.f = f;
    record_launch((const void*)sweep_vb_many<NX, NW, NS, COV>, dim3(nblocks), dim3((NW + NS) * 64),
                  std::max(sweep_vb_lds_bytes(a.KP, f.pw, COV != 0), sizeof(float) * ((size_t)a.KP * a.KP + a.KP + 2 * 16 * 5 + (size_t)f.ho_lds_floats)), p, true);
    }
    return;
  }
  if (nblocks > 0 && allow_full_lds((const void*)sweep_vb_kernel<NX, NW, NS, COV>, lds_ok)) hipLaunchKernelGGL((sweep_vb_kernel<NX, NW, NS, COV>), dim3(nblocks), dim3((NW + NS) * 64), std::max(sweep_vb_lds_bytes(a.KP, f.pw, COV != 0), sizeof(float) * ((size_t)a.KP * a.KP + a.KP + 2 * 16 * 5 + (size_t)f.ho_lds_floats)), st, a, f);
}

// f.nw = 16: 16 unit waves per block; anything else: 8 unit waves + 2 service waves
void launch_sweep_vb(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  const int nx = a.KP / 32;
  if (a.cov_S) {        // the tri-factorisation's F / G sweeps (callers: sweep_vb_cov_supported)
    if (f.nw == 16) launch_vb_inst<1, 16, 0, 1>(a, f, st); else launch_vb_inst<1, 8, 2, 1>(a, f, st);
    return;
  }
  if (f.nw == 16) { if (nx == 1) launch_vb_inst<1, 16, 0>(a, f, st); else launch_vb_inst<2, 16, 0>(a, f, st); }
  else            { if (nx == 1) launch_vb_inst<1, 8, 2>(a, f, st);  else launch_vb_inst<2, 8, 2>(a, f, st); }
}

// ELBO / exp_square_diff pieces of one sweep (bnmf_vb_optimised.py:163-177, 185-187), one wave per unit, lane = column:
//   [0] sum_k tau/2 (Var + (E - mu)^2)   [1] sum_k log(1/2 erfc(-mu sqrt(tau/2)))   [2] sum_k log tau   [3] sum_k lambda E
//   [4] sum_k S2self sum_miss S2other    [5] sum_k E^2 sum_miss Eother^2
// out: one row of 8 per BLOCK of four units (vb_finish_kernel only needs the totals).
__device__ __forceinline__ void vb_pieces_body(int n, int n0, int KP, int K, const float* mu, const float* tauq, const float* ex,
                                               const float* var, const float* lambda, const float* asq, const float* vsq, double* out) {
  const int lane = threadIdx.x & 63, u = blockIdx.x * 4 + (threadIdx.x >> 6);
  double p[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (u < n && lane < K) {
    const size_t q = (size_t)(n0 + u) * KP + lane;
    const double m = (double)mu[q], t = (double)tauq[q], e = (double)ex[q], v = (double)var[q];
    const double dm = e - m;
    p[0] = 0.5 * t * (v + dm * dm);
    p[1] = log(0.5 * erfc(-m * sqrt(t) * 0.7071067811865476));
    p[2] = log(t);
    p[3] = (double)lambda[(size_t)u * KP + lane] * e;
    p[4] = (v + e * e) * (double)asq[q];
    p[5] = e * e * (double)vsq[q];
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) p[c] += __shfl_xor(p[c], s, 64);
  }
  // one row of partial sums per block (4 units), in unit order
  __shared__ double red[4][6];
  if (lane == 0) for (int c = 0; c < 6; ++c) red[threadIdx.x >> 6][c] = p[c];
  sync_with_dma();
  if (threadIdx.x < 6) out[(size_t)blockIdx.x * 8 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void vb_pieces_kernel(int n, int n0, int KP, int K, const float* mu, const float* tauq, const float* ex,
                                                         const float* var, const float* lambda, const float* asq, const float* vsq, double* out) {
  vb_pieces_body(n, n0, KP, K, mu, tauq, ex, var, lambda, asq, vsq, out);
}
struct VbPiecesPack { int n, n0, KP, K; const float *mu, *tauq, *ex, *var, *lambda, *asq, *vsq; double* out; };
__global__ __launch_bounds__(256) void vb_pieces_many(const VbPiecesPack* list, int) {      // list form (many.h): blockIdx.z = model
  const VbPiecesPack p = load_pack(list, blockIdx.z);
  if ((int)blockIdx.x >= (p.n + 3) / 4) return;
  vb_pieces_body(p.n, p.n0, p.KP, p.K, p.mu, p.tauq, p.ex, p.var, p.lambda, p.asq, p.vsq, p.out);
}

// the same with sum_miss S2other / sum_miss Eother^2 taken from the slabs of kernel_maskgemm.hip ([msplit][n_pad][2 KP], local
// unit rows), added in slab order as the sweep adds them
__global__ __launch_bounds__(256) void vb_pieces_slabs_kernel(int n, int n0, int KP, int K, const float* mu, const float* tauq, const float* ex,
                                                               const float* var, const float* lambda, const float* mslabs, int msplit, int n_pad, double* out) {
  const int lane = threadIdx.x & 63, u = blockIdx.x * 4 + (threadIdx.x >> 6);
  double p[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (u < n && lane < K) {
    const size_t q = (size_t)(n0 + u) * KP + lane;
    const double m = (double)mu[q], t = (double)tauq[q], e = (double)ex[q], v = (double)var[q];
    const double dm = e - m;
    const size_t el = (size_t)u * (2 * KP) + lane;
    const float asq = slab_sum_ordered(mslabs, msplit, (size_t)n_pad * (2 * KP), el);
    const float vsq = slab_sum_ordered(mslabs, msplit, (size_t)n_pad * (2 * KP), el + KP);
    p[0] = 0.5 * t * (v + dm * dm);
    p[1] = log(0.5 * erfc(-m * sqrt(t) * 0.7071067811865476));
    p[2] = log(t);
    p[3] = (double)lambda[(size_t)u * KP + lane] * e;
    p[4] = (v + e * e) * (double)asq;
    p[5] = e * e * (double)vsq;
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) p[c] += __shfl_xor(p[c], s, 64);
  }
  __shared__ double red[4][6];
  if (lane == 0) for (int c = 0; c < 6; ++c) red[threadIdx.x >> 6][c] = p[c];
  sync_with_dma();
  if (threadIdx.x < 6) out[(size_t)blockIdx.x * 8 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
void launch_vb_pieces_slabs(int n, int n0, int KP, int K, const float* mu, const float* tauq, const float* ex, const float* var,
                            const float* lambda, const float* mslabs, int msplit, int n_pad, double* out, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(vb_pieces_slabs_kernel, dim3((n + 3) / 4), dim3(256), 0, st, n, n0, KP, K, mu, tauq, ex, var, lambda, mslabs, msplit, n_pad, out);
}

void launch_vb_pieces(int n, int n0, int KP, int K, const float* mu, const float* tauq, const float* ex, const float* var,
                      const float* lambda, const float* asq, const float* vsq, double* out, hipStream_t st) {
  if (n <= 0) return;
  if (g_recorder) {
    VbPiecesPack p; memset(&p, 0, sizeof(p));
    p.n = n; p.n0 = n0; p.KP = KP; p.K = K; p.mu = mu; p.tauq = tauq; p.ex = ex; p.var = var; p.lambda = lambda; p.asq = asq; p.vsq = vsq; p.out = out;
    record_launch((const void*)vb_pieces_many, dim3((n + 3) / 4), dim3(256), 0, p, true);
    return;
  }
  hipLaunchKernelGGL(vb_pieces_kernel, dim3((n + 3) / 4), dim3(256), 0, st, n, n0, KP, K, mu, tauq, ex, var, lambda, asq, vsq, out);
}

}  // namespace bnmtf
