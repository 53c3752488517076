// K3 "ahead": the on-chip half sweep with the sampler taken OFF the units' critical path (round 3).
//
// The sweep of sweep_chip.inc runs a column as  slot work -> reduce -> barrier -> sampler -> barrier : while the two
// sampler waves walk the ~50 dependent instructions of a draw, the other fourteen waves of the block idle (38 % of a
// column by the phase stamps), and nothing can be put there because the next column's numerator needs the draw.
// It does not need it for long, though: everything is linear in the draw.  With q^(c-1) = q before column c's delta,
//
//   sum_miss q^(c) v_{c+1} = sum_miss q^(c-1) v_{c+1} + delta_c * X_{c,c+1} ,     X_{c,c+1} = sum_miss v_c v_{c+1}
//   sum_l x_l C_{l,c+1}    = sum_l x^(c-1)_l C_{l,c+1} + delta_c * C_{c,c+1}
//
// so while the sampler draws column c the unit waves already do the whole slot work of column c + 1 with the state of
// one column earlier and post  A = sum q^(c-1) v_{c+1} - sum_l x^(c-1)_l C0_{l,c+1}  and  X_{c,c+1}  (C0 = the Gram with
// a zero diagonal); the sampler finishes the numerator with two FMAs,
//   numer_{c+1} = pl_{c+1} - tau x_{c+1} asq_{c+1} + tau (A + delta_c (X - C_{c,c+1})) ,
// the moment delta_c exists.  ONE barrier per column, no idle window; the price is a second register set of gathered
// values (v_c stays for X and for the deferred update q += delta_c v_c) -- which is why a wave here holds FOUR units
// (two pairs of the existing slot layout) in up to 256 registers, eight waves per block, and keeps its slot offsets
// packed two per register (16-bit byte offsets, taken apart by v_mad_u32_u16 with the panel base folded in).
// asq_c = sum_miss v_c^2 does not depend on the chain at all: the pre-pass forms it for every column beside q.
//
// Same arithmetic as the reference's column update (bnmf_gibbs_optimised.py:134-142, 167-177), same candidate
// sequence as oracle/rng.py; the order of the floating-point sums differs from sweep_chip.inc, so a problem is run by
// one of the two bodies throughout (api.hip picks per direction).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "sweep_common.h"

namespace bnmtf {

constexpr int kAheadWaves = 8;              // waves per block, two pairs (four units) each: 32 units per block
constexpr int kAheadPairsPerBlock = 2 * kAheadWaves;
constexpr int kAheadCands = 4;

struct AheadLds { int C0, pan, xs, pls, tab, ax, dr, asq, Cd, total; };      // float offsets
__host__ __device__ inline AheadLds ahead_lds(int KP, int pw) {
  AheadLds L;
  L.C0 = 0;                                  // [KP][KP] Gram of the other factor, diagonal zeroed
  L.pan = KP * KP;                           // pre-pass: two pair panels (4 pw) | main loop: three single panels (3 pw) ...
  L.xs = L.pan + 3 * pw;                     // ... and, behind them, what only the main loop needs: x [32][KP]
  L.pls = L.xs + 32 * KP;                    // tau P - lambda [32][KP]
  L.tab = L.pls + 32 * KP;                   // [2][128] float4 candidates (nl, z, u2, -) of columns c, c + 1
  L.ax = L.tab + 2 * 128 * 4;                // [2][32] (A, X)
  L.dr = L.ax + 2 * 32 * 2;                  // [2][32] (draw, delta)
  const int main_end = L.dr + 2 * 32 * 2, pre_end = L.pan + 4 * pw;
  L.asq = main_end > pre_end ? main_end : pre_end;   // [32][KP] sum_miss v_c^2: written by the pre-pass, read by the sampler
  L.Cd = L.asq + 32 * KP;                    // [KP] diagonal of the Gram
  L.total = L.Cd + KP;
  return L;
}

__device__ __forceinline__ uint32_t slot_lo(uint32_t w, uint32_t base) { uint32_t r; asm("v_mad_u32_u16 %0, %1, 1, %2" : "=v"(r) : "v"(w), "s"(base)); return r; }
__device__ __forceinline__ uint32_t slot_hi(uint32_t w, uint32_t base) { uint32_t r; asm("v_mad_u32_u16 %0, %1, 1, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(w), "s"(base)); return r; }
__device__ __forceinline__ uint32_t slot_lo2(uint32_t w, uint32_t base) { uint32_t r; asm("v_mad_u32_u16 %0, %1, 2, %2" : "=v"(r) : "v"(w), "s"(base)); return r; }
__device__ __forceinline__ uint32_t slot_hi2(uint32_t w, uint32_t base) { uint32_t r; asm("v_mad_u32_u16 %0, %1, 2, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(w), "s"(base)); return r; }

#ifdef BNMTF_PHASE_TIMING
#define ATICK(i, dep) do { const unsigned long long t_ = tick(dep); ph[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define ATICK(i, dep) do { } while (0)
#endif

template <int EM, int NX, int MODE>
__device__ __forceinline__ void sweep_ahead_body(const SweepArgs& a, const FastArgs& f, float* lds) {
  constexpr int KP = NX * 32, EH = EM / 2, NC = kAheadCands, NW = kAheadWaves;
  static_assert(EM % 2 == 0 && NC == 4, "slots in pairs, candidates in quads");
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) f32x2 lds_f2;
  typedef __attribute__((address_space(3))) f32x4 lds_f4;
  typedef __attribute__((address_space(3))) float lds_f;
  const int PW = f.pw;
  const AheadLds L = ahead_lds(KP, PW);
  float* Cs = lds + L.C0;
  float* pan = lds + L.pan;
  const uint32_t lds_b = (uint32_t)(uintptr_t)(lds_fp)lds;
  const uint32_t pan_b = lds_b + 4u * (uint32_t)L.pan;
  const uint32_t xs_b = lds_b + 4u * (uint32_t)L.xs, pls_b = lds_b + 4u * (uint32_t)L.pls, tab_b = lds_b + 4u * (uint32_t)L.tab;
  const uint32_t ax_b = lds_b + 4u * (uint32_t)L.ax, dr_b = lds_b + 4u * (uint32_t)L.dr, asq_b = lds_b + 4u * (uint32_t)L.asq, cd_b = lds_b + 4u * (uint32_t)L.Cd;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l5 = lane & 31;
  const int K = a.K;
  const float tau = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.tau)));

  // the wave's two pairs: P = 0, 1 -> pair blockIdx * 16 + P * 8 + wave ; unit-in-block ub = 2 * (P * 8 + wave) + half
  int u[2], ub[2];
  bool valid[2];
  uint32_t gi[2];
  uint32_t w16[2][EH];                       // slots (2h, 2h+1): 16-bit BYTE offsets inside a single-column panel (4 j)
#pragma unroll
  for (int P = 0; P < 2; ++P) {
    const int pair = blockIdx.x * kAheadPairsPerBlock + P * NW + wave;
    const bool on = pair < f.npairs && (int)f.pair_E[pair] <= EM;
    const uint32_t base = on ? f.pair_base[pair] : 0u;
    const int E = on ? (int)f.pair_E[pair] : 0;
    u[P] = on ? f.unit_map[2 * pair + half] : -1;
    valid[P] = u[P] >= 0;
    ub[P] = 2 * (P * NW + wave) + half;
    gi[P] = (uint32_t)a.n0 + (uint32_t)(valid[P] ? u[P] : 0);
    const uint32_t sent = (uint32_t)(f.mz + l5);
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      const uint32_t w = (2 * h < E) ? f.off16[((size_t)(base >> 1) + h) * 64 + lane] : (sent | (sent << 16));
      w16[P][h] = w << 2;
    }
  }
  // x = the units' rows of the factor, pl = tau P - lambda (P = the contraction's slabs summed): lane l5 holds columns l5, l5 + 32
  float x[2][NX], pl[2][NX];
#pragma unroll
  for (int P = 0; P < 2; ++P)
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) {
      const int kk = l5 + 32 * nx;
      float s = 0.f;
      if (valid[P]) s = slab_sum_ordered(a.slabs, a.split, (size_t)a.n_pad * KP, (size_t)u[P] * KP + kk);
      x[P][nx] = valid[P] ? a.Xself[(size_t)gi[P] * KP + kk] : 0.f;
      pl[P][nx] = valid[P] ? fmaf(tau, s, -a.lambda[(size_t)u[P] * KP + kk]) : 0.f;
    }
  for (int t = tid; t < KP * KP; t += NW * 64) Cs[t] = (t / KP == t % KP) ? 0.f : a.C32[t];
  if (tid < KP) lds[L.Cd + tid] = a.C32[tid * KP + tid];

#ifdef BNMTF_PHASE_TIMING
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_start = tick(x[0][0] + pl[0][0]);
  unsigned long long tlast = t_start;
#endif
  f32x2 q2[2][EH];                          // q on slots (2h, 2h+1)
  // ------------------------------------------------------------ pre-pass over pair panels: q = x . v_j , asq_c = sum_miss v_c^2
  {
    f32x2 accA[2][EH], accB[2][EH];
#pragma unroll
    for (int P = 0; P < 2; ++P)
#pragma unroll
      for (int h = 0; h < EH; ++h) { accA[P][h] = f32x2{0.f, 0.f}; accB[P][h] = f32x2{0.f, 0.f}; }
    const int chunks2 = (2 * PW) / 256;
    const uint32_t stride_b = (uint32_t)f.ld2_o * 8u;
    const __amdgpu_buffer_rsrc_t rs2 = panel_rsrc(f.XoT2, (size_t)(KP / 2) * f.ld2_o * 8);
    stage_panel_buf<NW>(rs2, 0u, pan, chunks2, wave, lane * 16);
    sync_with_dma();
    const int npair = KP / 2;
#ifdef X_NOPRE
    for (int kp = 0; kp < 0; ++kp) {
#else
    for (int kp = 0; kp < npair; ++kp) {
#endif
      if (kp + 1 < npair) stage_panel_buf<NW>(rs2, (uint32_t)(kp + 1) * stride_b, pan + (size_t)((kp + 1) & 1) * 2 * PW, chunks2, wave, lane * 16);
      uint32_t pb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pan_b + (uint32_t)((kp & 1) * 2 * PW) * 4u));   // element j of a pair panel sits 8 j bytes in
      asm volatile("" : "+s"(pb));             // opaque: the 2 x 56 slot addresses of the two buffers are not kept in registers across the loop
      const int k0 = 2 * kp, k1 = 2 * kp + 1;
#pragma unroll
      for (int P = 0; P < 2; ++P) {
        // both registers are read and the choice is made on the broadcast values: a select between x[P][0] and x[P][1]
        // itself turns the array into an indexed stack object
        f32x2 x01 = {half_bcast(x[P][0], k0 & 31, half), half_bcast(x[P][0], k1 & 31, half)};
        if (NX == 2) {
          const f32x2 xhi = {half_bcast(x[P][NX - 1], k0 & 31, half), half_bcast(x[P][NX - 1], k1 & 31, half)};
          x01 = k0 >= 32 ? xhi : x01;
        }
        f32x2 vv = {0.f, 0.f};
#pragma unroll
        for (int h = 0; h < EH; ++h) {
          const f32x2 va = *(lds_cf2*)(uintptr_t)slot_lo2(w16[P][h], pb);
          const f32x2 vb = *(lds_cf2*)(uintptr_t)slot_hi2(w16[P][h], pb);
          accA[P][h] = pk_fma(va, x01, accA[P][h]);
          accB[P][h] = pk_fma(vb, x01, accB[P][h]);
          vv = pk_fma(va, va, vv);
          vv = pk_fma(vb, vb, vv);
          // pinned here: left alone the compiler sinks the accumulation below the barrier and keeps every gathered value until then
          asm volatile("" : "+v"(accA[P][h]), "+v"(accB[P][h]));
        }
        const float s0 = half_sum_upper(vv.x), s1 = half_sum_upper(vv.y);
        if (l5 == 16) *(lds_f2*)(uintptr_t)(asq_b + 4u * (uint32_t)(ub[P] * KP + k0)) = f32x2{s0, s1};
      }
      sync_with_dma();
    }
#pragma unroll
    for (int P = 0; P < 2; ++P)
#pragma unroll
      for (int h = 0; h < EH; ++h) q2[P][h] = f32x2{accA[P][h].x + accA[P][h].y, accB[P][h].x + accB[P][h].y};
  }
  ATICK(0, q2[0][0].x);

  // ------------------------------------------------------------ main-loop state in LDS, first three panels, column 0's candidates
  const int chunks1 = PW / 256;
  const __amdgpu_buffer_rsrc_t rs1 = panel_rsrc(f.XoT, (size_t)KP * f.ldT_o * 4);
  const uint32_t col_b = (uint32_t)f.ldT_o * 4u;
  for (int p = 0; p < 3 && p < K; ++p) stage_panel_buf<NW>(rs1, (uint32_t)p * col_b, pan + (size_t)p * PW, chunks1, wave, lane * 16);
#pragma unroll
  for (int P = 0; P < 2; ++P)
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) {
      lds[L.xs + ub[P] * KP + l5 + 32 * nx] = x[P][nx];
      lds[L.pls + ub[P] * KP + l5 + 32 * nx] = pl[P][nx];
    }
  // roles: waves 0, 1 draw (one (unit, candidate) per lane), waves 2, 3 make the next column's candidates, waves 4..7 issue the LDS-DMA
  constexpr int NS = 2;
  const int s_unit = (wave & 1) * 16 + (lane >> 2), s_cand = lane & 3;      // waves 0..3
  bool s_valid = false;
  uint32_t s_row = (uint32_t)a.n0;
  if (wave < 2 * NS) {
    const int spr = blockIdx.x * kAheadPairsPerBlock + (s_unit >> 1);
    const int su = (spr < f.npairs && (int)f.pair_E[spr] <= kWideMaxSlots) ? f.unit_map[2 * spr + (s_unit & 1)] : -1;
    s_valid = su >= 0;
    s_row += (uint32_t)(s_valid ? su : 0);
  }
  auto fill_tab = [&](int col) {
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
    if (MODE == kSweepDraw) {
      uint32_t row = s_row;
      asm volatile("" : "+v"(row));                            // opaque: no partial rounds of this call are kept across columns
      const U4 r = philox4x32_10(row, (uint32_t)col, a.it, a.stream + 16u * (uint32_t)s_cand, a.key0, a.key1);
      const TnCand cd = tn_cand_pre(r.x, r.y);
      e = f32x4{cd.nl, cd.z, cd.u2, 0.f};
    }
    *(lds_f4*)(uintptr_t)(tab_b + 16u * (uint32_t)((col & 1) * 128 + s_unit * NC + s_cand)) = e;
  };
  if (wave >= NS && wave < 2 * NS) fill_tab(0);

  // the unit waves' C term and the posts
  float dprev_s = 0.f;                                 // sampler lanes: delta of the previous column
  // sampler: what can be read ahead of the barrier for column c
  float s_t0 = 0.f, s_c0 = 0.f, s_xo = 0.f;
  TnPre s_pre = {0.f, 0.f, 0.f, false};
  float s_taup = 0.f;
  auto sampler_prefetch = [&](int c) {
    const float plc = *(lds_f*)(uintptr_t)(pls_b + 4u * (uint32_t)(s_unit * KP + c));
    s_xo = *(lds_f*)(uintptr_t)(xs_b + 4u * (uint32_t)(s_unit * KP + c));
    const float asq = *(lds_f*)(uintptr_t)(asq_b + 4u * (uint32_t)(s_unit * KP + c));
    const float cdiag = *(lds_f*)(uintptr_t)(cd_b + 4u * (uint32_t)c);
    s_c0 = c > 0 ? Cs[(c - 1) * KP + c] : 0.f;
    s_taup = tau * (cdiag - asq);
    s_pre = tn_fast_pre(s_taup);
    s_t0 = fmaf(-tau * s_xo, asq, plc);
  };
  auto sampler_draw = [&](int c) {
    const uint32_t par = (uint32_t)(c & 1);
    const f32x2 ax = *(lds_f2*)(uintptr_t)(ax_b + 8u * (par * 32u + (uint32_t)s_unit));
    const f32x4 ce = *(lds_f4*)(uintptr_t)(tab_b + 16u * (par * 128u + (uint32_t)(s_unit * NC + s_cand)));
    const float numer = fmaf(tau, fmaf(dprev_s, ax.y - s_c0, ax.x), s_t0);
    float r = 0.f;
    if (MODE == kSweepDraw) {
      const TnFast tf = tn_fast_post(s_pre, numer);
      TnCand cand = {ce.x, ce.y, ce.z};
      bool need = s_valid && tf.live;
      for (uint32_t cbase = 0;;) {
        float xc;
        const bool acc = tn_cand_post(tf, cand, &xc);
        const int xa = __builtin_bit_cast(int, acc ? tn_guard(xc) : -1.0f);      // draws are >= 0
        const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xa, 0x00, 0xF, 0xF, true));   // quad_perm [0,0,0,0]
        const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xa, 0x55, 0xF, 0xF, true));   // [1,1,1,1]
        const float x2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xa, 0xAA, 0xF, 0xF, true));   // [2,2,2,2]
        const float x3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xa, 0xFF, 0xF, 0xF, true));   // [3,3,3,3]
        float first = x3;
        first = x2 >= 0.f ? x2 : first;
        first = x1 >= 0.f ? x1 : first;
        first = x0 >= 0.f ? x0 : first;
        if (need && first >= 0.f) { r = first; need = false; }
        cbase += NC;
        if (__ballot(need) == 0ull || cbase >= 4096u) break;
        uint32_t row = s_row;
        asm volatile("" : "+v"(row));                          // opaque: nothing of this Philox call is hoisted out of the column loop
        const U4 ph4 = philox4x32_10(row, (uint32_t)c, a.it, a.stream + 16u * (cbase + (uint32_t)s_cand), a.key0, a.key1);
        cand = tn_cand_pre(ph4.x, ph4.y);
      }
    } else {
      const float mu = numer / s_taup;
      r = fmaxf((s_valid && s_taup > 0.f && mu > 0.f) ? mu : 0.f, a.min_x);
    }
    dprev_s = r - s_xo;
    // (the unit's x row in LDS is brought up to date by its own wave a column later: the C term of column c + 1,
    // formed while this draw is made, must see the OLD x_c)
    if (s_cand == 0) *(lds_f2*)(uintptr_t)(dr_b + 8u * (par * 32u + (uint32_t)s_unit)) = f32x2{r, dprev_s};
  };

  f32x2 vs[2][2][EH];                       // gathered values [set][pair][slots 2h, 2h+1]; column c lives in set c & 1
  // A_{cn} = sum_miss q v_cn - sum_l x_l C0_{l,cn}  and  X = sum_miss v_{cn-1} v_cn  from the gathered set SN (other set: SO)
  auto reduce_post = [&](auto sn_c, int cn, bool with_x) {
    constexpr int SN = decltype(sn_c)::value, SO = 1 - SN;
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      f32x2 s2[2] = {{0.f, 0.f}, {0.f, 0.f}}, x2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int h = 0; h < EH; ++h) {
        s2[h & 1] = pk_fma(q2[P][h], vs[SN][P][h], s2[h & 1]);
        x2[h & 1] = pk_fma(vs[SO][P][h], vs[SN][P][h], x2[h & 1]);
      }
      float s_t = (s2[0].x + s2[0].y) + (s2[1].x + s2[1].y);
      float x_t = with_x ? (x2[0].x + x2[0].y) + (x2[1].x + x2[1].y) : 0.f;
#pragma unroll
      for (int nx = 0; nx < NX; ++nx) s_t = fmaf(-*(lds_f*)(uintptr_t)(xs_b + 4u * (uint32_t)(ub[P] * KP + l5 + 32 * nx)), Cs[cn * KP + l5 + 32 * nx], s_t);
      s_t = half_sum_upper(s_t);
      x_t = half_sum_upper(x_t);
      if (l5 == 16) *(lds_f2*)(uintptr_t)(ax_b + 8u * ((uint32_t)(cn & 1) * 32u + (uint32_t)ub[P])) = f32x2{s_t, x_t};
    }
  };
  auto gather = [&](auto sn_c, uint32_t pb) {
    constexpr int SN = decltype(sn_c)::value;
#pragma unroll
    for (int P = 0; P < 2; ++P)
#pragma unroll
      for (int h = 0; h < EH; ++h) {
        vs[SN][P][h].x = *(lds_cf*)(uintptr_t)slot_lo(w16[P][h], pb);
        vs[SN][P][h].y = *(lds_cf*)(uintptr_t)slot_hi(w16[P][h], pb);
      }
  };
  auto panel_base = [&](int c) {
    uint32_t pb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pan_b + (uint32_t)((c % 3) * PW) * 4u));
    asm volatile("" : "+s"(pb));               // opaque: slot addresses are made per gather, never kept
    return pb;
  };
  using c0t = std::integral_constant<int, 0>;
  using c1t = std::integral_constant<int, 1>;
#pragma unroll
  for (int P = 0; P < 2; ++P)
#pragma unroll
    for (int h = 0; h < EH; ++h) { vs[0][P][h] = f32x2{0.f, 0.f}; vs[1][P][h] = f32x2{0.f, 0.f}; }

  sync_with_dma();                            // panels 0..2, x, pl, candidates of column 0
  // priming: column 0's sums from the state as it is
  gather(c0t{}, panel_base(0));
  reduce_post(c0t{}, 0, false);
  if (wave < NS) sampler_prefetch(0);
  sync_with_dma();
  ATICK(1, q2[0][0].x);

  // One column: the sampler waves draw column c; everybody applies delta_{c-1}, gathers v_{c+1} and posts its sums.
  auto step = [&](auto sn_c, int c) {
    constexpr int SN = decltype(sn_c)::value;          // set of v_{c+1} (it held v_{c-1})
    if (wave < NS) sampler_draw(c);
    else if (wave < 2 * NS && c + 1 < K) fill_tab(c + 1);
    __builtin_amdgcn_sched_barrier(0);
    ATICK(2, dprev_s);
    if (c >= 1) {
      // delta_{c-1}: q += delta v_{c-1}; the unit's x row follows
#pragma unroll
      for (int P = 0; P < 2; ++P) {
        const f32x2 dr = *(lds_f2*)(uintptr_t)(dr_b + 8u * ((uint32_t)((c - 1) & 1) * 32u + (uint32_t)ub[P]));
        const f32x2 dp2 = {dr.y, dr.y};
#pragma unroll
        for (int h = 0; h < EH; ++h) q2[P][h] = pk_fma(dp2, vs[SN][P][h], q2[P][h]);
        if (l5 == 16) *(lds_f*)(uintptr_t)(xs_b + 4u * (uint32_t)(ub[P] * KP + c - 1)) = dr.x;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    ATICK(3, q2[0][0].x);
    if (c + 1 < K) {
      gather(sn_c, panel_base(c + 1));
      reduce_post(sn_c, c + 1, true);
    }
    __builtin_amdgcn_sched_barrier(0);
    ATICK(4, q2[0][0].x);
    if (wave >= 2 * NS) {
      // panel c + 3 into the buffer column c's gathers (a column ago) have left; the pieces issued a column ago have had a
      // whole slot phase to land: wait for them here, ahead of the new ones, so the barrier below publishes panel c + 2
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (c + 3 < K) {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        float* dst = pan + (size_t)((c + 3) % 3) * PW;
        for (int ch = wave - 2 * NS; ch < chunks1; ch += NW - 2 * NS)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr)(dst + (size_t)ch * 256), 16, lane * 16,
                                                    (int)((uint32_t)(c + 3) * col_b + (uint32_t)ch * 1024u), 0, 0);
      }
    }
    if (wave < NS && c + 1 < K) sampler_prefetch(c + 1);
    ATICK(5, dprev_s);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ATICK(6, dprev_s);
  };
#ifdef X_NOLOOP
  for (int c = 0; c < 0; c += 2) {
#else
  for (int c = 0; c < K; c += 2) {
#endif
    step(c1t{}, c);
    if (c + 1 < K) step(c0t{}, c + 1);
  }
  // the last column's delta
  {
    f32x2 dl[2];
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      dl[P] = *(lds_f2*)(uintptr_t)(dr_b + 8u * ((uint32_t)((K - 1) & 1) * 32u + (uint32_t)ub[P]));
      if (l5 == 16) *(lds_f*)(uintptr_t)(xs_b + 4u * (uint32_t)(ub[P] * KP + K - 1)) = dl[P].x;
    }
    if ((K - 1) & 1) {
#pragma unroll
      for (int P = 0; P < 2; ++P)
#pragma unroll
        for (int h = 0; h < EH; ++h) q2[P][h] = pk_fma(f32x2{dl[P].y, dl[P].y}, vs[1][P][h], q2[P][h]);
    } else {
#pragma unroll
      for (int P = 0; P < 2; ++P)
#pragma unroll
        for (int h = 0; h < EH; ++h) q2[P][h] = pk_fma(f32x2{dl[P].y, dl[P].y}, vs[0][P][h], q2[P][h]);
    }
  }
#ifdef BNMTF_PHASE_TIMING
  if (blockIdx.x % 61 == 0 && lane == 0)
    printf("ahead block %d wave %d EM %d: prepass %llu setup %llu | sampler/fill %llu update %llu gather+reduce %llu dma+prefetch %llu barrier %llu (cycles, %d columns)\n",
           (int)blockIdx.x, wave, EM, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], K);
#endif
  // ------------------------------------------------------------ results
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int P = 0; P < 2; ++P) {
    // (unit indices are read again rather than kept in registers through the column loop)
    const int pair = blockIdx.x * kAheadPairsPerBlock + P * NW + wave;
    const bool on = pair < f.npairs && (int)f.pair_E[pair] <= EM;
    u[P] = on ? f.unit_map[2 * pair + half] : -1;
    valid[P] = u[P] >= 0;
    gi[P] = (uint32_t)a.n0 + (uint32_t)(valid[P] ? u[P] : 0);
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) {
      const int kk = l5 + 32 * nx;
      x[P][nx] = lds[L.xs + ub[P] * KP + kk];
      if (valid[P] && kk < K) a.Xself[(size_t)gi[P] * KP + kk] = x[P][nx];
    }
  }
  if (f.stats) {                      // per-block partial sums -> slab, summed by finish_kernel
    double* red = reinterpret_cast<double*>(pan);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // panels are dead from here on
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      double px = 0.0, sq = 0.0, sq2 = 0.0;
#pragma unroll
      for (int nx = 0; nx < NX; ++nx) {
        float s = 0.f;
        if (valid[P]) s = slab_sum_ordered(a.slabs, a.split, (size_t)a.n_pad * KP, (size_t)u[P] * KP + l5 + 32 * nx);
        px += (double)s * (double)x[P][nx];
      }
#pragma unroll
      for (int h = 0; h < EH; ++h) {
        const double qa = (double)q2[P][h].x, qb = (double)q2[P][h].y;
        sq += qa + qb; sq2 += qa * qa + qb * qb;
      }
      px = half_sum_d(px); sq = half_sum_d(sq); sq2 = half_sum_d(sq2);
      if (l5 == 0) { red[ub[P] * 3 + 0] = valid[P] ? px : 0.0; red[ub[P] * 3 + 1] = sq; red[ub[P] * 3 + 2] = sq2; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < 3) {
      double s = 0.0;
      for (int w = 0; w < 4 * NW; ++w) s += red[w * 3 + tid];
      f.stats[(size_t)blockIdx.x * 4 + tid] = s;
    }
  }
}

template <int NX, int MODE>
__global__ __launch_bounds__(kAheadWaves * 64, 1) void sweep_ahead_kernel(SweepArgs a, FastArgs f) {
  extern __shared__ float lds[];
  const int wv = (int)(threadIdx.x >> 6);
  int e0 = 0;
#pragma unroll
  for (int P = 0; P < 2; ++P) {
    const int pr = blockIdx.x * kAheadPairsPerBlock + P * kAheadWaves + wv;
    const int e = __builtin_amdgcn_readfirstlane(pr < f.npairs ? (int)f.pair_E[pr] : 0);
    if (e <= kWideMaxSlots && e > e0) e0 = e;
  }
#ifdef AHEAD_ONLY_EM
  (void)e0; sweep_ahead_body<AHEAD_ONLY_EM, NX, MODE>(a, f, lds); return;
#endif
  if (e0 <= 16) sweep_ahead_body<16, NX, MODE>(a, f, lds);
  else if (e0 <= 24) sweep_ahead_body<24, NX, MODE>(a, f, lds);
  else if (e0 <= 28) sweep_ahead_body<28, NX, MODE>(a, f, lds);
  else sweep_ahead_body<32, NX, MODE>(a, f, lds);
}

bool sweep_ahead_supported(int KP, int pw) { return (size_t)ahead_lds(KP, pw).total * sizeof(float) <= 160 * 1024; }

template <int NX, int MODE>
static void launch_ahead_inst(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  static std::atomic<uint64_t> lds_ok{0};
  const int nblocks = (f.npairs + kAheadPairsPerBlock - 1) / kAheadPairsPerBlock;
  if (nblocks > 0 && allow_full_lds((const void*)sweep_ahead_kernel<NX, MODE>, lds_ok))
    hipLaunchKernelGGL((sweep_ahead_kernel<NX, MODE>), dim3(nblocks), dim3(kAheadWaves * 64), (size_t)ahead_lds(a.KP, f.pw).total * sizeof(float), st, a, f);
}

// f describes the pairs of the 16-wave layout (16 pairs per block, at most kWideMaxSlots slots each)
void launch_sweep_ahead(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  const int nx = a.KP / 32;
  if (a.mode == kSweepDraw) { if (nx == 1) launch_ahead_inst<1, kSweepDraw>(a, f, st); else launch_ahead_inst<2, kSweepDraw>(a, f, st); }
  else                      { if (nx == 1) launch_ahead_inst<1, kSweepMode>(a, f, st); else launch_ahead_inst<2, kSweepMode>(a, f, st); }
}

}  // namespace bnmtf
