// K3 "turns": the on-chip half sweep with two groups of units per block taking turns (round 3).
//
// The sweep of sweep_chip.inc runs a column as  slot work -> reduce -> barrier -> sampler -> barrier : while the two
// sampler waves walk the ~50 dependent instructions of a draw, the other fourteen waves of the block idle (38 % of a
// column by the phase stamps).  The rows of a factor are independent given the other factor, so here a block's 32
// units form two groups, A and B, and EVERY wave holds a pair of each (four units, up to 256 registers, eight waves per
// block).  A column is two half steps: A's slot work (apply the previous draw, gather column c, post the numerator's
// sum) -- barrier -- B's slot work -- barrier; the draw of A's column c is made by one wave WHILE everybody does B's
// slot work, and is first needed when A's turn comes again, a whole half step later (and the other way round).  No
// wave ever waits for the sampler; what the draw costs is its ~55 instructions on one wave.
// Unlike the "two groups of waves half a column out of phase" tried in round 2 (slower: a group's waves idled during
// the other group's turn, and a slot phase did not get shorter with half the waves), the groups here alternate INSIDE
// each wave, so all eight waves work all the time and nothing is parked.
//
// What else differs from sweep_chip.inc:
//  * asq_c = sum_miss v_c^2 does not depend on the chain: the pre-pass (which reads every pair panel anyway) forms it for
//    every column beside q, so the column loop has one sum (sum q v) and one half-wave reduction per unit;
//  * the units' x rows, tau P - lambda and asq live in LDS: the sampler lane (unit, candidate) reads what it needs itself,
//    the unit waves post ONE number per unit and column:  A = sum_miss q v_c - sum_l x_l C0_{l,c}  (C0 = the Gram with a
//    zero diagonal), and  numer_c = pl_c - tau x_c asq_c + tau A ,  tau_p = tau (C_cc - asq_c);
//  * candidates (Philox + log / sqrt / cos) are made a half step ahead by another wave, the LDS-DMA of the next panel is
//    issued behind a wave's own gathers (while it waits for them anyway).
//
// Same arithmetic as the reference's column update (bnmf_gibbs_optimised.py:134-142, 167-177), same candidate
// sequence as oracle/rng.py; the order of the floating-point sums differs from sweep_chip.inc, so a problem is run by
// one of the two bodies throughout (api.hip picks per direction).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "sweep_common.h"

namespace bnmtf {

constexpr int kTurnsWaves = 8;              // waves per block; wave w holds pair w (group A) and pair 8 + w (group B)
constexpr int kTurnsPairs = 2 * kTurnsWaves;
constexpr int kTurnsCands = 4;
constexpr int kTurnsPanelStride = 9216;     // floats between the two single-column panel buffers (>= pw): a ds_read immediate

struct TurnsLds { int C0, pan, xs, pls, tab, ax, dr, asq, Cd, total; };      // float offsets
__host__ __device__ inline TurnsLds turns_lds(int KP, int pw) {
  TurnsLds L;
  L.C0 = 0;                                  // [KP][KP] Gram of the other factor, diagonal zeroed
  L.pan = KP * KP;                           // pre-pass: two pair panels (4 pw) | column loop: two single panels ...
  L.xs = L.pan + kTurnsPanelStride + pw;     // ... and, behind them, what only the column loop needs: x [32][KP]
  L.pls = L.xs + 32 * KP;                    // tau P - lambda [32][KP]
  L.tab = L.pls + 32 * KP;                   // [2 groups][64] float4 candidates (nl, z, u2, -)
  L.ax = L.tab + 2 * 64 * 4;                 // [32] A
  L.dr = L.ax + 32;                          // [2][32] (draw, delta)
  const int main_end = L.dr + 2 * 32 * 2, pre_end = L.pan + 4 * pw;
  L.asq = main_end > pre_end ? main_end : pre_end;   // [32][KP] sum_miss v_c^2: written by the pre-pass, read by the sampler
  L.Cd = L.asq + 32 * KP;                    // [KP] diagonal of the Gram
  L.total = L.Cd + KP;
  return L;
}

#ifdef BNMTF_PHASE_TIMING
#define TTICK(i, dep) do { const unsigned long long t_ = tick(dep); ph[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define TTICK(i, dep) do { } while (0)
#endif

template <int EM, int NX, int MODE>
__device__ __forceinline__ void sweep_turns_body(const SweepArgs& a, const FastArgs& f, float* lds) {
  constexpr int KP = NX * 32, EH = EM / 2, NC = kTurnsCands, NW = kTurnsWaves;
  static_assert(EM % 2 == 0 && NC == 4, "slots in pairs, candidates in quads");
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) f32x2 lds_f2;
  typedef __attribute__((address_space(3))) f32x4 lds_f4;
  typedef __attribute__((address_space(3))) float lds_f;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int PW = f.pw;
  const TurnsLds L = turns_lds(KP, PW);
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

  // group G = 0 (A), 1 (B): pair blockIdx * 16 + G * 8 + wave ; unit-in-block ub = 16 G + 2 wave + half
  int u[2];
  bool valid[2];
  uint32_t gi[2];
  const int ub0 = 2 * wave + half;                 // group A's unit; group B's is ub0 + 16
  uint32_t addr[2][EM];                            // slot addresses as LDS BYTE addresses inside panel buffer 0 (sentinel: a zero word on bank l5)
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    const int pair = blockIdx.x * kTurnsPairs + G * NW + wave;
    const bool on = pair < f.npairs && (int)f.pair_E[pair] <= EM;
    const uint32_t base = on ? f.pair_base[pair] : 0u;
    const int E = on ? (int)f.pair_E[pair] : 0;
    u[G] = on ? f.unit_map[2 * pair + half] : -1;
    valid[G] = u[G] >= 0;
    gi[G] = (uint32_t)a.n0 + (uint32_t)(valid[G] ? u[G] : 0);
    const uint32_t sent = (uint32_t)(f.mz + l5);
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      const uint32_t w = (2 * h < E) ? f.off16[((size_t)(base >> 1) + h) * 64 + lane] : (sent | (sent << 16));
      addr[G][2 * h] = pan_b + 4u * (w & 0xFFFFu);
      addr[G][2 * h + 1] = pan_b + 4u * (w >> 16);
    }
  }
  // x = the units' rows of the factor, pl = tau P - lambda (P = the contraction's slabs summed): lane l5 holds columns l5, l5 + 32
  float x[2][NX], pl[2][NX];
#pragma unroll
  for (int G = 0; G < 2; ++G)
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) {
      const int kk = l5 + 32 * nx;
      float s = 0.f;
      if (valid[G]) s = slab_sum_ordered(a.slabs, a.split, (size_t)a.n_pad * KP, (size_t)u[G] * KP + kk);
      x[G][nx] = valid[G] ? a.Xself[(size_t)gi[G] * KP + kk] : 0.f;
      pl[G][nx] = valid[G] ? fmaf(tau, s, -a.lambda[(size_t)u[G] * KP + kk]) : 0.f;
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
    for (int G = 0; G < 2; ++G)
#pragma unroll
      for (int h = 0; h < EH; ++h) { accA[G][h] = f32x2{0.f, 0.f}; accB[G][h] = f32x2{0.f, 0.f}; }
    const int chunks2 = (2 * PW) / 256;
    const uint32_t stride_b = (uint32_t)f.ld2_o * 8u;
    const __amdgpu_buffer_rsrc_t rs2 = panel_rsrc(f.XoT2, (size_t)(KP / 2) * f.ld2_o * 8);
    stage_panel_buf<NW>(rs2, 0u, pan, chunks2, wave, lane * 16);
    sync_with_dma();
    const int npair = KP / 2;
    for (int kp = 0; kp < npair; ++kp) {
      // element j of a pair panel sits 8 j bytes in: 2 * addr - pan_b (+ the buffer's offset)
      const uint32_t boff = (uint32_t)((kp & 1) * 2 * PW) * 4u - pan_b;
      const int k0 = 2 * kp, k1 = 2 * kp + 1;
      int ch = wave;                           // this wave's pieces of the next pair panel, issued between its gathers
      float* nxt = pan + (size_t)((kp + 1) & 1) * 2 * PW;
      const uint32_t nxt_off = (uint32_t)(kp + 1) * stride_b;
#if defined(TURNS_PRE_NO_DMA)
      const bool more = false;       // experiment only (wrong results): what does the pre-pass cost without its staging?
#elif defined(TURNS_PRE_DMA_TOP)
      const bool more = false;
      if (kp + 1 < npair) stage_panel_buf<NW>(rs2, nxt_off, nxt, chunks2, wave, lane * 16);
#else
      const bool more = kp + 1 < npair;
#endif
#pragma unroll
      for (int G = 0; G < 2; ++G) {
        // both registers are read and the choice is made on the broadcast values: a select between x[G][0] and x[G][1]
        // itself turns the array into an indexed stack object
        f32x2 x01 = {half_bcast(x[G][0], k0 & 31, half), half_bcast(x[G][0], k1 & 31, half)};
        if (NX == 2) {
          const f32x2 xhi = {half_bcast(x[G][NX - 1], k0 & 31, half), half_bcast(x[G][NX - 1], k1 & 31, half)};
          x01 = k0 >= 32 ? xhi : x01;
        }
        f32x2 vv = {0.f, 0.f};
        // groups of four slot pairs, software pipelined by hand: the gathers of group g + 1 are issued, then group g is
        // accumulated.  The empty asm statements are ordered among themselves and pin that shape -- left alone the compiler
        // either hoists every gather of the panel to the top (two registers per slot) or sinks the accumulation below the barrier.
        constexpr int GH = 4, NG = (EH + GH - 1) / GH;
        f32x2 va[2][GH], vb[2][GH];
        auto issue = [&](int g, int set) {
#pragma unroll
          for (int t = 0; t < GH; ++t) {
            const int h = g * GH + t;
            if (h < EH) {
              asm volatile("" : "+v"(addr[G][2 * h]), "+v"(addr[G][2 * h + 1]));
              va[set][t] = *(lds_cf2*)(uintptr_t)(2u * addr[G][2 * h] + boff);
              vb[set][t] = *(lds_cf2*)(uintptr_t)(2u * addr[G][2 * h + 1] + boff);
            }
          }
        };
        issue(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g + 1 < NG) issue(g + 1, (g + 1) & 1);
          if (more && ch < chunks2 && g < NG - 1) {            // one piece per group: the texture path is never asked for a burst
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_ptr)(nxt + (size_t)ch * 256), 16, lane * 16, (int)(nxt_off + (uint32_t)ch * 1024u), 0, 0);
            ch += NW;
          }
#pragma unroll
          for (int t = 0; t < GH; ++t) {
            const int h = g * GH + t;
            if (h < EH) {
              accA[G][h] = pk_fma(va[g & 1][t], x01, accA[G][h]);
              accB[G][h] = pk_fma(vb[g & 1][t], x01, accB[G][h]);
#ifndef TURNS_PRE_NO_VV
              vv = pk_fma(va[g & 1][t], va[g & 1][t], vv);
              vv = pk_fma(vb[g & 1][t], vb[g & 1][t], vv);
#endif
              asm volatile("" : "+v"(accA[G][h]), "+v"(accB[G][h]));
            }
          }
        }
        const float s0 = half_sum_upper(vv.x), s1 = half_sum_upper(vv.y);
        if (l5 == 16) *(lds_f2*)(uintptr_t)(asq_b + 4u * (uint32_t)((ub0 + 16 * G) * KP + k0)) = f32x2{s0, s1};
      }
      for (; more && ch < chunks2; ch += NW)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_ptr)(nxt + (size_t)ch * 256), 16, lane * 16, (int)(nxt_off + (uint32_t)ch * 1024u), 0, 0);
      sync_with_dma();
    }
#pragma unroll
    for (int G = 0; G < 2; ++G)
#pragma unroll
      for (int h = 0; h < EH; ++h) q2[G][h] = f32x2{accA[G][h].x + accA[G][h].y, accB[G][h].x + accB[G][h].y};
  }
  TTICK(0, q2[0][0].x);

  // ------------------------------------------------------------ column-loop state in LDS, the first two panels, column 0's candidates
  const int chunks1 = PW / 256;
  const __amdgpu_buffer_rsrc_t rs1 = panel_rsrc(f.XoT, (size_t)KP * f.ldT_o * 4);
  const uint32_t col_b = (uint32_t)f.ldT_o * 4u;
  for (int p = 0; p < 2 && p < K; ++p) stage_panel_buf<NW>(rs1, (uint32_t)p * col_b, pan + (size_t)p * kTurnsPanelStride, chunks1, wave, lane * 16);
#pragma unroll
  for (int G = 0; G < 2; ++G)
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) {
      lds[L.xs + (ub0 + 16 * G) * KP + l5 + 32 * nx] = x[G][nx];
      lds[L.pls + (ub0 + 16 * G) * KP + l5 + 32 * nx] = pl[G][nx];
    }
  // duties: wave 0 draws (one (unit, candidate) per lane: the sixteen units of the group whose sum was posted last), wave 1
  // makes the candidates of the group whose turn it is
  const int s_u = lane >> 2, s_cand = lane & 3;           // unit within a group, candidate
  bool s_valid[2] = {false, false};
  uint32_t s_row[2] = {(uint32_t)a.n0, (uint32_t)a.n0};
  if (wave < 2) {
#pragma unroll
    for (int G = 0; G < 2; ++G) {
      const int spr = blockIdx.x * kTurnsPairs + G * NW + (s_u >> 1);
      const int su = (spr < f.npairs && (int)f.pair_E[spr] <= kWideMaxSlots) ? f.unit_map[2 * spr + (s_u & 1)] : -1;
      s_valid[G] = su >= 0;
      s_row[G] += (uint32_t)(s_valid[G] ? su : 0);
    }
  }
  auto fill_tab = [&](int G, int col) {
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
    if (MODE == kSweepDraw) {
      uint32_t row = G ? s_row[1] : s_row[0];
      asm volatile("" : "+v"(row));                            // opaque: no partial rounds of this call are kept across columns
      const U4 r = philox4x32_10(row, (uint32_t)col, a.it, a.stream + 16u * (uint32_t)s_cand, a.key0, a.key1);
      const TnCand cd = tn_cand_pre(r.x, r.y);
      e = f32x4{cd.nl, cd.z, cd.sw, 0.f};
    }
    *(lds_f4*)(uintptr_t)(tab_b + 16u * (uint32_t)(G * 64 + lane)) = e;
  };
  auto sampler_draw = [&](int G, int c) {
    const uint32_t un = (uint32_t)(16 * G + s_u);
    const float ax = *(lds_f*)(uintptr_t)(ax_b + 4u * un);
    const f32x4 ce = *(lds_f4*)(uintptr_t)(tab_b + 16u * (uint32_t)(G * 64 + lane));
    const float plc = *(lds_f*)(uintptr_t)(pls_b + 4u * (un * KP + (uint32_t)c));
    const float xo = *(lds_f*)(uintptr_t)(xs_b + 4u * (un * KP + (uint32_t)c));
    const float asq = *(lds_f*)(uintptr_t)(asq_b + 4u * (un * KP + (uint32_t)c));
    const float cdiag = *(lds_f*)(uintptr_t)(cd_b + 4u * (uint32_t)c);
    const float taup = tau * (cdiag - asq);
    const float numer = fmaf(tau, ax, fmaf(-tau * xo, asq, plc));
    const bool sv = G ? s_valid[1] : s_valid[0];
    float r = 0.f;
    if (MODE == kSweepDraw) {
      const TnFast tf = tn_fast_params(numer, taup);
      TnCand cand = {ce.x, ce.y, ce.z};
      bool need = sv && tf.live;
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
        uint32_t row = G ? s_row[1] : s_row[0];
        asm volatile("" : "+v"(row));                          // opaque: nothing of this Philox call is hoisted out of the column loop
        const U4 ph4 = philox4x32_10(row, (uint32_t)c, a.it, a.stream + 16u * (cbase + (uint32_t)s_cand), a.key0, a.key1);
        cand = tn_cand_pre(ph4.x, ph4.y);
      }
    } else {
      const float mu = numer / taup;
      r = fmaxf((sv && taup > 0.f && mu > 0.f) ? mu : 0.f, a.min_x);
    }
    if (s_cand == 0) *(lds_f2*)(uintptr_t)(dr_b + 8u * ((uint32_t)(c & 1) * 32u + un)) = f32x2{r, r - xo};
  };

  f32x2 vs[2][EH];                          // gathered values of the group's current column [group][slots 2h, 2h+1]
#pragma unroll
  for (int G = 0; G < 2; ++G)
#pragma unroll
    for (int h = 0; h < EH; ++h) vs[G][h] = f32x2{0.f, 0.f};
  if (wave == 1) fill_tab(0, 0);
  sync_with_dma();                            // panels 0 and 1, x, pl, asq, candidates of (A, 0)
  TTICK(1, q2[0][0].x);

  // One half step: group G's turn at column c (panel buffer BUF = c & 1).  Meanwhile wave 0 draws the other group's pending
  // column (pc; -1: none) and wave 1 makes the candidates the NEXT half step's draw (this group, column c) will take.
  auto half_step = [&](auto g_c, auto buf_c, int c, int pc) {
    constexpr int G = decltype(g_c)::value, BUF = decltype(buf_c)::value;
    const uint32_t un = (uint32_t)(ub0 + 16 * G);
#ifndef TURNS_DUTY_LATE
    if (wave == 0) { if (pc >= 0) sampler_draw(1 - G, pc); }
    else if (wave == 1) { if (!(G == 0 && c == 0)) fill_tab(G, c); }
    __builtin_amdgcn_sched_barrier(0);
#endif
    TTICK(2, q2[0][0].x);
    if (c >= 1) {
      // this group's draw of column c - 1: q += delta v_{c-1}; the x row in LDS follows (the C term below wants it current)
      const f32x2 dr = *(lds_f2*)(uintptr_t)(dr_b + 8u * ((uint32_t)((c - 1) & 1) * 32u + un));
      const f32x2 dp2 = {dr.y, dr.y};
#pragma unroll
      for (int h = 0; h < EH; ++h) q2[G][h] = pk_fma(dp2, vs[G][h], q2[G][h]);
      if (l5 == 16) *(lds_f*)(uintptr_t)(xs_b + 4u * (un * KP + (uint32_t)(c - 1))) = dr.x;
    }
    __builtin_amdgcn_sched_barrier(0);
    TTICK(3, q2[0][0].x);
    // gather v_c: address register + immediate, no VALU
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      asm volatile("" : "+v"(addr[G][2 * h]), "+v"(addr[G][2 * h + 1]));   // opaque: keeps addr + const from being hoisted into registers
      vs[G][h].x = *(lds_cf*)(uintptr_t)(addr[G][2 * h] + (uint32_t)(BUF * kTurnsPanelStride * 4));
      vs[G][h].y = *(lds_cf*)(uintptr_t)(addr[G][2 * h + 1] + (uint32_t)(BUF * kTurnsPanelStride * 4));
    }
#ifdef TURNS_DUTY_LATE
    // the duties sit HERE, behind the wave's own gathers: those take the LDS a few hundred cycles to serve (all eight waves
    // gather at once), which is when the draw's dependent chain (wave 0) and the Philox rounds (wave 1) cost nothing
    __builtin_amdgcn_sched_barrier(0);
    if (wave == 0) { if (pc >= 0) sampler_draw(1 - G, pc); }
    else if (wave == 1) { if (!(G == 0 && c == 0)) fill_tab(G, c); }
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (G == 0 && c + 1 < K && c >= 1) {
      // the next column's panel, into the buffer column c - 1 has left (its last gathers were B's, a half step ago): issued
      // here, behind the wave's own gathers, while it would wait for them anyway; it has this half step's rest and all of
      // B's turn to land (the barrier that ends B's turn waits for it: sync_with_dma)
      float* dst = pan + (size_t)(1 - BUF) * kTurnsPanelStride;
      for (int ch = wave; ch < chunks1; ch += NW)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr)(dst + (size_t)ch * 256), 16, lane * 16, (int)((uint32_t)(c + 1) * col_b + (uint32_t)ch * 1024u), 0, 0);
    }
    // A = sum q v_c - sum_l x_l C0_{l,c}
    f32x2 s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int h = 0; h < EH; ++h) s2[h & 1] = pk_fma(q2[G][h], vs[G][h], s2[h & 1]);
    float s_t = (s2[0].x + s2[0].y) + (s2[1].x + s2[1].y);
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) s_t = fmaf(-*(lds_f*)(uintptr_t)(xs_b + 4u * (un * KP + (uint32_t)(l5 + 32 * nx))), Cs[c * KP + l5 + 32 * nx], s_t);
    s_t = half_sum_upper(s_t);
    if (l5 == 16) *(lds_f*)(uintptr_t)(ax_b + 4u * un) = s_t;
    __builtin_amdgcn_sched_barrier(0);
    TTICK(4, s_t);
    if (G == 1) sync_with_dma();              // B's turn ends the column: the next panel has landed
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TTICK(5, s_t);
  };
  using g0 = std::integral_constant<int, 0>;
  using g1 = std::integral_constant<int, 1>;
  for (int c = 0; c < K; c += 2) {
    half_step(g0{}, g0{}, c, c - 1);          // A's turn at c; B's column c - 1 is drawn meanwhile
    half_step(g1{}, g0{}, c, c);              // B's turn at c; A's column c is drawn meanwhile
    if (c + 1 < K) {
      half_step(g0{}, g1{}, c + 1, c);
      half_step(g1{}, g1{}, c + 1, c + 1);
    }
  }
  // B's last column is still to be drawn
  if (wave == 0) sampler_draw(1, K - 1);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // the last column's delta
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    const uint32_t un = (uint32_t)(ub0 + 16 * G);
    const f32x2 dl = *(lds_f2*)(uintptr_t)(dr_b + 8u * ((uint32_t)((K - 1) & 1) * 32u + un));
    if (l5 == 16) *(lds_f*)(uintptr_t)(xs_b + 4u * (un * KP + (uint32_t)(K - 1))) = dl.x;
#pragma unroll
    for (int h = 0; h < EH; ++h) q2[G][h] = pk_fma(f32x2{dl.y, dl.y}, vs[G][h], q2[G][h]);
  }
#ifdef BNMTF_PHASE_TIMING
  if (blockIdx.x % 61 == 0 && lane == 0)
    printf("turns block %d wave %d EM %d: prepass %llu setup %llu | duty %llu update %llu gather+reduce %llu barrier %llu (cycles, %d columns)\n",
           (int)blockIdx.x, wave, EM, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], K);
#endif
  // ------------------------------------------------------------ results
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    // (unit indices are read again rather than kept in registers through the column loop)
    const int pair = blockIdx.x * kTurnsPairs + G * NW + wave;
    const bool on = pair < f.npairs && (int)f.pair_E[pair] <= EM;
    u[G] = on ? f.unit_map[2 * pair + half] : -1;
    valid[G] = u[G] >= 0;
    gi[G] = (uint32_t)a.n0 + (uint32_t)(valid[G] ? u[G] : 0);
#pragma unroll
    for (int nx = 0; nx < NX; ++nx) {
      const int kk = l5 + 32 * nx;
      x[G][nx] = lds[L.xs + (ub0 + 16 * G) * KP + kk];
      if (valid[G] && kk < K) a.Xself[(size_t)gi[G] * KP + kk] = x[G][nx];
    }
  }
  if (f.stats) {                      // per-block partial sums -> slab, summed by finish_kernel
    double* red = reinterpret_cast<double*>(pan);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // panels are dead from here on
#pragma unroll
    for (int G = 0; G < 2; ++G) {
      double px = 0.0, sq = 0.0, sq2 = 0.0;
#pragma unroll
      for (int nx = 0; nx < NX; ++nx) {
        float s = 0.f;
        if (valid[G]) s = slab_sum_ordered(a.slabs, a.split, (size_t)a.n_pad * KP, (size_t)u[G] * KP + l5 + 32 * nx);
        px += (double)s * (double)x[G][nx];
      }
#pragma unroll
      for (int h = 0; h < EH; ++h) {
        const double qa = (double)q2[G][h].x, qb = (double)q2[G][h].y;
        sq += qa + qb; sq2 += qa * qa + qb * qb;
      }
      px = half_sum_d(px); sq = half_sum_d(sq); sq2 = half_sum_d(sq2);
      const int un = ub0 + 16 * G;
      if (l5 == 0) { red[un * 3 + 0] = valid[G] ? px : 0.0; red[un * 3 + 1] = sq; red[un * 3 + 2] = sq2; }
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
__global__ __launch_bounds__(kTurnsWaves * 64, 1) void sweep_turns_kernel(SweepArgs a, FastArgs f) {
  extern __shared__ float lds[];
  const int wv = (int)(threadIdx.x >> 6);
  int e0 = 0;
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    const int pr = blockIdx.x * kTurnsPairs + G * kTurnsWaves + wv;
    const int e = __builtin_amdgcn_readfirstlane(pr < f.npairs ? (int)f.pair_E[pr] : 0);
    if (e <= kWideMaxSlots && e > e0) e0 = e;
  }
#ifdef TURNS_ONLY_EM
  (void)e0; sweep_turns_body<TURNS_ONLY_EM, NX, MODE>(a, f, lds); return;
#endif
  if (e0 <= 16) sweep_turns_body<16, NX, MODE>(a, f, lds);
  else if (e0 <= 24) sweep_turns_body<24, NX, MODE>(a, f, lds);
  else if (e0 <= 28) sweep_turns_body<28, NX, MODE>(a, f, lds);
  else if (e0 <= 30) sweep_turns_body<30, NX, MODE>(a, f, lds);
  else sweep_turns_body<32, NX, MODE>(a, f, lds);      // (this class spills a few registers: rows with more than 960 missing entries)
}

bool sweep_turns_supported(int KP, int pw) { return pw <= kTurnsPanelStride && (size_t)turns_lds(KP, pw).total * sizeof(float) <= 160 * 1024; }

template <int NX, int MODE>
static void launch_turns_inst(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  static std::atomic<uint64_t> lds_ok{0};
  const int nblocks = (f.npairs + kTurnsPairs - 1) / kTurnsPairs;
  if (nblocks > 0 && allow_full_lds((const void*)sweep_turns_kernel<NX, MODE>, lds_ok))
    hipLaunchKernelGGL((sweep_turns_kernel<NX, MODE>), dim3(nblocks), dim3(kTurnsWaves * 64), (size_t)turns_lds(a.KP, f.pw).total * sizeof(float), st, a, f);
}

// f describes the pairs of the 16-wave layout (16 pairs per block, at most kWideMaxSlots slots each)
void launch_sweep_turns(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  const int nx = a.KP / 32;
  if (a.mode == kSweepDraw) { if (nx == 1) launch_turns_inst<1, kSweepDraw>(a, f, st); else launch_turns_inst<2, kSweepDraw>(a, f, st); }
  else                      { if (nx == 1) launch_turns_inst<1, kSweepMode>(a, f, st); else launch_turns_inst<2, kSweepMode>(a, f, st); }
}

}  // namespace bnmtf
