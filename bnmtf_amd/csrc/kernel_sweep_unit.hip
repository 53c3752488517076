// K3u (round 6): the on-chip half sweep with ONE UNIT PER 64-LANE WAVE -- the shape for few units per CU: the shards of a
// multi-GPU run (8192^2 over 8 ranks: 1 024 units = four per CU) and small single-GPU problems.
//
// Why a shape of its own.  With four units on a CU nothing can be amortised over units: sweep_chip.inc's 2-wave blocks (two
// 32-lane units per wave, a staging wave) spend ~3 200 cycles per column -- ~300 instructions of a wave that is alone on its
// SIMD and retires a dependent instruction every ~8-10 cycles -- and ~25 % of a half sweep in a pre-pass that pulls every pair
// panel of the other factor through the 64 B/clk LDS-DMA path for FOUR units.  Here:
//  * a unit owns the whole wave: half the slots per lane (13 instead of 26 at 10 % missing of 8192), and everything that is one
//    value per unit -- x_k, the conditional's numerator and precision, the draw -- is wave-UNIFORM: one readlane into a scalar
//    register instead of per-half selects, one candidate evaluation, no cross-wave exchange, no second barrier.
//  * latent factor k lives in lane k (K <= 64): one register for x, one for tau P - lambda, the K x K term of a column is one
//    LDS read and one FMA per lane.
//  * q = U_i . V_j on the unit's missing entries is rebuilt per sweep straight from the other factor's row-major copy (each lane
//    reads the 256-byte rows of its own entries out of L2: 13 slots x 64 FMAs) -- 213 KB per unit instead of 2.1 MB of pair
//    panels per block, ~5 k cycles instead of ~30 k.
//  * the column panels go through THREE LDS buffers, staged by a wave that does nothing else, two panels in flight: a 33 KiB
//    panel lands ~1 100 cycles after its first piece is issued (tools/dma_probe.hip: ~550 + bytes / 61), a column takes less.
//    The third buffer is beyond the 64 KiB a ds_read immediate spans, so the slot addresses are held twice (26 registers of
//    the 512 a lone wave has).
// Slot layout: the host's pair layout (api.hip build_dir) with both 32-lane halves of a pair belonging to the SAME unit: lanes l
// and l + 32 share residue class l mod 32 (each half of a ds_read_b32 touches 32 distinct banks), the class's entries dealt to
// them in turn.  The order in which a unit's partial sums are added therefore differs from the 32-lane shapes': the chain agrees
// with theirs to fp32 rounding (and, in draw mode, up to rejections that rounding flips), not bit for bit; it IS bit for bit the
// same for every launch geometry of THIS shape (a unit's arithmetic depends on its own missing list only), so every rank of an
// N-rank run and the single-rank run of the same shape draw the same chain.
// Reference: the for-k loops of bnmf_gibbs_optimised.py:134-142 (and nmf_icm.py:124-134 in the mode update).
#include <algorithm>
#include <cstdlib>

#include "sweep_common.h"

namespace bnmtf {

constexpr int kUnitCands = 4;                  // candidates of a draw held in the LDS table (further ones: 64 at a time)
constexpr int kUnitPanelStride = 9216;         // floats between panel buffers (>= pw)
constexpr int kUnitPieces = kUnitPanelStride / 256;     // LDS-DMA pieces the staging wave issues per panel (a constant: its wait counts are immediates)

__host__ __device__ inline size_t sweep_unit_lds_floats(int KP, int nu) {
  return (size_t)KP * KP + 3 * (size_t)kUnitPanelStride + (size_t)nu * KP * kUnitCands * 4 + 64 + (size_t)nu * 64;
}

template <int O>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {      // lane O of every quad to its four lanes (quad_perm [O,O,O,O])
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, O * 0x55, 0xF, 0xF, true);
}

template <int EM, int MODE, int NU>
__device__ __forceinline__ void sweep_unit_body(const SweepArgs& a, const FastArgs& f, float* lds) {
  static_assert(EM % 2 == 0, "slots are processed in pairs");
  constexpr int EH = EM / 2, NC = kUnitCands, S = kUnitPanelStride;
  const int KP = a.KP, K = a.K;
  float* Cs = lds;                                // [KP][KP]
  float* pan = lds + KP * KP;                     // three panel buffers, S floats apart
  typedef float f32x4t __attribute__((ext_vector_type(4)));
  f32x4t* tab = reinterpret_cast<f32x4t*>(pan + 3 * S);        // [NU][KP][NC] candidates: (nl, z, sw, -) = tn_cand_pre of the Philox words
  float* red = pan + 3 * S + (size_t)NU * KP * NC * 4;         // [64] end-of-sweep partial sums, then [NU][64] the waves' copies of x
  const uint32_t pan_b = (uint32_t)(uintptr_t)(lds_fp)pan;
  const int tid = threadIdx.x, lane = tid & 63, l5 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t rs1 = panel_rsrc(f.XoT, (size_t)KP * f.ldT_o * 4);
  const int chunks = f.pw / 256;

  if (wave == NU) {
    // ---------------------------------------------------------------- the staging wave
    // panel k goes to buffer k mod 3; it is issued when column k - 3 is done (barrier B_{k-3}) and has to be there when column
    // k starts (barrier B_{k-1}): two columns to land.  Every panel is kUnitPieces pieces (the last ones repeat the panel's last
    // piece: same bytes to the same place), so "panel k has landed, panel k + 1 may still be in flight" is s_waitcnt vmcnt(kUnitPieces).
    auto issue = [&](int k) {
      float* dst = pan + (size_t)(k % 3) * S;
      const uint32_t off = (uint32_t)k * (uint32_t)f.ldT_o * 4u;
      typedef __attribute__((address_space(3))) void* lds_ptr;
      for (int c = 0; c < kUnitPieces; ++c) {
        const int cc = c < chunks ? c : chunks - 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr)(dst + (size_t)cc * 256), 16, lane * 16, (int)(off + (uint32_t)cc * 1024u), 0, 0);
      }
    };
    // (the wait counter holds 63 at most: the issue of a panel blocks while more than that are outstanding -- back-pressure, and
    // never more than "the newest panel" behind a vmcnt(kUnitPieces))
    static_assert(kUnitPieces <= 63, "a panel's pieces must fit the vector-memory wait counter");
    issue(0);
    if (K > 1) { issue(1); asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kUnitPieces) : "memory"); }      // B_start: panel 0 there
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (K > 2) issue(2);
    for (int k = 0; k < K; ++k) {
      // B_k: column k is done with buffer k mod 3; column k + 1 needs panel k + 1: only panel k + 2 may still be in flight
      if (k + 2 < K) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kUnitPieces) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (k + 3 < K) issue(k + 3);
    }
    if (f.stats) asm volatile("s_barrier" ::: "memory");
    return;
  }

  // -------------------------------------------------------------------- a unit wave
  const int pair = blockIdx.x * NU + wave;
  const bool wave_on = pair < f.npairs && (int)f.pair_E[pair] <= kUnitMaxSlots;
  const uint32_t base = wave_on ? f.pair_base[pair] : 0u;
  const int E = wave_on ? (int)f.pair_E[pair] : 0;
  const int u = wave_on ? f.unit_map[2 * pair] : -1;
  const bool valid = u >= 0;
  const uint32_t gi = (uint32_t)a.n0 + (uint32_t)(valid ? u : 0);
  const float tau = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *a.tau)));
  const bool col_lane = lane < KP;                       // lane k holds latent factor k
#ifdef BNMTF_PHASE_TIMING
  const unsigned long long t_start = tick(0.f);
#endif

  float Pk = 0.f;                                        // the contraction's row of this unit (kept for the statistics)
  if (valid && col_lane) Pk = slab_sum_ordered(a.slabs, a.split, (size_t)a.n_pad * KP, (size_t)u * KP + lane);
  float x = (valid && col_lane) ? a.Xself[(size_t)gi * KP + lane] : 0.f;
  const float pl = (valid && col_lane) ? fmaf(tau, Pk, -a.lambda[(size_t)u * KP + lane]) : 0.f;

  // slot addresses: LDS byte addresses inside buffer 0 (and the same + 2 S floats for buffer 2)
  uint32_t addr[EM], addr2[EM];
  uint32_t jj[EM];
#pragma unroll
  for (int h = 0; h < EH; ++h) {
    const uint32_t sent = (uint32_t)(f.mz + l5);
    const uint32_t w = (2 * h < E) ? f.off16[((size_t)(base >> 1) + h) * 64 + lane] : (sent | (sent << 16));
    jj[2 * h] = w & 0xFFFFu; jj[2 * h + 1] = w >> 16;
  }
#pragma unroll
  for (int s = 0; s < EM; ++s) { addr[s] = pan_b + 4u * jj[s]; addr2[s] = addr[s] + (uint32_t)(2 * S * 4); }
  for (int t = tid; t < KP * KP; t += NU * 64) Cs[t] = a.C32[t];        // (the staging wave has left: NU * 64 threads)
#ifdef BNMTF_PHASE_TIMING
  const unsigned long long t_tab = tick(x + pl + __builtin_bit_cast(float, addr[EM - 1]));
#endif

  // candidate table: entry ((wave * KP + col) * NC + c) = the word-only part (log, sqrt, cos: tn_cand_pre) of candidate c of
  // column col, from words (x, y) of Philox(row, col, it, stream | c << 4) -- the streams of oracle/rng.py; every wave fills its
  // own unit's, all lanes at once (in the column loop these five transcendentals would sit on a lone wave's chain)
  if (MODE == kSweepDraw) {
    for (int e = lane; e < KP * NC; e += 64) {
      const int c = e % NC, col = e / NC;
      const U4 r = philox4x32_10(gi, a.col0 + (uint32_t)col, a.it, a.stream + 16u * (uint32_t)c, a.key0, a.key1);
      const TnCand cd = tn_cand_pre(r.x, r.y);
      tab[(size_t)wave * KP * NC + e] = f32x4t{cd.nl, cd.z, cd.sw, 0.f};
    }
  }

#ifdef BNMTF_PHASE_TIMING
  const unsigned long long t_q = tick(x);
#endif
  // q = U_i . V_j on the missing entries, from the other factor's row-major copy.  A lane's entry needs a whole row of it (256
  // bytes at K = 64).  The four lanes of a QUAD read a row together, 64 contiguous bytes per load instruction, one owner's slot
  // after the other: a load instruction then touches 16 rows x 64 bytes, and a cache line is used up by two consecutive
  // instructions.  (First version: every lane read its own rows, 16 bytes per instruction -- 64 lines per instruction, each
  // needed by eight instructions, four waves' worth thrashing the vector L1: 74 k cycles of a 165 k-cycle kernel, by the stamps.)
  // Lane r of a quad covers columns 16 i + 4 r .. + 3 of the row, i = 0 .. KP / 16 - 1; the partial dot products meet in a
  // two-step butterfly inside the quad (a fixed order: the same bits in every launch geometry).
  f32x2 q2[EH], vp2[EH];
#pragma unroll
  for (int h = 0; h < EH; ++h) { q2[h] = f32x2{0.f, 0.f}; vp2[h] = f32x2{0.f, 0.f}; }
  {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    float* xs = red + 64 + wave * 64;                      // this wave's copy of x, lane k <-> column k
    xs[lane] = x;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (wave-private: no barrier)
    const int r4 = lane & 3;
    f32x4 xq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xq[i] = (16 * i < KP) ? *reinterpret_cast<const f32x4*>(xs + 16 * i + 4 * r4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int nld = KP / 16;                               // 16-byte loads per lane and row (4 at KP = 64, 2 at KP = 32)
    constexpr int OB = NU <= 4 ? 4 : 2;                    // owners' rows in flight: 16 or 8 loads (registers: 64 or 32)
#pragma unroll
    for (int s = 0; s < EM; ++s) {
      float qs = 0.f;
#pragma unroll
      for (int o0 = 0; o0 < 4; o0 += OB) {
        f32x4 v[OB][4];
        bool real[OB];
#pragma unroll
        for (int oo = 0; oo < OB; ++oo) {
          const int o = o0 + oo;
          const uint32_t jr = o == 0 ? quad_bcast<0>(jj[s]) : o == 1 ? quad_bcast<1>(jj[s]) : o == 2 ? quad_bcast<2>(jj[s]) : quad_bcast<3>(jj[s]);
          real[oo] = s < E && jr < (uint32_t)f.Xo_rows;
          const float* row = f.Xo + (size_t)(real[oo] ? jr : 0u) * KP + 4 * r4;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[oo][i] = i < nld ? *reinterpret_cast<const f32x4*>(row + 16 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int oo = 0; oo < OB; ++oo) {
          float acc = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc = fmaf(xq[i].x, v[oo][i].x, acc); acc = fmaf(xq[i].y, v[oo][i].y, acc);
            acc = fmaf(xq[i].z, v[oo][i].z, acc); acc = fmaf(xq[i].w, v[oo][i].w, acc);
          }
          acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
          acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
          if (r4 == o0 + oo) qs = real[oo] ? acc : 0.f;
        }
      }
      if (s & 1) q2[s >> 1].y = qs; else q2[s >> 1].x = qs;
    }
  }
#ifdef BNMTF_PHASE_TIMING
  const unsigned long long t_qd = tick(q2[0].x);
#endif
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // B_start: Gram, table, first panel
#ifdef BNMTF_PHASE_TIMING
  const unsigned long long t_loop0 = tick(q2[0].x);
  unsigned long long t_bar = 0;
#endif

  float dprev = 0.f;
  const uint32_t mytab_b = (uint32_t)(uintptr_t)(lds_fp)(pan + 3 * S) + 16u * (uint32_t)(wave * KP * NC + (lane & (NC - 1)));

  // One column; BUF (which panel buffer holds column k) is compile-time: the buffer's offset is a ds_read immediate.
  auto column = [&](auto buf_c, int k) {
    constexpr int BUF = decltype(buf_c)::value;
    const float xk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), k));       // wave-uniform
    const float plk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pl), k));
    const float crow = Cs[k * KP + (lane & (KP - 1))];             // row k of the Gram, on its way while the slots are worked on (lanes >= KP: x = 0)
    const float ckk = Cs[k * KP + k];
    TnCand cand = {0.f, 0.f, 0.f};                                  // lane c: candidate c of this column (lanes >= NC repeat them)
    if (MODE == kSweepDraw) {
      const f32x4t cd = *(__attribute__((address_space(3))) const f32x4t*)(uintptr_t)(mytab_b + (uint32_t)(k * NC * 16));
      cand.nl = cd.x; cand.z = cd.y; cand.sw = cd.z;
    }
    // (A) column k-1's update from the registers that still hold v_{k-1}, (B) gather v_k into them
    const f32x2 dp2 = {dprev, dprev};
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      q2[h] = pk_fma(dp2, vp2[h], q2[h]);
      if (BUF == 2) {
        asm volatile("" : "+v"(addr2[2 * h]), "+v"(addr2[2 * h + 1]));
        vp2[h].x = *(lds_cf*)(uintptr_t)(addr2[2 * h]);
        vp2[h].y = *(lds_cf*)(uintptr_t)(addr2[2 * h + 1]);
      } else {
        asm volatile("" : "+v"(addr[2 * h]), "+v"(addr[2 * h + 1]));
        vp2[h].x = *(lds_cf*)(uintptr_t)(addr[2 * h] + (uint32_t)(BUF * S * 4));
        vp2[h].y = *(lds_cf*)(uintptr_t)(addr[2 * h + 1] + (uint32_t)(BUF * S * 4));
      }
    }
    // (C) sum q v and sum v^2, two accumulators each (a lone wave: the dependent FMAs would wait for one another)
    f32x2 qa = {0.f, 0.f}, qb = {0.f, 0.f}, va = {0.f, 0.f}, vb = {0.f, 0.f};
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      if (h & 1) { qb = pk_fma(q2[h], vp2[h], qb); vb = pk_fma(vp2[h], vp2[h], vb); }
      else       { qa = pk_fma(q2[h], vp2[h], qa); va = pk_fma(vp2[h], vp2[h], va); }
    }
    float asq_t = (va.x + va.y) + (vb.x + vb.y);
    float corr_t = fmaf(-xk, asq_t, (qa.x + qa.y) + (qb.x + qb.y));      // sum (q - x_k v) v = sum q v - x_k sum v^2
    corr_t = fmaf(-x, crow, corr_t);                                     // - sum_l x_l C_lk (all l: the l = k term is put back below)
    // wave sums (DPP inside the rows, then row 15 -> rows 1, 3 and lane 31 -> rows 2, 3): the totals sit in lane 63
    corr_t = dpp_xor_row_sum(corr_t); asq_t = dpp_xor_row_sum(asq_t);
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(corr_t), "+v"(asq_t));
    const float corr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, corr_t), 63));
    const float asq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, asq_t), 63));
    const float tau_p = tau * (ckk - asq);
    const float numer = fmaf(tau, fmaf(xk, ckk, corr), plk);
    float xnew = 0.f;
    if (MODE == kSweepDraw) {
      // lanes 0 .. NC-1 evaluate the table's candidates; the first accepted one is the draw.  A column whose NC candidates are all
      // rejected (about one in forty) evaluates the next 64 in one go, lane c candidate NC + c, and so on: the oracle's sequence.
      const TnFast tf = tn_fast_params(numer, tau_p);
      float xc;
      bool acc = tn_cand_post(tf, cand, &xc);
      xc = tn_guard(xc);
      const unsigned long long mlive = valid ? __ballot(tf.live) : 0ull;          // (uniform: all lanes or none)
      unsigned long long m = __ballot(acc) & ((1ull << NC) - 1ull);
      const bool live = mlive != 0ull;
      if (__builtin_expect(live && m == 0ull, 0)) {
        uint32_t cbase = NC;
        do {
          const U4 r = philox4x32_10(gi, a.col0 + (uint32_t)k, a.it, a.stream + 16u * (cbase + (uint32_t)lane), a.key0, a.key1);
          acc = tn_eval_fast(tf, r.x, r.y, &xc);
          xc = tn_guard(xc);
          m = __ballot(acc);
          cbase += 64u;
        } while (m == 0ull && cbase < 4096u);
      }
      const int first = m ? __builtin_ctzll(m) : 0;
      const float xd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xc), first));
      xnew = (live && m) ? xd : 0.f;
    } else {
      const float mu = numer / tau_p;
      xnew = fmaxf((valid && tau_p > 0.f && mu > 0.f) ? mu : 0.f, a.min_x);
    }
    dprev = xnew - xk;
    if (lane == k) x = xnew;
#ifdef BNMTF_PHASE_TIMING
    const unsigned long long tb0 = tick(dprev);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // B_k (the staging wave waits for the panels' landing)
#ifdef BNMTF_PHASE_TIMING
    t_bar += tick(dprev) - tb0;
#endif
  };
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  using c2 = std::integral_constant<int, 2>;
  for (int k = 0; k < K; k += 3) {
    column(c0{}, k);
    if (k + 1 < K) column(c1{}, k + 1);
    if (k + 2 < K) column(c2{}, k + 2);
  }

#ifdef BNMTF_PHASE_TIMING
  const unsigned long long t_loop1 = tick(dprev);
  if ((blockIdx.x % 64 == 3) && lane == 0)
    printf("unit block %d wave %d EM %d E %d: slabs+x+addr+C %llu table %llu qinit %llu wait %llu | loop %llu (%d columns; at barriers %llu) cycles\n", (int)blockIdx.x, wave, EM, E,
           t_tab - t_start, t_q - t_tab, t_qd - t_q, t_loop0 - t_qd, t_loop1 - t_loop0, K, t_bar);
#endif
  // ------------------------------------------------------------------------ results
  if (valid && lane < K) a.Xself[(size_t)gi * KP + lane] = x;
  if (f.stats) {                      // per-block partial sums (sum P.x', sum_miss q, sum_miss q^2) -> slab, summed by finish_kernel
    double px = (double)Pk * (double)x, sq = 0.0, sq2 = 0.0;
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      const double qa = (double)fmaf(dprev, vp2[h].x, q2[h].x), qb = (double)fmaf(dprev, vp2[h].y, q2[h].y);
      sq += qa + qb; sq2 += qa * qa + qb * qb;
    }
    if (!valid) { px = 0.0; sq = 0.0; sq2 = 0.0; }
#pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1) { px += __shfl_xor(px, mm, 64); sq += __shfl_xor(sq, mm, 64); sq2 += __shfl_xor(sq2, mm, 64); }
    double* redd = reinterpret_cast<double*>(red);
    if (lane == 0) { redd[wave * 3 + 0] = px; redd[wave * 3 + 1] = sq; redd[wave * 3 + 2] = sq2; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < 3) {
      double s = 0.0;
      for (int w = 0; w < NU; ++w) s += redd[w * 3 + tid];
      f.stats[(size_t)blockIdx.x * 4 + tid] = s;
    }
  }
}

template <int MODE, int NU>
__global__ __launch_bounds__((NU + 1) * 64, 1) void sweep_unit_kernel(SweepArgs a, FastArgs f) {
  extern __shared__ float lds[];
  const int wv = (int)(threadIdx.x >> 6);
  const int pr = blockIdx.x * NU + wv;
  int e0 = __builtin_amdgcn_readfirstlane((wv < NU && pr < f.npairs) ? (int)f.pair_E[pr] : 0);
  if (e0 > kUnitMaxSlots) e0 = 0;
  if (e0 <= 8) sweep_unit_body<8, MODE, NU>(a, f, lds);
  else if (e0 <= 12) sweep_unit_body<12, MODE, NU>(a, f, lds);
  else if (e0 <= 14) sweep_unit_body<14, MODE, NU>(a, f, lds);
  else if (e0 <= 16) sweep_unit_body<16, MODE, NU>(a, f, lds);
  else if (e0 <= 24) sweep_unit_body<24, MODE, NU>(a, f, lds);
  else sweep_unit_body<32, MODE, NU>(a, f, lds);
}

bool sweep_unit_supported(int KP, int pw) { return pw <= kUnitPanelStride && sizeof(float) * sweep_unit_lds_floats(KP, 8) <= 160 * 1024; }

template <int MODE, int NU>
static void launch_unit_inst(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  static std::atomic<uint64_t> lds_ok{0};
  const int nblocks = (f.npairs + NU - 1) / NU;
  const size_t lds = sizeof(float) * sweep_unit_lds_floats(a.KP, NU);
  if (nblocks > 0 && allow_full_lds((const void*)sweep_unit_kernel<MODE, NU>, lds_ok))
    hipLaunchKernelGGL((sweep_unit_kernel<MODE, NU>), dim3(nblocks), dim3((NU + 1) * 64), lds, st, a, f);
}

// f describes the unit-per-wave layout (f.nw = unit waves per block: 4 or 8)
void launch_sweep_unit(const SweepArgs& a, const FastArgs& f, hipStream_t st) {
  if (a.mode == kSweepDraw) { if (f.nw == 8) launch_unit_inst<kSweepDraw, 8>(a, f, st); else launch_unit_inst<kSweepDraw, 4>(a, f, st); }
  else                      { if (f.nw == 8) launch_unit_inst<kSweepMode, 8>(a, f, st); else launch_unit_inst<kSweepMode, 4>(a, f, st); }
}

}  // namespace bnmtf
