// Host-side model object behind the opaque bnmtf_handle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/bnmtf_hip.h"
#include "kernels.h"

namespace bnmtf {

void set_error(const char* fmt, ...);

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      ::bnmtf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return BNMTF_EHIP;                                                                      \
    }                                                                                         \
  } while (0)

#define CHK(expr)                  \
  do {                             \
    int rc_ = (expr);              \
    if (rc_ != BNMTF_OK) return rc_; \
  } while (0)

// One sweep direction: "rows" updates U / F (units = rows of R, inner = columns),
// "cols" updates V / G.  Everything a half sweep touches lives here.
struct Dir {
  // geometry
  int nglob = 0;        // units in the whole problem (I or J)
  int n = 0, n0 = 0;    // this rank's units [n0, n0+n)
  int m = 0;            // inner extent = nglob of the other direction
  int W = 0, KP = 0;    // factor width (K or L) and its padding (32 / 64)
  int n_pad = 0, split = 1, ipw = 0, inner_pad = 0;
  // device data
  float* big = nullptr;        // [inner_pad][n_pad]  masked R shard, output index contiguous
  float* slabs = nullptr;      // [split][n_pad][KP]
  float* lambda = nullptr;     // [n][KP]
  uint32_t* slot_ptr = nullptr;
  uint32_t* idx = nullptr;
  float* q = nullptr;
  size_t nslots = 0, nmiss = 0;
  // this direction's factor, replicated on every rank
  float* X = nullptr;  int xrows = 0;   // [xrows][KP] row major (GEMM A operand of the other direction)
  float* XT = nullptr; int ldT = 0;     // [KP][ldT]  (gather source / LDS panels of the other direction)
  float* XT2 = nullptr;                 // [KP/2][ldT][2] column pairs interleaved (pre-pass panels)
  double* Cpart = nullptr; double* spart = nullptr; double* s2part = nullptr;   // per-block Gram partials
  // fast sweep layout (bank-aware slots)
  int* f_unit_map = nullptr; uint32_t* f_pair_E = nullptr; uint32_t* f_pair_base = nullptr; uint32_t* f_off = nullptr;
  int f_npairs = 0, f_emax = 0, mz = 0; size_t f_slots = 0;
  int f_nw = 8;                          // waves per fast-sweep block
  uint32_t* f_off16 = nullptr; bool pair_ok = false;
  int pw = 0;                                     // LDS panel floats = round_up(mz + 32, 256)
  int nch = 1, mh = 0, pw_chunk = 0, pw1 = 0;     // an inner extent of two LDS panels (FastArgs::nch): chunk split, panel floats of the longer chunk / of chunk 1
  int* f_gen_units = nullptr; int f_gen_count = 0;
  // the unit-per-wave layout (kernel_sweep_unit.hip; built for directions of at most kUnitMaxUnits local units): pair p = unit p
  bool uw_ok = false; int u_nw = 4, u_emax = 0;
  int* u_unit_map = nullptr; uint32_t* u_pair_E = nullptr; uint32_t* u_pair_base = nullptr; uint32_t* u_off16 = nullptr;
  double* stats = nullptr; int stats_blocks = 0;
  bool fast_ok = false;
  int gemm_tw = 4;                       // contraction: 32-column tiles per wave
  bool use_turns = false;                // the wide layout run by kernel_sweep_turns.hip (two unit groups per block taking turns)
  bool use_wide = false;                 // 16-wave sweep kernel (pairs dealt to blocks per wave class) instead of the 8-wave one
  int vb_path = 0;                       // BNMTF_VB_PATH as it stood when the layout was built: 0 by policy, 1 masked sums, 2 pair panels (latched: the relayout builds what the chosen sweep reads)
  double* C64 = nullptr; float* C32 = nullptr; double* colsum = nullptr;   // Gram of X
  // q hand-over between the half sweeps (round 3, one GPU, 16-wave kernels on both directions): this direction's blocks end a
  // sweep by writing q = X_i . Xo_j of their missing entries, sorted by the OTHER direction's blocks, into that direction's
  // regions; they start a sweep by reading their own region instead of running the pre-pass (build_handover, api.hip)
  uint32_t* ho_in = nullptr;            // [slot rows / 2][64] two 16-bit offsets (entries) into the block's region per word, like f_off16
  uint32_t* ho_out = nullptr;           // same shape: offsets into the block's staging area (sorted by destination block)
  uint32_t* ho_pk = nullptr;            // [blocks][blocks of the other direction][3]: staging start, count, global offset in the other's regions
  uint32_t* ho_region_ofs = nullptr;    // [blocks + 1] first entry of each block's region
  float* ho_region = nullptr;           // the regions: what the other direction's sweep writes and this one reads
  int ho_blocks = 0, ho_lds_floats = 0;  // ... and the LDS floats the region (as reader) and the staging area (as writer) take
  bool ho_ready = false;                // tables exist
  mutable bool ho_filled = false;       // the regions hold q of the current (X, Xo)
  uint16_t* f_row_blk = nullptr;        // block (of ho_ppb pairs: the sweep kernel's unit waves, 16 or 8) of every slot row: build_handover's input
  int ho_ppb = 0;
  bool use_twin = false;                // the 16-wave layout run by 8-wave blocks, two to a CU (BNMTF_TWIN=1)
  bool gram_packed = false;             // colsum / colsum2 live behind C64 in one allocation (what exchange_factor all-reduces)
  hipEvent_t ev_sweep = nullptr, ev_gram = nullptr, ev_gathered = nullptr, ev_gram_all = nullptr;   // exchange_factor (several GPUs)
  float* snap_dst = nullptr;            // set for ONE relayout: where its rows also go, packed [rows][W] (run()'s sample hand-off)
  bool gram_pending = false;            // the summed Gram is still on its way on the exchange stream
  bool gemm_ahead = false;              // this direction's next contraction was launched inside the other factor's exchange (several GPUs)
  // VB only
  float *mu = nullptr, *tauq = nullptr, *var = nullptr, *S2 = nullptr, *S2T = nullptr;
  float *XS = nullptr, *vb_asq = nullptr, *vb_vsq = nullptr;   // fast VB sweep: (E, S2) pair panels; per (unit, column) sums for the ELBO pieces
  // VB on the on-chip sweep kernels (kernel_maskgemm.hip): the mask's bits of the local units [inner_pad / 32][n_pad], this
  // factor's moments as bf16 planes for the OTHER direction's masked sums, and the slabs of this direction's masked sums
  uint32_t* mbits = nullptr; uint32_t* XB = nullptr; int xb_rows = 0; float* mslabs = nullptr; int msplit = 0, mipw = 0;
  unsigned* xb_umax = nullptr; int* xb_cexp = nullptr; float* xb_mpart = nullptr; bool xb_umax_posted = false;       // per column of [S2 | E^2]: largest element's bits, exponent of the fixed-point grid
  bool wide_can = false;                 // the 16-wave kernels can run on this direction (<= kWideMaxSlots slots per lane, LDS fits)
  double* colsum2 = nullptr;
  double* vb_stats = nullptr;           // VB: [n][8] per-unit partial sums (generic sweep) or [ceil(n/4)][8] per-block (fast sweep)
  int vb_stat_rows = 0;                 // rows of vb_stats the last sweep wrote
  // cond-params scratch
  double *numer = nullptr, *taup = nullptr;
  std::vector<uint32_t> obs_count;      // host, all nglob units
};

struct Comm;   // RCCL wrapper (comm.cpp)

// A model small enough for the one-launch path (kernel_small.hip, api_small.inc): ONE device allocation holds its static tables
// (both masked operands, slot / segment / permutation tables, prior rates) and its state (factors, q of the missing entries,
// Gram matrices); `dev` is the launch descriptor with the per-call fields blank.
struct SmallModel {
  char* arena = nullptr; size_t arena_bytes = 0;
  SmallLaunch dev;
  int em = 8, nt = 1024;             // slot class (max over the two directions) and threads per block of the launch
  size_t lds_bytes = 0;
  bool q_valid = false;              // cols.q holds q of the current state (hand-over from call to call)
  // per-call buffers, grown on demand: [gunit | rec | clock] and the samples
  char* call_buf = nullptr; size_t call_cap = 0;
  float* smp = nullptr; size_t smp_cap = 0;
  float* state_host = nullptr; size_t state_host_cap = 0;    // page-locked staging of a batch's final states (owned by the call's first model)
};

}  // namespace bnmtf

struct bnmtf_model {
  int I = 0, J = 0, K = 0, L = 0;
  double alpha = 0, beta = 0;
  uint64_t seed = 0, iteration = 0;
  int device = 0, rank = 0, world = 1;
  hipStream_t stream = nullptr;
  hipStream_t aux_stream = nullptr; hipEvent_t ev_aux0 = nullptr, ev_aux1 = nullptr;   // BNMTF S step: the b side of the system beside the A side (api_models.inc)
  hipStream_t xchg_stream = nullptr;    // several GPUs: every collective is issued here, beside the compute stream
  bnmtf::Dir rows, cols;
  // full-matrix copies for predict()/validation
  float* Rfull = nullptr; uint8_t* Mtrain = nullptr; uint8_t* Mscratch = nullptr;
  double *Ad = nullptr, *Bd = nullptr, *out6 = nullptr;
  // scalars
  double n_obs = 0, sumR = 0, sumR2 = 0;
  double* tau_d = nullptr; float* tau_f = nullptr;
  double* acc = nullptr;     // [4]
  double* rec = nullptr; size_t rec_cap = 0;
  bool have_state = false, vb_ready = false;
  double *A2d = nullptr, *B2d = nullptr, *vb_rec = nullptr; size_t vb_rec_cap = 0;
  double* vbred = nullptr;               // VB over several GPUs: the 20 sums exchanged per iteration
  bool use_fast = true, last_sweep_fast = false;   // fast sweep kernel when the shape allows it
  // the one-launch path for small models (api_small.inc): its arena; whether the multi-launch path's structures exist yet (they
  // are built on first need for a model that starts small); which of the two holds the current state
  bnmtf::SmallModel* small = nullptr;
  int small_mode = 1;          // bnmtf_set_small_path: 0 never, 1 when it is the faster path for the call (api_small.inc small_wanted), 2 always
  bool std_built = true, small_cur = false, std_cur = false;
  bool pool_stream = false;    // the stream goes back to the per-process pool at destroy (small models)
  bool one_arena = false;      // Rfull, Mtrain, the scalars, the posterior sums and Ad / Bd live in the small model's arena (one allocation per model)
  std::vector<double> lam_rows, lam_cols, lam_S;   // prior rates as given (build_standard may run after bnmtf_create has returned)
  // BNMTF extras
  float* S = nullptr;            // [K][L] on device (row major, unpadded)
  bnmtf::Dir reff, ceff;         // effective factors U_eff = F S (I x L), V_eff = G S^T (J x K): factor storage only
  float *slabsS = nullptr, *CfS = nullptr, *deltaS = nullptr, *s_partial = nullptr, *s_w = nullptr, *s_omp = nullptr, *lambdaS = nullptr;
  double *s_numer = nullptr, *s_taup = nullptr;
  int s_blocks = 0;
  // dense S system (kernel_ssys.hip), K, L <= 32
  bool ssys = false; int ss_nsplit = 1;
  // the dense S system (kernel_ssys.hip): packed upper triangles of the per-column masked Grams (Wc) and of G_j G_j^T (Gc),
  // column-range slabs of their product, AB = [A (n2 x n2) | b (n2)] (one buffer, one all-reduce), the residual, the
  // per-block partials of b, and the chain's first sampler candidates of the iteration
  float *ss_Wc = nullptr, *ss_Gc = nullptr, *ss_slabs = nullptr, *ss_AB = nullptr, *ss_r = nullptr, *ss_bpart = nullptr, *ss_cands = nullptr, *ss_tinv = nullptr, *ss_rec = nullptr;
  // posterior means accumulated on the device (bnmtf_set_expectation): sums over the iterations burn_in, burn_in + thinning, ...
  int exp_burn = -1, exp_thin = 1; uint64_t exp_count = 0;
  double *exp_rows = nullptr, *exp_cols = nullptr, *exp_S = nullptr, *exp_tau = nullptr;
  // variational tri-factorisation (api_trivb.inc)
  bool tri_ready = false;
  float *muS = nullptr, *tauS = nullptr, *varS = nullptr;   // q(S) beside expS = S   [K][L]
  float *mv_rows = nullptr, *mv_cols = nullptr;             // masked variance sums mvG [I_loc][32], mvF [J_loc][32]
  int* tri_order = nullptr; size_t tri_order_cap = 0;        // per iteration: K L entries of S, K columns of F, L columns of G
  double* tri_sums = nullptr;                                // 3 x 8 masked sums (metric_kernel passes)
  uint32_t s_word0 = 0; int s_ldword = 0;                    // a block of a wider S (bnmtf_set_s_block): the Philox column word of its entry (k, l) is s_word0 + k s_ldword + l (0: k L + l)
  bool tri_w_current = false;                                // ss_Wc holds the masked column Grams of the current q(F) (formed behind its sweep)
  bool tri_mv_cols_current = false;                          // mv_cols holds the masked variance sums of the current q(F) (formed in the G step)
  bool tri_pv_current = false;                               // bnmtf_vb_run: slabsS holds R~^T E[F] of the current E[F] (formed behind the F sweep: the G step's and the next S system's)
  float* ss_Aperm = nullptr;                                 // the S system in the order of the current pass (bnmtf_vb's chain)
  double* tri_third = nullptr;                               // per-block partial sums of exp_square_diff's third term (run loop)
  int last_tri_path = 0;                                     // the F / G sweeps of the last bnmtf_vb_run: 1 generic kernel, 2 on-chip (pair panels + covariance term)
  // profiling
  double* gunit = nullptr;               // [rec_cap] Gamma(alpha_s, 1) variates staged by run()
  std::vector<double> gunit_host;
  double min_tn = 0.0;                   // ICM: lower clamp of every mode update (run(iterations, minimum_TN))
  float cur_min_x = 0.f;                 // clamp in force for the sweeps being enqueued
  uint32_t profiling = 0;                // bit k: bracket the launches of kernel k with events
  bool ho_regions_current = false;              // the regions hold q of the state as the last run call left it (no set_state since)
  bool ho_enabled = false, ho_active = false;   // q hand-over between the half sweeps (Dir::ho_*): tables built / in use by the running loop
  uint32_t col0 = 0; bool block_mode = false;    // a column block of a wider factorisation (bnmf_set_column_block): Philox column offset; no q hand-over
  double* AdW = nullptr; double* BdW = nullptr; int ABd_width = 0;     // the metric kernel's operand copies for factors wider than 64 columns (bnmtf_metric_sums_wide)
  bool uw_force = false;           // BNMTF_UNIT=1: the unit-per-wave sweep even where the hand-over is on (A/B)
  uint64_t ho_refresh = 64;                     // the rows sweep runs its pre-pass every ho_refresh-th iteration
  uint64_t profile_stride = 1;           // ... in every profile_stride-th iteration
  double kernel_ms[BNMTF_KERNEL_COUNT] = {0};
  uint64_t kernel_launches[BNMTF_KERNEL_COUNT] = {0};
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending_events;
  std::vector<hipEvent_t> event_pool;
  bnmtf::Comm* comm = nullptr;
  std::string description;
  int last_vb_path = 0;                 // 0: no variational sweep yet, 1 generic kernel, 2 pair-panel kernel, 3 on-chip kernel + masked sums
  // sample hand-off (all_U / all_V ...): device snapshots + a copy stream (api.hip, SampleSink)
  hipStream_t copy_stream = nullptr;
  float* snap_dev = nullptr; size_t snap_dev_cap = 0;       // [depth][floats per iteration]
  float* snap_host = nullptr; size_t snap_host_cap = 0;     // pinned ring, used when the caller's buffers are pageable
  hipEvent_t snap_ready[8] = {nullptr}, copy_done[8] = {nullptr};
  double create_ms = 0.0;                                   // wall time of bnmtf_create (host layout + uploads)
};
