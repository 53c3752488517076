// Kernel launch interface shared by api.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bnmtf {

// RNG stream ids (counter word 3, low 4 bits) -- same numbers as oracle/rng.py
constexpr uint32_t kStreamRows = 0, kStreamCols = 1, kStreamS = 2, kStreamTau = 3, kStreamHook = 8;

// ---------------------------------------------------------------------------
// K1/K2: masked contraction  out[n][KP] = sum_inner big[inner][n] * X[inner][KP]
//   big  : R~ shard stored so that the OUTPUT index is the contiguous one
//          (rows step:  big[j][i] = R~[i][j];   cols step: big[i][j] = R~[i][j])
//   X    : the other factor, row-major, padded rows are zero
//   slabs: [split][n_pad][KP] partial sums, summed by the consumer (sweep prologue)
// ---------------------------------------------------------------------------
struct GemmArgs {
  const float* big; int ld;          // leading dimension of big (= n_pad)
  const float* X;                    // [inner_pad][KP]
  float* slabs;                      // [split][n_pad][KP]
  int n_pad, split, inner_per_wave;  // inner range of wave w in split s: [(s*4+w)*ipw, +ipw)
  int tw;                            // 32-column tiles per wave: 4 (128 columns) or 2 (64 columns, short output sides)
  int split0 = 0, nsplit = 0;        // this launch's inner slices [split0, split0 + nsplit) (nsplit = 0: all `split` of them)
};
void launch_gemm(const GemmArgs& a, int KP, hipStream_t st);

// ---------------------------------------------------------------------------
// VB (kernel_maskgemm.hip): out[u][0:2KP] = sum_{r in miss(u)} [S2o | Eo^2][r][:] as a product with the mask's bits
//   bits : [ldw][n_pad] u32, bit b of word w = entry (unit, inner 32 w + b) is missing (zero beyond the inner extent)
//   XB   : the moments of the other factor on a per-column fixed-point grid (2^(cexp[c] - 22)) as three planes of balanced
//          base-256 digits in fragment layout [3][rows_pad / 16][2 KP][16 bytes]; umax: scratch of vb_colmax_kernel
//   slabs: [split][n_pad][2 KP] fp32 partial sums (one per inner slice; integer inside), added by the consumer in slab order
// ---------------------------------------------------------------------------
struct MaskGemmArgs {
  const uint32_t* bits; int ldw;
  const uint32_t* XB; int rows_pad; const int* cexp;
  float* slabs;
  int n_pad, split, inner_per_wave;      // inner_per_wave: inner rows of a BLOCK's slice here (its four waves share the slice and split the columns)
  int ncol = 0;                  // 2 KP (set by launch_maskgemm)
};
void launch_maskgemm(const MaskGemmArgs& a, int KP, hipStream_t st);
// umax [2 KP]: the column maxima's bits.  have_max: already there (the relayout's Gram blocks and gram_reduce_kernel: PostArgs::umax);
// otherwise a pass of its own takes them first
void launch_vb_planes(const float* S2, const float* E, int rows, int rows_pad, int KP, unsigned* umax, bool have_max, int* cexp, uint32_t* XB, hipStream_t st);
void launch_mask_bits(const uint8_t* M, int I, int J, int by_rows, int unit0, int n, int m, int n_pad, int ldw, uint32_t* bits, hipStream_t st);

// ---------------------------------------------------------------------------
// Gram: C[a][b] = sum_r X[r][a] X[r][b] (fp64 accumulate), colsum[a] = sum_r X[r][a]
//   also for VB: colsum2[a] = sum_r S2[r][a] when S2 != nullptr
// ---------------------------------------------------------------------------
struct GramArgs {
  const float* X; const float* S2; int rows, KP;
  double* C64; float* C32; double* colsum; double* colsum2;
};
void launch_gram(const GramArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------
// K3: sequential-k conditional update of one factor (one half sweep)
// ---------------------------------------------------------------------------
enum SweepMode { kSweepDraw = 0, kSweepMode = 1, kSweepVB = 2 };

struct SweepArgs {
  int n, n0;                 // local units, global index of the first
  const int* unit_list;      // generic kernel: when non-null, the n units to process (local indices)
  int K, KP;                 // true and padded width
  int mode;                  // SweepMode
  int cond_k;                // >= 0: only evaluate column cond_k, write numer/tau, change nothing
  int qinit_only;            // generic kernel: compute q = X_i . Xo_j on the missing entries and stop
  int only_k;                // >= 0 (VB hooks): update only this column (update_U(k) / update_V(k))
  uint32_t col0;             // this factor holds columns [col0, col0 + K) of a wider factorisation (ranks above 64 run as column blocks:
                             // bnmf_set_column_block): the Philox column word of local column k is col0 + k
  float min_x;               // mode updates: lower clamp of the new value (ICM minimum_TN; 0 otherwise)
  int vb_moments;            // VB: also refresh exp/var from the new mu/tau (update_exp_U(k))
  double* vb_stats;          // VB: [n][8] per-unit ELBO / exp_square_diff partial sums (may be null)
  // numerators
  const float* slabs; int split, n_pad;
  const float* lambda;       // [n][KP] local
  // own factor (global, row-major [*][KP]) and its transposed copy [KP][ldT]
  float* Xself; float* XselfT; int ldT_self;
  // other factor, transposed [KP][ldT_o]; index m_sentinel reads 0
  const float* XoT; int ldT_o;
  const float* C32;          // Gram of other factor [KP][KP]
  // missing-entry slots (64 per unit row-step)
  const uint32_t* slot_ptr; const uint32_t* idx; float* q;
  // precision
  const float* tau;          // device scalar
  // RNG
  uint32_t key0, key1, it, stream;
  // outputs
  double* acc;               // [0] += sum P.X', [1] += sum_miss q, [2] += sum_miss q^2 (may be null)
  double* numer_out; double* tau_out;   // cond mode, length n
  // VB extras (mode == kSweepVB)
  float* mu_self; float* tau_self; float* var_self; float* S2self; float* S2selfT;
  const float* S2oT; const double* colsum2_o;   // other's var+exp^2 (transposed) and its column sums
  // variational tri-factorisation (kernel_trivb.hip): covariance term of update_F / update_G and the column order
  const float* cov_S;        // E[S]; element (column c of this factor, inner t) at cov_S[c * cov_sc + t * cov_st]; null: no term
  int cov_sc, cov_st, cov_n; // strides and inner extent (L for the F step, K for the G step)
  const float* cov_mv;       // [n][32] masked variance sums of the other factor (mvG / mvF), local units
  const int* order;          // K column indices in update order, or null (0 .. K-1)
};
void launch_sweep(const SweepArgs& a, hipStream_t st);

// on-chip path (sweep_chip.inc): bank-aware slot layout, q in registers, panels in LDS
struct FastArgs {
  const int* unit_map;         // [2*npairs] local unit of each half wave, or -1
  const uint32_t* pair_E;      // [npairs] slots of the pair (max of its two units)
  const uint32_t* pair_base;   // [npairs] first slot row of the pair
  const uint32_t* off;         // [(base+s)*64 + lane] inner index j (j mod 32 == lane mod 32) or mz + lane%32
  const uint32_t* off16;       // [(base/2+h)*64 + lane] slots 2h (low 16 bits) and 2h+1 (high) packed, or null
  int nw;                      // unit waves per block (2, 4, 8 or 16)
  int npairs;                  // pairs in descending slot-count order
  int mz, pw;                  // inner extent rounded up to 32 (sentinel zero words at mz..mz+31); panel floats pw = round_up(mz+32, 256)
  // an inner extent too long for one LDS panel (round 3): the inner indices are cut in TWO chunks, [0, mh) and [mh, m); a
  // unit's slots are its chunk-0 entries (first half of its slot rows) then its chunk-1 entries, inner indices LOCAL to the
  // chunk; a column is staged and gathered chunk by chunk (nch = 2, sweep_chip.inc); pw = the longer chunk's panel floats.
  int nch, mh, pw1;            // chunks (1 or 2), first inner index of chunk 1 (multiple of 256), panel floats of chunk 1
  // q hand-over between the half sweeps (16-wave kernel, one GPU; model.h Dir::ho_*): read q from the block's region instead of
  // the pre-pass / write q sorted by the other direction's blocks at the end
  int twin;                    // the twin shape: 8-wave blocks, two to a CU, on the 16-wave layout (sweep_chip.inc)
  int ho_lds_floats;           // floats behind the Gram in LDS that the hand-over's region / staging area needs (0: none)
  int ho_read, ho_write, ho_nb_other, ho_rows_total;       // ho_rows_total: slot rows of the whole direction (= the end of the last block's slice)
  const uint32_t* ho_in; const float* ho_region; const uint32_t* ho_region_ofs;
  const uint32_t* ho_out; const uint32_t* ho_pk; float* ho_dst;
  const float* XoT; int ldT_o;   // other factor transposed [KP][ldT_o]
  const float* XoT2; int ld2_o;  // other factor, column pairs interleaved [KP/2][ld2_o][2]
  const float* Xo; int Xo_rows;  // other factor row major [Xo_rows][KP] (kernel_sweep_unit.hip: q rebuilt from its rows)
  double* stats;               // [blocks][4] partial (sum P.X', sum_miss q, sum_miss q^2) or null
  // VB sweep (kernel_sweep_vb.hip)
  const float* XoS;            // other factor's (E, S2) pair panels [KP][ld2_o][2]
  float* vb_asq; float* vb_vsq;  // [rows][KP] per (unit, column): sum_miss S2other, sum_miss Eother^2 (for vb_pieces_kernel)
  // VB on the on-chip kernels (sweep_chip.inc, MODE = kSweepVB): the two masked sums of every (unit, column), from
  // kernel_maskgemm.hip: [msplit][n_pad][2 KP] slabs, columns [0, KP) = sum_miss S2other, [KP, 2 KP) = sum_miss Eother^2
  const float* mslabs; int msplit;
};
// 2/4/8-wave instantiations (kernel_sweep_fast.hip): pairs that need more than kFastMaxSlots slots per lane go to the generic kernel
constexpr int kFastMaxSlots = 56;
bool sweep_fast_supported(int KP, int pw);
// an inner extent m too long for one LDS panel: can it be cut in two chunks [0, mh) and [mh, m) that fit one each?  Then mh (a
// multiple of 256), the panel floats of the longer chunk (pw: LDS sizing) and of chunk 1 (pw1).
bool sweep_two_chunks_plan(int KP, int m, int* mh, int* pw, int* pw1);
void launch_sweep_fast(const SweepArgs& a, const FastArgs& f, hipStream_t st);
void launch_sweep_small(const SweepArgs& a, const FastArgs& f, hipStream_t st);   // nw = 2, 4 (kernel_sweep_small.hip)
// one unit per 64-lane wave, f.nw = 4 or 8 unit waves + a staging wave per block (kernel_sweep_unit.hip): few units per CU --
// shards of a multi-GPU run, small problems.  The layout: Dir::u_* (both halves of a pair hold entries of the same unit)
constexpr int kUnitMaxUnits = 2048;          // local units of a direction up to which the shape is used: eight per CU
constexpr int kUnitMaxSlots = 32;            // ... and slots per lane (2 048 missing entries per unit); a direction with a fuller unit keeps the pair layout's kernels
bool sweep_unit_supported(int KP, int pw);
void launch_sweep_unit(const SweepArgs& a, const FastArgs& f, hipStream_t st);
// 16-wave instantiation (kernel_sweep_wide.hip): at most kWideMaxSlots slots per lane, 16 pairs per block
constexpr int kWideMaxSlots = 32;
constexpr int kTwinPanelStride = 8448;       // the twin shape (sweep_chip.inc, TW = 1): floats between its two panel buffers = the largest panel it takes
bool sweep_wide_supported(int KP, int pw);
void launch_sweep_wide(const SweepArgs& a, const FastArgs& f, hipStream_t st);
// the same pair layout run by 8-wave blocks of four units per wave, two groups of units taking turns so that a group's draw
// is made during the other group's slot work (kernel_sweep_turns.hip; an experiment kept behind BNMTF_TURNS=1: correct,
// slower than the 16-wave kernel -- DESIGN.md section 7.3)
bool sweep_turns_supported(int KP, int pw);
void launch_sweep_turns(const SweepArgs& a, const FastArgs& f, hipStream_t st);
// VB sweep in the 16-wave shape (kernel_sweep_vb.hip) and the ELBO pieces it leaves to a second pass
bool sweep_vb_supported(int KP, int pw);
bool sweep_vb_cov_supported(int KP, int pw);   // ... with SweepArgs::cov_S / order: the tri-factorisation's F / G sweeps (K, L <= 32)
int sweep_vb_blocks(int npairs, int nw = 8);  // blocks (= rows of FastArgs::stats) of a VB sweep over npairs pairs with nw unit waves per block (16 or 8)
void launch_sweep_vb(const SweepArgs& a, const FastArgs& f, hipStream_t st);
void launch_vb_pieces(int n, int n0, int KP, int K, const float* mu, const float* tauq, const float* ex, const float* var,
                      const float* lambda, const float* asq, const float* vsq, double* out, hipStream_t st);
// the same pieces with the two masked sums taken from the slabs of kernel_maskgemm.hip (the on-chip VB sweep)
void launch_vb_pieces_slabs(int n, int n0, int KP, int K, const float* mu, const float* tauq, const float* ex, const float* var,
                            const float* lambda, const float* mslabs, int msplit, int n_pad, double* out, hipStream_t st);

// relayout + Gram after a sweep / state upload: X -> XT, XT2, partial Gram slabs; then the reduction
struct PostArgs {
  const float* X; int rows, KP;
  float* XT; int ldT; float* XT2; int ld2;
  double* Cpart; double* spart;          // [blocks][KP*KP], [blocks][KP]
  double* C64; float* C32; double* colsum;
  // VB: second moment matrix S2 = var + exp^2
  const float* S2; float* S2T; double* s2part; double* colsum2;
  float* XS;                             // VB: [KP][ldT][2] (E, S2) interleaved per row: pair panels of the fast VB sweep, or null
  float* mpart; unsigned* umax;          // VB, whole-factor launches (or null): per-block and final column maxima of [S2 | E^2] (kernel_maskgemm.hip)
  // which half of the work a launch does (launch_post: both, every row).  Several GPUs: the Gram partial is formed over the
  // rank's OWN rows [own0, own1) only (blocks blk0 .. of 32 rows; rows outside the range count as zero) and the partial
  // C64 | colsum is summed over the ranks by one all-reduce; the layouts are written for all rows once they are gathered.
  int do_layout, do_gram, blk0, own0, own1;
  // the sample hand-off of run(): the layout blocks also write the rows packed [rows][snapW] into a snapshot slot (what a
  // separate compaction kernel did), or null
  float* snap; int snapW;
};
void launch_post(const PostArgs& a, hipStream_t st);
void launch_post_layout(const PostArgs& a, hipStream_t st);                       // XT / XT2 (/ S2T / XS) of every row
void launch_post_gram_rows(const PostArgs& a, int own0, int own1, hipStream_t st);   // C64, colsum (, colsum2) of the rows [own0, own1) only
// bnmtf_create's passes over the I x J data, on the device (kernel_layout.hip): the masked (and, for the rows direction, transposed)
// contraction operand big[r][ul], and every unit's missing inner indices in order (64-wide slots, padded with m)
void launch_masked_operand(const float* R, const uint8_t* M, int I, int J, int by_rows, int unit0, int n, int m, float* out, int ld, hipStream_t st);
// the same operand for the RESIDUAL data of a column block: M ? R - sum_b A_b[i] . B_b[j] : 0 (up to kMaxOtherBlocks products of
// width W_b, row-major [.][KP_b] factors): what the other column blocks of a wider factorisation explain is taken off the data
constexpr int kMaxOtherBlocks = 4;      // (a model of four column blocks has three others; the S blocks of a wide tri-factorisation take four carriers)
struct ResidualSpec { int n; const float* A[kMaxOtherBlocks]; const float* B[kMaxOtherBlocks]; int KP[kMaxOtherBlocks]; int W[kMaxOtherBlocks]; };
void launch_residual_operand(const float* R, const uint8_t* M, int I, int J, int by_rows, int unit0, int n, int m, float* out, int ld,
                             const ResidualSpec& rs, hipStream_t st);
void launch_missing_lists(const uint8_t* M, int I, int J, int by_rows, int unit0, int n, int m, const uint32_t* ptr, uint32_t* idx, hipStream_t st);
// q hand-over tables of one writer / reader pair of directions, built on the device (kernel_handover.hip)
struct HandoverArgs {
  // the writer's and the reader's slot layouts (FastArgs::off, pair_base, pair_E, unit_map), pairs, pairs per block (the sweep
  // kernel's unit waves: 16 or 8), blocks, inner extents
  const uint32_t* w_off; const uint32_t* w_pB; const uint32_t* w_pE; const int* w_umap; int w_npairs, w_ppb, w_nb; uint32_t w_inner;
  const uint32_t* r_off; const uint32_t* r_pB; const uint32_t* r_pE; const int* r_umap; int r_npairs, r_ppb, r_nb; uint32_t r_inner;
  const uint16_t* r_row_blk;                 // reader: block of every slot row
  const uint32_t* r_ptr; const uint32_t* r_idx;   // reader: sorted missing inner indices per unit (Dir::slot_ptr, Dir::idx)
  // scratch
  uint32_t* inv; uint32_t* dst_slot; uint32_t* dst_rank; uint32_t* count;
  uint32_t* sbase; uint32_t* stotal; uint32_t* rbase; uint32_t* rdata; uint32_t* rsize;
  // results
  uint32_t* rofs; uint16_t* t_out; uint16_t* t_in; uint32_t* pk;
  uint32_t* limits;                          // [0] longest staging area, [1] largest region, [2] entries without a partner, [3] all regions (entries)
};
void launch_handover_build(const HandoverArgs& a, hipStream_t st);
void launch_gram_cast(const double* C64, float* C32, int n, hipStream_t st);      // C32 = (float) C64 after the partial sums were all-reduced
constexpr int kPostRows = 32;
inline int post_blocks(int rows) { return (rows + kPostRows - 1) / kPostRows; }

// ---------------------------------------------------------------------------
// end of iteration: masked SSE from Gram identities, tau draw, metrics record
// ---------------------------------------------------------------------------
struct FinishArgs {
  const double* Cr64; const double* Cc64; const double* sr; const double* sc; int KP;
  const double* acc;          // [3] atomically accumulated sums (generic sweep)
  const double* stats; int nstats;   // [nstats][4] per-block partial sums (fast sweep), may be null
  double n_obs, sumR, sumR2;  // over the training mask
  double alpha, beta;
  int update;                 // 0 draw, 1 mode (tau = alpha_s/beta_s), 2 ICM (tau = (alpha_s - 1)/beta_s, gamma_mode)
  uint32_t key0, key1, it;
  const double* gunit;        // Gamma(alpha_s, 1) variate of this iteration computed ahead on the host, or null (draw here)
  double* tau_d; float* tau_f;
  double* rec;                // [5]: tau, MSE, R2, Rp, SSE  (slot of this iteration)
  const float* copy_src; float* copy_dst; int copy_n;     // a small array that rides along (the tri-factorisation's sample of S into its slot), or null
};
void launch_finish(const FinishArgs& a, hipStream_t st);
struct VbFinishArgs {
  const double* Cr64; const double* Cc64; const double* sr; const double* sc; const double* s2r; const double* s2c; int KP;
  const double* acc;                   // [3] sum P.X', sum_miss q, sum_miss q^2 (cols sweep)
  const double* stats_r; int nr;       // [nr][8] rows-sweep per-unit sums
  const double* stats_c; int nc;       // [nc][8] cols-sweep per-unit sums
  double n_obs, sumR, sumR2, alpha, beta;
  double* tau_d; float* tau_f;         // exptau
  double* rec;                         // [16]: exptau, MSE, R2, Rp, ESD, beta_s, then 4 ELBO sums for U and 4 for V
  const double* extra; int n_extra;    // (tri-factorisation: partial sums of exp_square_diff's third term, added to ESD; null: none)
  const double* sweep_stats; int n_sweep_stats;    // [n][4] per-block sums of the on-chip cols sweep, added to acc here (null: acc holds everything)
};
void launch_vb_finish(const VbFinishArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------
// direct masked metric sums in fp64 (predict / validation path)
// ---------------------------------------------------------------------------
struct MetricArgs {
  const float* R; const uint8_t* Mp; int I, J;
  const double* A; const double* B; int K;     // A [I][K], B [J][K]
  const double* A2; const double* B2;          // optional (VB): second-moment factors; out[6] = sum_mask (A2_i.B2_j - sum_k A_ik^2 B_jk^2)
  double* out6;                                // 8 doubles
};
void launch_metric_sums(const MetricArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------
// BNMTF (kernel_bnmtf.hip)
// ---------------------------------------------------------------------------
struct SmallProductArgs {   // out = X . S (transposeS = 0, out width L) or X . S^T (transposeS = 1, out width K)
  const float* X; int rows, KPin; const float* S; int K, L; int transposeS; float* out; int KPout;
};
void launch_small_product(const SmallProductArgs& a, hipStream_t st);
struct SlabProductArgs {    // out[n][KPout] = (sum of the `split` slabs [n_pad][KPin]) . S: contraction slabs times S (kernel_bnmtf.hip)
  const float* slabs; int split, n_pad, KPin; const float* S; int K, L; int n; float* out; int KPout;
  int transposeS = 0;          // 1: out[n][K] = slabs[n][L] . S^T (the F side: R~ (G S^T) = (R~ G) S^T); out may be slab 0 of `slabs` when KPout == 32
};
void launch_slab_product(const SlabProductArgs& a, hipStream_t st);
void launch_cfs(const double* Cf64, int KPk, const float* S, int K, int L, float* CfS, hipStream_t st);
struct SRowArgs {           // one row k of S: J-vectors h_k, w_k and per-block partial eta / Omega
  int n, n0, k, K, L, KPk, KPl;
  const float* G;                      // [J][KPl]
  const float* FT; int ldT;            // F transposed [KPk][ldT]
  const float* slabs; int split, n_pad;   // Pv = R~^T F partial slabs, width KPk
  const uint32_t* slot_ptr; const uint32_t* idx; float* q;   // generic (64-wide) slots of the cols direction
  const float* CfS;                    // [K][L] current Cf.S
  const float* delta_prev; int apply_prev;   // delta of row k-1: q += F_{i,k-1} (G_j . delta) before use
  float* partial;                      // [blocks][L]: per-block share of eta
};
constexpr int kSOmegaChunks = 8;       // column chunks of the Omega^k partial sums
struct SOmegaArgs {         // w[j][k] and the chunk partials of Omega^k for every k (once per iteration)
  int n, n0, K, L, KPk, KPl, nch, zero_row;
  const float* F;                      // [I+][KPk] row major (row zero_row is zero)
  const float* G;                      // [J][KPl]
  const float* Cf32;                   // F^T F [KPk][KPk]
  const uint32_t* slot_ptr; const uint32_t* idx;
  float* w;                            // [J][KPk]
  float* omp;                          // [K][nch][LP*LP]
};
void launch_srow_omega(const SOmegaArgs& a, hipStream_t st);
void launch_srow_qinit(const SOmegaArgs& a, const float* Ueff, float* q, hipStream_t st);   // q = (F S)_i . G_j on the column slots (KPl == 32)
void launch_srow_gather(const SRowArgs& a, int blocks, hipStream_t st);
struct SDrawArgs {
  int k, K, L, KPk, nblocks, update, cond_l, LP, nch;
  float min_x;                         // mode updates: lower clamp (ICM minimum_TN)
  const float* partial; const float* omp; float* S; const float* lambdaS; const float* tau;
  const double* Cf64; float* CfS; float* delta_out;
  uint32_t key0, key1, it;
  uint32_t word0; int ldword;          // the Philox column word of entry (k, l) is word0 + k ldword + l: (0, L) for a whole S, (row0 Lw + col0, Lw) for a block of a wider one
  double* numer_out; double* tau_out;
};
void launch_srow_draw(const SDrawArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------
// BNMTF S step as a dense K.L x K.L system (kernel_ssys.hip); K, L <= 32
// ---------------------------------------------------------------------------
// packed upper triangle of a symmetric index pair (k <= k' < K): p = k K - k (k - 1) / 2 + (k' - k); rows of the packed
// arrays are padded to a multiple of 64 (the GEMM's 2 x 2 MFMA tiles) with zeros
__host__ __device__ inline int tri_count(int K) { return K * (K + 1) / 2; }
__host__ __device__ inline int tri_padded(int K) { return (tri_count(K) + 63) / 64 * 64; }
__host__ __device__ inline int tri_index(int k, int kp, int K) { return k * K - k * (k - 1) / 2 + (kp - k); }
// where packed entry p sits in its row: the two 32-entry tiles of a 64-group are interleaved word by word, so that one
// 8-byte load gives a lane its element of both tiles
__host__ __device__ inline int tri_pos(int p) { return (p & ~63) + 2 * (p & 31) + ((p >> 5) & 1); }
__host__ __device__ inline int tri_unpos(int pos) { return (pos & ~63) + 32 * (pos & 1) + ((pos & 63) >> 1); }   // inverse of tri_pos
__host__ __device__ inline void tri_unindex(int p, int K, int* k, int* kp) {
  int kk = 0;
  while (p >= K - kk) { p -= K - kk; ++kk; }
  *k = kk; *kp = kk + p;
}
struct SColGramArgs {       // W~_j = C~f - sum_{i in miss(j)} (F_i F_i^T + diag varF_i) for the local columns
  int n, K;
  const float* F;                      // [I+][32] row major, padding slots of idx point at a zero row
  const float* varF;                   // [I+][32] or null (Gibbs)
  const double* Cf64;                  // F^T F [32][32]
  const double* cf_diag_extra;         // VB: sum_i (E[F_ik]^2 + varF_ik) [32] (replaces the diagonal of Cf64), or null
  const uint32_t* slot_ptr; const uint32_t* idx;   // 64-wide slots of the cols direction
  float* Wc;                           // [n + 2][tri_padded(K)]: the packed upper triangle of W~_j (pads and the two extra rows stay zero)
  float* var_obs_out = nullptr;        // VB with varF: [n][32] the observed rows' variance sums (total minus the gathered missing ones) go here -- the G step's mv
  const float* var_obs = nullptr;      // VB, instead of varF: [n][32] sum_{i in Omega_j} varF_ik (the G step's masked variance sums) -- the diagonal's sum over the MISSING rows is the total minus this
};
struct GammaPackArgs;
struct SSysBArgs;
// pack_too.n > 0: the packing of G's second moments in the same launch; b_too != nullptr: and the blocks of b = sum_j Pv_j (x) G_j
void launch_scol_gram(const SColGramArgs& a, const GammaPackArgs& pack_too, hipStream_t st, const SSysBArgs* b_too = nullptr);
struct GammaPackArgs {      // Gc[j][r(l, l')] = G_jl G_jl' (+ varG_jl when l = l': the second moment, VB) for the local columns
  int n, n0, L;
  const float* G; const float* varG;   // [J][32] (global rows n0 + j); varG null for Gibbs
  float* Gc;                           // [n + 2][tri_padded(L)]
};
void launch_gamma_pack(const GammaPackArgs& a, hipStream_t st);
struct SSysGemmArgs {       // slabs[s][p][r] = sum_{j in range s} Wc[j][p] Gc[j][r]: A on the packed pairs p = (k <= k'), r = (l <= l')
  int n, K, L, nsplit;
  const float* Wc; const float* Gc;
  float* slabs;                        // [nsplit][tri_padded(K)][tri_padded(L)]
};
inline int ssys_gemm_wave_tiles(int K, int L) { return (tri_padded(K) / 64) * (tri_padded(L) / 64); }
void launch_ssys_gemm(const SSysGemmArgs& a, hipStream_t st);
struct SSysBArgs { int n, n0, K, L; const float* slabs; int split, n_pad; const float* G; float* b; };   // b[block][K L]: per 64-column block partials of sum_j Pv_jk G_jl
inline int ssys_b_blocks(int n) { return (n + 63) / 64 > 0 ? (n + 63) / 64 : 1; }
void launch_ssys_b(const SSysBArgs& a, hipStream_t st);
void launch_ssys_reduce(const float* slabs, int nsplit, int K, int L, float* A, hipStream_t st);   // sum the slabs on k <= k', mirror
void launch_ssys_sum_parts(const float* parts, int nparts, size_t n, float* out, hipStream_t st);
// r = b - A S.  bparts != nullptr: b is summed from its nparts per-block parts first (and stored).  cands [n2][4][4]: the chain's first candidates (draws)
// tinv != nullptr: K more blocks make Ti_k = (I + N_k)^-1 [K][32][32] for the chain's rows (needs K, L, the device's tau)
void launch_ssys_residual(const float* A, float* b, const float* bparts, int nparts, const float* S, int n2, float* r, hipStream_t st,
                          float* cands = nullptr, uint32_t it = 0, uint32_t key0 = 0, uint32_t key1 = 0,
                          float* tinv = nullptr, int K = 0, int L = 0, const float* tau = nullptr, float* own8 = nullptr, float* recT = nullptr, float* Tn = nullptr);
struct SSysChainArgs {
  int K, L, update, cond;              // update: 0 draw, else mode (clamped from below by min_x); cond >= 0: evaluate entry cond only
  float min_x;
  const float* A; const float* r0; float* S; const float* lambdaS; const float* tau;
  uint32_t key0, key1, it;
  const float4* cands;                 // [K L][4] {-log u1, z, u2, -} of iteration `it` (ssys_residual_kernel); draws only
  const float* Tinv;                   // [K][32][32]: (I + N_k)^-1 of every row's diagonal block (ssys_residual_kernel's extra blocks)
  const float4* own8; const float4* recT;   // [K L][2] per-entry and [K L][4] per-candidate constants of the draws (ssys_residual_kernel)
  const float* Tn;                     // [K L]: the row's own old values up to the entry put back (ssys_residual_kernel)
  double* numer_out; double* tau_out;
};
void launch_ssys_chain(const SSysChainArgs& a, hipStream_t st);


// ---------------------------------------------------------------------------
// variational tri-factorisation (kernel_trivb.hip); K, L <= 32
// ---------------------------------------------------------------------------
struct SmallProductVbArgs {  // effective factor: mean out = X.S (or X.S^T) and its second moment outS2
  const float* X; const float* varX; int rows; const float* S; const float* varS; int K, L, transposeS; float* out; float* outS2;
};
void launch_small_product_vb(const SmallProductVbArgs& a, hipStream_t st);
struct MaskedColsumArgs {    // out[u][c] = sum over the unit's observed inner indices of V[.][c]
  int n; const uint32_t* slot_ptr; const uint32_t* idx; const float* V; const double* colsum2; const double* C64; float* out;
};
void launch_masked_colsum(const MaskedColsumArgs& a, hipStream_t st);
struct SSysChainVbArgs {
  int K, L, n_order, only_params;
  const int* order;                    // entries a = k L + l in update order
  const float* A; const float* r0; const float* lambdaS; const float* tau;
  float* E; float* var; float* mu; float* tauq;     // q(S): expS (in/out), varS, muS, tauS  [K L]
  const float* Aperm;                  // whole passes: A~ in the pass's order (launch_ssys_permute), or null
};
void launch_ssys_chain_vb(const SSysChainVbArgs& a, hipStream_t st);
void launch_ssys_permute(const float* A, const int* order, int n, float* out, hipStream_t st);   // out[s][p] = A[order[s]][order[p]]
struct TriFactorArgs { int which, side, rows, K, L; const float* X; const float* varX; const float* S; const float* varS; double* out; };
void launch_tri_factors(const TriFactorArgs& a, hipStream_t st);
// exp_square_diff's third term (bnmtf_vb_optimised.py:238) from the masked variance sums the G step already holds:
//   sum_jk mv[j][k] ((sum_l S_kl G_jl)^2 - sum_l S_kl^2 G_jl^2),  mv[j][k] = sum_{i in Omega_j} varF_ik;  one partial sum per block
struct TriThirdArgs { int rows, K, L; const float* G; const float* S; const float* mv; double* part; };
int tri_third_blocks(int rows);
void launch_tri_third(const TriThirdArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------
// small models (kernel_small.hip): the WHOLE run(n) of a BNMF Gibbs model in one launch, one 16-wave block per model --
// both factors, the Gram matrices and every per-column exchange in the CU's LDS, q of the missing entries in registers,
// R~ streamed from L2 by the block's own f32 MFMA contraction; grid = number of models (the folds / ranks / restarts of a
// model search: code/cross_validation/line_search_cross_validation.py:54-131 run them one after the other).  Same
// conditionals, same Philox keying and candidate sequence as the large path (bnmf_gibbs_optimised.py:133-155, :167-177).
// ---------------------------------------------------------------------------
constexpr int kSmallThreads = 1024;     // threads of the largest block (16 waves = one CU); row stride of the slot tables.  Smaller models run 256- or 512-thread blocks, several to a CU
constexpr int kSmallStride = 33;        // floats per factor row in LDS (odd: a column gather and a row read are both conflict-free)
constexpr int kSmallKP = 32;            // K <= 32
constexpr int kSmallMaxSlots = 64;      // missing entries per entry thread (slot classes 8 / 16 / 32 / 64)
constexpr uint32_t kSmallNone = 0xFFFFFFFFu;
struct SmallDirDev {
  int n, m;                    // units (rows of this direction's factor), inner extent
  int ldb, ldn;                // leading dimensions: big [round_up(m, 4)][ldb], XT / PT [32][ldn]  (multiples of 64)
  int nthreads, em;            // entry threads in use; slots per entry thread
  const float* big;            // masked R, unit index contiguous (rows: big[j][i], cols: big[i][j])
  const float* lambda;         // [n][32] prior rates, zero padded
  const uint16_t* idx;         // [em][1024] 33 j for the inner index j of slot e of entry thread t (33 m: empty -> the zero row)
  const uint16_t* unit_of;     // [1024] unit of entry thread t
  const uint16_t* seg;         // [n][2] first entry thread and number of entry threads of unit u
  const uint32_t* perm;        // [em][1024] where the OTHER direction keeps this slot's q (e' * 1024 + t'), kSmallNone: empty
  float* X;                    // state [n][32] row major
  float* XT;                   // [32][ldn] the same transposed (column k's old values, coalesced over the units)
  float* q;                    // [em][1024] q of the missing entries as this direction's sweep left them
  uint2* tab;                  // [K nc0][ldn] Philox words of the first nc0 candidates of every (unit, column) of a half sweep (nc0 <= 8, nc0 ldn <= 1024)
  double* C64;                 // [32][32] Gram of X
  double* colsum;              // [32]
};
struct SmallLaunch {
  SmallDirDev rows, cols;
  float* PT;                   // [32][cols.ldn] Pv = R~^T U of the cols contraction (for sum_Omega R.Rp)
  int K;
  double n_obs, sumR, sumR2, alpha, beta;
  double* tau_d; float* tau_f;
  // this call
  int n_iter, update; float min_x;
  uint32_t key0, key1; unsigned long long it0;
  int q_valid;                 // cols.q holds q of the current state (left there by the previous call)
  uint32_t refresh;            // the rows sweep rebuilds q from the factors when iteration % refresh == 0
  const double* gunit;         // [n_iter] Gamma(alpha_s, 1) variates (draws)
  double* rec;                 // [n_iter][5] tau, MSE, R^2, Rp, SSE
  unsigned long long* clock;   // [n_iter + 1] wall_clock64 at the start and after every iteration
  float* U_s; float* V_s;      // samples [n_iter][n][K], or null
  double* expR; double* expC; double* exp_tau; int exp_burn, exp_thin;   // posterior sums (bnmtf_set_expectation), or null / -1
  // tri-factorisation (L > 0; bnmtf_gibbs_optimised.py:138-180): rows = F (I x K), cols = G (J x L), S (K x L) between them.  The F
  // sweep sees G S^T as its other factor, the G sweep F S (formed in place in LDS), the K L entries of S are updated one after the
  // other between the two by the whole block (kernel_small.hip, "S step")
  int L;
  float* S;                    // [K][L] state
  const float* lambdaS;        // [K][L] prior rates
  float* ZT;                   // [32][cols.ldn] Pv = R~^T F (k-major): b = Pv^T G of the S step, and (R~^T F) S is the G sweep's contraction
  uint2* stab;                 // [K L][4] Philox words of the first four candidates of every S draw of an iteration
  float* Wg;                   // [cols.n][56] packed second moments of F's rows over a column's missing entries (the dense S step, K, L <= 10)
  float* S_s;                  // samples [n_iter][K][L], or null
  double* expS;                // posterior sum [K][L], or null
};
// the S step's dense form (the K L x K L system in LDS, the chain on one wave: kernel_small.hip) takes the ranks the reference searches
constexpr int kTriDenseK = 10;
__host__ __device__ inline bool small_tri_dense(int K, int L) { return L > 0 && K <= kTriDenseK && L <= kTriDenseK; }
size_t small_lds_bytes(int I, int J, int nt, int K = 0, int L = 0);  // LDS a block of nt threads needs for an I x J model (K, L: the tri-factorisation's ranks)
void launch_small_gibbs(const SmallLaunch* dev_launches, int n_models, int em, int nt, size_t lds_bytes, hipStream_t st, bool tri = false);

// small helpers
void launch_sum_stats(const double* stats, int nblocks, double* acc, hipStream_t st);   // acc[0..2] += column sums of stats
void launch_sum_cols(const double* stats, int nrows, int ld, int ncols, double* out, hipStream_t st);        // out[c] += column sums, ncols <= 8
void launch_compact_rows(const float* X, int rows, int W, int KP, float* dst, hipStream_t st);               // [rows][KP] -> [rows][W]
void launch_accumulate(const float* X, size_t n, double* sum, const double* tau, double* tausum, hipStream_t st);   // sum += X (fp64), *tausum += *tau
void launch_transpose(const float* X, int rows, int KP, float* XT, int ldT, hipStream_t st);
void launch_tn_sample(const double* mu, const double* tau, size_t n, uint64_t seed, uint32_t it, uint32_t col,
                      uint32_t elem0, double* out, hipStream_t st);
void launch_tn_moments(const double* mu, const double* tau, size_t n, double* e, double* v, hipStream_t st);
void launch_gamma_sample(double alpha, double beta, uint64_t seed, uint32_t it, double* out, hipStream_t st);

}  // namespace bnmtf
