/*
 * bnmtf_hip.h -- C ABI of libbnmtf_hip.so: the MI355X (gfx950) implementation of
 * the Gibbs / VB inference hot path of ThomasBrouwer/BNMTF.
 *
 * The reference has no FFI layer: its boundary is the duck-typed Python class
 * contract of code/models/ bnmf_gibbs_optimised.py, bnmtf_gibbs_optimised.py, bnmf_vb_optimised.py.
 * Each entry point below names the reference method(s) it replaces (file:line,
 * relative to the reference checkout).  The Python package bnmtf_amd binds these with ctypes
 * and reproduces the class surface; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain C, no exceptions; every function returns 0 on success or a negative
 *     BNMTF_E* code; bnmtf_last_error() gives the message (thread-local).
 *   - host pointers are caller-owned, C-contiguous (row-major) and only touched
 *     during the call; every call is synchronous at return.
 *   - factor matrices cross the boundary as double (what the reference's
 *     attributes hold); the device computes the O(I*J*K) contractions with
 *     fp32-exact products on the bf16 matrix cores (every fp32 operand split
 *     into three bf16 terms, six products per fp32 product, fp32 accumulation)
 *     and every reduction that feeds tau / metrics in fp64.
 *   - one host thread per handle; a handle owns its device buffers and stream.
 *   - multi-GPU = one process (and one handle) per GPU, rows of R split over
 *     `world` ranks for the U/F sweep, columns for the V/G sweep, freshly drawn
 *     factor blocks exchanged with an RCCL all-gather.
 */
#ifndef BNMTF_HIP_H
#define BNMTF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: these are its only dynamic symbols */
#define BNMTF_API __attribute__((visibility("default")))

#define BNMTF_OK 0
#define BNMTF_EINVAL (-1)   /* bad argument / unsupported shape */
#define BNMTF_EHIP (-2)     /* HIP runtime error */
#define BNMTF_ENOMEM (-3)
#define BNMTF_ECOMM (-4)    /* RCCL error */
#define BNMTF_ESTATE (-5)   /* call order (e.g. run before set_state) */

#define BNMTF_MAX_RANK 64   /* K and L <= 64 (one wave lane per latent factor) */

typedef struct bnmtf_model* bnmtf_handle;

/* update rule applied to each conditional: Gibbs draw (the path), or the
 * deterministic mode max(0,mu) = the ICM update of nmf_icm.py:124-134, which
 * shares all tau/mu kernels and serves as an exact end-to-end parity harness. */
#define BNMTF_UPDATE_DRAW 0
#define BNMTF_UPDATE_MODE 1
#define BNMTF_UPDATE_ICM 2        /* nmf_icm / nmtf_icm: TN mode max(0, mu), clamped from below by minimum_TN; tau = gamma_mode */

typedef struct bnmtf_problem {
  int32_t I, J;              /* shape of R */
  int32_t K;                 /* latent factors (rows factor U / F) */
  int32_t L;                 /* 0: BNMF  R ~ U.V^T ;  >0: BNMTF  R ~ F.S.G^T */
  const float* R;            /* I x J, full matrix; entries with M==0 are kept only for predict() */
  const uint8_t* M;          /* I x J, 1 = observed, 0 = missing */
  const double* lambda_rows; /* I x K   lambdaU / lambdaF */
  const double* lambda_cols; /* J x K (lambdaV) or J x L (lambdaG) */
  const double* lambda_S;    /* K x L, BNMTF only (else NULL) */
  double alpha, beta;        /* Gamma prior of tau */
  uint64_t seed;             /* Philox key */
  int32_t device;            /* HIP device ordinal */
  int32_t rank, world;       /* this process' shard; world == 1: single GPU */
  const uint8_t* comm_id;    /* 128-byte id from bnmtf_comm_unique_id (world > 1) */
} bnmtf_problem;

/* ---- library ---------------------------------------------------------- */
BNMTF_API int bnmtf_version(void);
BNMTF_API const char* bnmtf_last_error(void);
BNMTF_API int bnmtf_device_count(int* count);
/* the contiguous block of `n` units (rows for the U/F sweep, columns for V/G) owned by `rank`:
 * first = n*rank/world, count = n*(rank+1)/world - first.  Pure function, no GPU needed. */
BNMTF_API int bnmtf_shard_range(int64_t n, int rank, int world, int64_t* first, int64_t* count);
/* rank 0 makes the id, the host code ships it to the other ranks */
BNMTF_API int bnmtf_comm_unique_id(uint8_t out[128]);

/* ---- life cycle: bnmf_gibbs_optimised.__init__ (bnmf_gibbs_optimised.py:54-78),
 *      bnmtf_gibbs_optimised.__init__ (bnmtf_gibbs_optimised.py:56-84).  Shape /
 *      empty-row assertions stay in the Python class (exact messages); create
 *      re-checks them and fails with BNMTF_EINVAL. ---------------------- */
BNMTF_API int bnmtf_create(const bnmtf_problem* p, bnmtf_handle* out);
BNMTF_API int bnmtf_destroy(bnmtf_handle h);
BNMTF_API int bnmtf_sync(bnmtf_handle h);

/* size_Omega and the per-row / per-column observed counts (bit-exact integers;
 * size_Omega = M.sum(), bnmf_gibbs_optimised.py:65; counts :83-84).
 * row/col may be NULL. */
BNMTF_API int bnmtf_omega_counts(bnmtf_handle h, uint64_t* total, uint32_t* row, uint32_t* col);

/* Page-locked host memory for the sample arrays run() fills (all_U, all_V, ...; bnmf_gibbs_optimised.py:125-127,
 * 146-148).  *_gibbs_run accepts ANY host pointer for its sample outputs; buffers from here (or registered by the
 * caller with hipHostRegister) are written by asynchronous device-to-host copies that overlap the following
 * iterations, pageable ones go through an internal pinned ring and a host memcpy. */
BNMTF_API int bnmtf_host_alloc(size_t bytes, void** out);
BNMTF_API int bnmtf_host_free(void* p);

/* approx_expectation(burn_in, thinning) (bnmf_gibbs_optimised.py:182-187, bnmtf_gibbs_optimised.py:216-223) on the
 * device: with burn_in >= 0 every *_gibbs_run call sums (fp64) the samples of its iterations burn_in, burn_in +
 * thinning, ... and bnmtf_get_expectation returns their means -- no iterations x I x K array crosses to the host (the
 * model-selection drivers of code/cross_validation/ only need these means).  burn_in < 0 switches it off (default).
 * A: I x K, S: K x L (BNMTF, else ignored), B: J x K (or J x L); any may be NULL. */
BNMTF_API int bnmtf_set_expectation(bnmtf_handle h, int burn_in, int thinning);
BNMTF_API int bnmtf_get_expectation(bnmtf_handle h, double* A, double* S, double* B, double* tau, uint64_t* count);

/* Gibbs iteration counter = RNG counter word 2 (continues across run calls). */
BNMTF_API int bnmtf_set_iteration(bnmtf_handle h, uint64_t it);
/* tau alone (self.tau = ... between half sweeps: the factors on the device stay as they are) */
BNMTF_API int bnmtf_set_tau(bnmtf_handle h, double tau);
BNMTF_API int bnmtf_get_iteration(bnmtf_handle h, uint64_t* it);

/* ---- BNMF Gibbs state: attributes U, V, tau (bnmf_gibbs_optimised.py:102-117) */
BNMTF_API int bnmf_set_state(bnmtf_handle h, const double* U, const double* V, double tau);
BNMTF_API int bnmf_get_state(bnmtf_handle h, double* U, double* V, double* tau);

/* tauU(k)/muU(tauUk,k) (which=0, :167-171) and tauV(k)/muV (which=1, :173-177) for
 * the current state, through the same kernels the sampler uses.  numer_out is
 * (-lambda[:,k] + tau * sum(...)), so that mu = numer / tau_k as the reference
 * divides by the caller-supplied tauUk; tau_out is tauU(k). Length I (or J). */
BNMTF_API int bnmf_cond_params(bnmtf_handle h, int which, int k, double* numer_out, double* tau_out);

/* beta_s() (:164-165) for the current state: beta + 0.5 * masked SSE */
BNMTF_API int bnmtf_beta_s(bnmtf_handle h, double* out);

/* run(iterations) (:121-157): n_iter full sweeps (U columns, V columns, tau draw,
 * metrics on the training mask).  Outputs, each may be NULL:
 *   U_out [n_iter][I][K], V_out [n_iter][J][K]   (all_U / all_V, fp32 samples)
 *   tau_out [n_iter]                             (all_tau)
 *   perf_out [n_iter][3]  MSE, R^2, Rp           (all_performances)
 *   times_out [n_iter]    cumulative seconds     (all_times)           */
BNMTF_API int bnmf_gibbs_run(bnmtf_handle h, int n_iter, int update, float* U_out, float* V_out,
                   double* tau_out, double* perf_out, double* times_out);

/* The same call for n_models independent models at once -- the folds x ranks x restarts a model search fits one after the
 * other (code/cross_validation/line_search_cross_validation.py:54-131, line_search_bnmf.py:53-76,
 * parallel_matrix_cross_validation.py:40-74).  Models on the one-launch path (small models: DESIGN.md section 4,
 * kernel_small.hip -- one block per model, the whole run in one launch) that share a device go down in ONE grid; any other
 * handle in the list is run by bnmf_gibbs_run in turn.  Every model draws exactly the chain its own bnmf_gibbs_run call
 * would draw.  The output arrays are arrays of n_models pointers (or NULL), each as in bnmf_gibbs_run; U_final [I][K], V_final [J][K],
 * tau_final [1] (doubles): the state each model ends with -- what bnmf_get_state would return, fetched for the whole batch with one
 * synchronisation. */
BNMTF_API int bnmf_gibbs_run_many(const bnmtf_handle* hs, int n_models, int n_iter, int update, float* const* U_outs, float* const* V_outs,
                        double* const* tau_outs, double* const* perf_outs, double* const* times_outs,
                        double* const* U_final, double* const* V_final, double* const* tau_final);

/* ---- ranks above 64: a BNMF factorisation as column blocks (round 6) -------
 * The reference takes any K (code/models/bnmf_gibbs_optimised.py:54-78).  The kernels hold a latent factor per wave lane
 * (K <= BNMTF_MAX_RANK), so a wider model runs as ceil(K / 64) column blocks, one handle each, driven by the host class
 * (bnmtf_amd/_blocked.py): the conditionals of block b's columns given the other blocks are those of a rank-K_b model on
 * the residual data R - sum_{b' != b} U_b' V_b'^T, so the blocks' half sweeps in turn -- all blocks' rows (:134-137), then
 * all blocks' columns (:139-142) -- are the reference's sequential column order.  One GPU, BNMF Gibbs / ICM. */
/* this handle holds columns [col0, col0 + K) of the wider model: the Philox column word of its column k is col0 + k (the
 * draws are keyed by the wide model's column index: oracle/rng.py); no one-launch path, no q hand-over */
BNMTF_API int bnmf_set_column_block(bnmtf_handle h, int col0);
/* the contraction operands of h become M . (R - sum_b U_b V_b^T) over the n_others (<= 4) BNMF handles' current factors
 * (n_others = 0: M . R again); predict() / metric entry points keep the full R.  h itself may be a BNMTF handle (a block of
 * the S of a wider tri-factorisation, below) */
BNMTF_API int bnmf_set_residual_data(bnmtf_handle h, const bnmtf_handle* others, int n_others);
/* one half of an iteration of run(): contraction, the K sequential column updates of U (which = 0, :134-137) or V
 * (which = 1, :139-142) with the handle's current tau and iteration counter, then the relayout + Gram the other half reads.
 * tau, metrics, samples and the iteration counter are the caller's. */
BNMTF_API int bnmf_half_sweep(bnmtf_handle h, int which, int update);
/* bnmtf_metric_sums for explicit factors of ANY width Kc: A [I][Kc], B [J][Kc] (compute_MSE / R2 / Rp, :208-223, and
 * beta_s, :164-165, of a column-blocked model) */
BNMTF_API int bnmtf_metric_sums_wide(bnmtf_handle h, const uint8_t* Mp, const double* A, const double* B, int Kc, double sums_out[6]);

/* the variational model (bnmf_vb_optimised.py) as column blocks: one half of an iteration (:134-141) -- update_U(k) +
 * update_exp_U(k) for every k (which = 0) or the same for V -- with the handle's current exptau (bnmtf_set_tau); and
 * exp_square_diff() (:185-187) in its two parts, out[0] = sum_Omega (R - E[U] E[V]^T)^2, out[1] = the second-moment sum
 * over the handle's own columns (additive over blocks) */
BNMTF_API int bnmf_vb_half_sweep(bnmtf_handle h, int which);
BNMTF_API int bnmf_vb_esd_terms(bnmtf_handle h, double out[2]);

/* the tri-factorisation with K or L above 64 (bnmtf_gibbs_optimised.py:56-84 takes any; bnmtf_amd/_blocked.py: TriBlocks).  F's
 * column blocks and G's are BNMF handles against the effective factors (G S_b^T, F S_c), driven through the entry points above;
 * block (b, c) of S is a BNMTF handle (F_b, S_bc, G_c) on the data minus what every other block of S explains:
 * this handle is rows row0.., columns col0.. of a K x L_wide S -- its draws are keyed by their place there (oracle/rng.py: the
 * Philox column word of S_kl is k L + l); its S step runs row by row, no one-launch path, no dense system */
BNMTF_API int bnmtf_set_s_block(bnmtf_handle h, int row0, int col0, int L_wide);
/* rows k0 .. k1-1 of the handle's S, every l in order (:157-160), for its current F, G, tau, iteration counter and data */
BNMTF_API int bnmtf_s_rows(bnmtf_handle h, int k0, int k1, int update);

/* ---- BNMTF Gibbs (bnmtf_gibbs_optimised.py) ---------------------------- */
BNMTF_API int bnmtf_set_state(bnmtf_handle h, const double* F, const double* S, const double* G, double tau);
BNMTF_API int bnmtf_get_state(bnmtf_handle h, double* F, double* S, double* G, double* tau);
/* which = 0: tauF(k)/muF (:195-199), length I; 1: tauS(k,l)/muS (:201-205), length 1;
 * 2: tauG(l)/muG (:207-211), length J (k ignored). */
BNMTF_API int bnmtf_cond_params(bnmtf_handle h, int which, int k, int l, double* numer_out, double* tau_out);
/* run(iterations) (:138-180): F columns, S row-major, G columns, tau. */
BNMTF_API int bnmtf_gibbs_run(bnmtf_handle h, int n_iter, int update, float* F_out, float* S_out, float* G_out,
                    double* tau_out, double* perf_out, double* times_out);

/* bnmtf_gibbs_run for n_models independent tri-factorisations at once: the folds and the (K, L) neighbours a greedy model search
 * fits one after the other (code/cross_validation/greedy_search_cross_validation.py:60-130,
 * experiments/experiments_gdsc/cross_validation/gibbs_nmtf/greedysearch_xval_gibbs.py:17-60).  Models of the one-launch path
 * (K, L <= 32, I, J <= 1024, factors within one CU's LDS: kernel_small.hip) that share a device go down in ONE grid, a block per
 * model -- F sweep, the K L entries of S, G sweep, tau, the metrics of every iteration inside the launch; any other handle is run
 * by bnmtf_gibbs_run in turn.  Every model draws the chain its own bnmtf_gibbs_run call would draw.  Arrays of n_models pointers
 * (or NULL), each as in bnmtf_gibbs_run; F_final [I][K], S_final [K][L], G_final [J][L], tau_final [1] (doubles): the state each
 * model ends with, fetched for the whole batch with one synchronisation. */
BNMTF_API int bnmtf_gibbs_run_many(const bnmtf_handle* hs, int n_models, int n_iter, int update, float* const* F_outs, float* const* S_outs,
                         float* const* G_outs, double* const* tau_outs, double* const* perf_outs, double* const* times_outs,
                         double* const* F_final, double* const* S_final, double* const* G_final, double* const* tau_final);

/* ---- BNMF VB (bnmf_vb_optimised.py) ------------------------------------ */
/* all eight q-parameter matrices + exptau; any pointer may be NULL (left as is) */
BNMTF_API int bnmf_vb_set_state(bnmtf_handle h, const double* muU, const double* tauU, const double* expU, const double* varU,
                      const double* muV, const double* tauV, const double* expV, const double* varV, double exptau);
BNMTF_API int bnmf_vb_get_state(bnmtf_handle h, double* muU, double* tauU, double* expU, double* varU,
                      double* muV, double* tauV, double* expV, double* varV);
/* update_U(k)+update_exp_U(k) (which=0, :189-191,199-204) or update_V/update_exp_V
 * (which=1, :193-195,206-211) for one column; moments=0 skips the update_exp part. */
BNMTF_API int bnmf_vb_update(bnmtf_handle h, int which, int k, int moments);
/* exp_square_diff() (:185-187) */
BNMTF_API int bnmf_vb_exp_square_diff(bnmtf_handle h, double* out);
/* Hook (tests): the two chain-independent masked sums that update_U(k) / update_V(k) use for every (unit, column) of one
 * direction (which = 0: rows i, other factor V; which = 1: columns j, other factor U), formed the way run() forms them on its
 * on-chip path (csrc/kernel_maskgemm.hip: the mask's bits x fixed-point digit planes of the moments on the int8 matrix cores):
 *   asq[u][k] = sum_{r: M = 0} (varO[r][k] + expO[r][k]^2)     (tauU[i][k] = exptau * (sum_j S2V[j][k] - asq[i][k]),  :189-199)
 *   vsq[u][k] = sum_{r: M = 0} expO[r][k]^2                     (the unit's own term of the numerator of muU)
 * for the rank's local units, [n][K] doubles each. */
BNMTF_API int bnmf_vb_masked_sums(bnmtf_handle h, int which, double* asq, double* vsq);
/* run(iterations) (:121-153).  exptau_out[n_iter] (all_exp_tau), perf_out[n_iter][3], times_out[n_iter],
 * elbo_terms_out[n_iter][10] = the O(I*J) / O((I+J)K) pieces of elbo() (:163-177) that live on the device:
 *   {exp_square_diff, beta_s,
 *    sum tauU/2 (varU+(expU-muU)^2), sum log(0.5 erfc(-muU sqrt(tauU/2))), sum log tauU, sum lambdaU expU,
 *    the same four for V};  the host finishes the scalar algebra (digamma, gammaln). */
BNMTF_API int bnmf_vb_run(bnmtf_handle h, int n_iter, double* exptau_out, double* perf_out,
                double* elbo_terms_out, double* times_out);
/* run(iterations) of n_models variational models on one device, walked in lock-step: every kernel of an iteration is ONE launch
 * for all of them (csrc/many.h, api_many.inc) -- the reference's model searches (experiments_gdsc/cross_validation/vb_nmf/
 * linesearch_xval_vb.py:17-52) are dozens of independent models of 622 x 138.  Every model ends with the bits of its own
 * bnmf_vb_run.  Outputs model-major ([n_models][n_iter]..., as bnmf_vb_run's; times = the batch's clock); any may be null.  Models
 * that cannot share launches (several GPUs, the 16-wave shapes of large problems, kernel timers on) run one after the other.
 * launch_info (optional, 2 ints): the models that shared launches, the argument-list uploads. */
BNMTF_API int bnmf_vb_run_many(bnmtf_handle* hs, int n_models, int n_iter, double* exptau_out, double* perf_out,
                     double* elbo_terms_out, double* times_out, int* launch_info);

/* ---- BNMTF VB (bnmtf_vb_optimised.py); K, L <= 32, one GPU ---------------- */
/* the twelve q-parameter matrices (F: I x K, S: K x L, G: J x L) + exptau; any pointer may be NULL (left as is) */
BNMTF_API int bnmtf_vb_set_state(bnmtf_handle h, const double* muF, const double* tauF, const double* expF, const double* varF,
                       const double* muS, const double* tauS, const double* expS, const double* varS,
                       const double* muG, const double* tauG, const double* expG, const double* varG, double exptau);
BNMTF_API int bnmtf_vb_get_state(bnmtf_handle h, double* muF, double* tauF, double* expF, double* varF,
                       double* muS, double* tauS, double* expS, double* varS,
                       double* muG, double* tauG, double* expG, double* varG);
/* update_F(k) (which = 0, :241-250), update_S(k,l) (which = 1, :252-262), update_G(l) (which = 2, :264-273) for the
 * current state; moments != 0 also runs the matching update_exp_* (:276-285). */
BNMTF_API int bnmtf_vb_update(bnmtf_handle h, int which, int k, int l, int moments);
/* exp_square_diff() (:235-239); sums_out (may be NULL): n, sum R, sum R^2, sum P, sum P^2, sum R P over the training
 * mask with P = E[F] E[S] E[G]^T */
BNMTF_API int bnmtf_vb_exp_square_diff(bnmtf_handle h, double* esd_out, double sums_out[6]);
/* run(iterations) (:160-205).  orders [n_iter][K L + K + L]: per iteration the update order of the S entries (k L + l),
 * the F columns and the G columns (the reference shuffles them with random.shuffle, :171-186; the host class draws the
 * same shuffles).  exptau_out [n_iter]; perf_out [n_iter][3]; times_out [n_iter]; elbo_terms_out [n_iter][10] =
 * {exp_square_diff, beta_s, then for F and for G: sum tau/2 (var+(exp-mu)^2), sum log(0.5 erfc(-mu sqrt(tau/2))),
 * sum log tau, sum lambda exp} -- the K L terms of S and the scalar algebra are the host's. */
BNMTF_API int bnmtf_vb_run(bnmtf_handle h, int n_iter, const int32_t* orders, double* exptau_out, double* perf_out,
                 double* elbo_terms_out, double* times_out);

/* ---- metrics: predict()/predict_while_running()/quality('MSE')/log_likelihood
 *      (:191-223,247-251).  Six fp64 sums over mask Mp (I x J bytes; NULL = the
 *      training mask): n, sum R, sum R^2, sum P, sum P^2, sum R*P with
 *      P = A.B^T (S == NULL, A: I x K, B: J x K) or A.S.B^T (S: K x L, B: J x L).
 *      A == NULL uses the handle's current state. */
BNMTF_API int bnmtf_metric_sums(bnmtf_handle h, const uint8_t* Mp, const double* A, const double* S,
                      const double* B, double sums_out[6]);

/* ---- distributions (stand-alone hooks; code/models/distributions/) ----- */
/* TN_vector_draw (truncated_normal_vector.py:37-50): out[e] ~ TN(mu[e],tau[e]) on
 * [0,inf); RNG counter (elem0+e, col, it, STREAM_HOOK). */
BNMTF_API int bnmtf_tn_sample(const double* mu, const double* tau, size_t n, uint64_t seed, uint64_t it,
                    uint32_t col, uint32_t elem0, int device, double* out);
/* TN_vector_expectation / TN_vector_variance (:53-73) */
BNMTF_API int bnmtf_tn_moments(const double* mu, const double* tau, size_t n, int device,
                     double* exp_out, double* var_out);
/* gamma_draw (gamma.py:11-14), shape alpha, rate beta */
BNMTF_API int bnmtf_gamma_sample(double alpha, double beta, uint64_t seed, uint64_t it, int device, double* out);

/* ---- masked K-means for initialise(init_FG='kmeans') (code/models/kmeans/kmeans.py): the two O(points x coordinates x K)
 *      passes of an iteration; X [n_points][n_coords] fp64 (as the reference computes), M 0/1 bytes; the handle is a plain pointer of its own kind. */
BNMTF_API int bnmtf_kmeans_create(const double* X, const uint8_t* M, int n_points, int n_coords, int K, int device, void** out);
BNMTF_API int bnmtf_kmeans_destroy(void* h);
/* assignment() (kmeans.py:87-119): closest centroid by MSE over the shared observed coordinates (none: infinitely far;
 * ties: lowest index); dist_out = that MSE (+inf without overlap) */
BNMTF_API int bnmtf_kmeans_assign(void* h, const double* centroids, const uint8_t* mask_centroids, int32_t* assign_out, double* dist_out);
/* the sums update() needs (kmeans.py:126-182): cnt_out / tot_out [K][n_coords] = number / sum of the values of the
 * cluster's members that observe the coordinate (assign < 0: the point belongs to no cluster) */
BNMTF_API int bnmtf_kmeans_sums(void* h, const int32_t* assign, double* cnt_out, double* tot_out);
/* X[index][:] = values: `self.centroids[c] = self.X[index]` (kmeans.py:141) makes the refilled centroid a view of the data
 * point, so the means written into the centroid afterwards (:158-163) change the point; the host class mirrors that */
BNMTF_API int bnmtf_kmeans_set_row(void* h, int index, const double* values);

/* ---- measurement aids (bench.py) --------------------------------------- */
#define BNMTF_KERNEL_GEMM_ROWS 0   /* P  = R~ . V      (U/F step numerators)        */
#define BNMTF_KERNEL_GEMM_COLS 1   /* Pv = R~^T . U    ("the U^T.R step")          */
#define BNMTF_KERNEL_SWEEP_ROWS 2
#define BNMTF_KERNEL_SWEEP_COLS 3
#define BNMTF_KERNEL_SWEEP_S 4      /* BNMTF: the K*L sequential S entries */
#define BNMTF_KERNEL_COUNT 8
/* enable = 1: run() brackets each launch of the listed kernels with hipEvents on the handle's
 * stream; enable = 2 + k (+ 32 (n - 1)): only kernel k (BNMTF_KERNEL_*), in every n-th iteration (an event record
 * costs the queue a few microseconds: a timed run samples); 0: off.  Totals are read back with bnmtf_kernel_stats. */
BNMTF_API int bnmtf_set_profiling(bnmtf_handle h, int enable);
/* ICM: the lower clamp run(iterations, minimum_TN) applies to every updated entry (nmf_icm.py:129,135;
 * nmtf_icm.py:147,153,159).  Used by *_gibbs_run with update = BNMTF_UPDATE_ICM.  Default 0. */
BNMTF_API int bnmtf_set_minimum_tn(bnmtf_handle h, double minimum_TN);
/* sweep kernel selection: 1 (default) = register/LDS-resident fast path when the shape
 * fits, 0 = always the generic kernel (any mask, q in global memory).  Same results. */
BNMTF_API int bnmtf_set_sweep_path(bnmtf_handle h, int fast);
/* the one-launch path for small models, two- and three-factor (K, L <= 32, I, J <= 1024, factors within one CU's LDS; run() = one
 * launch, one block per model).  mode 1 (default): taken when it is the faster way to run the call -- always for the models of
 * 256- and 512-thread blocks, for the ones that fill a CU (1024-thread blocks) from three models per bnmf_gibbs_run_many call on
 * (two per bnmtf_gibbs_run_many call; a tri-factorisation with a rank above 10 -- the sequential form of its S step -- only in a batch of at
 * least (K L + 50) / 80 models); 0: never (the multi-launch path); 2: always.  Same chain either way up to fp32 summation order.
 * bnmtf_is_small: would a bnmf_gibbs_run / bnmtf_gibbs_run of this handle alone take it? */
BNMTF_API int bnmtf_set_small_path(bnmtf_handle h, int mode);
/* the handle's communicator as it reports itself: kind 0 none (one GPU), 1 RCCL (ranks = ncclCommCount), 2 the in-process test
 * transport; bench.py prints both next to its own world size */
BNMTF_API int bnmtf_comm_info(bnmtf_handle h, int* kind, int* ranks);
/* 1 when the library was built with `make EXPERIMENTS=1`: the measured-and-not-adopted kernels (two unit groups taking turns,
 * BNMTF_TURNS=1; two 8-wave blocks per CU, BNMTF_TWIN=1; the f32-MFMA contraction, BNMTF_GEMM=f32 -- DESIGN.md section 7) are
 * compiled in and their switches honoured.  The shipped build has none of them. */
BNMTF_API int bnmtf_has_experiments(void);
BNMTF_API int bnmtf_is_small(bnmtf_handle h, int* out);
BNMTF_API int bnmtf_kernel_stats(bnmtf_handle h, int kernel, double* total_ms, uint64_t* launches);
/* geometry of the last create: padded shapes, split factor, slot counts (for DESIGN/bench) */
BNMTF_API int bnmtf_describe(bnmtf_handle h, char* buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* BNMTF_HIP_H */
