// Test infrastructure (CPU box only): drives the C ABI of the library's HOST side, linked against hip_stub.cpp instead of the
// HIP runtime, under AddressSanitizer + UBSan (`make asan`) or ThreadSanitizer (`make tsan`).  Kernels do not run, so no number
// that comes back means anything; what is checked is the host code the calls go through: bnmtf_create's layout passes (worker
// threads, slot tables, shard ranges, the hand-over table sizes), the arenas and pools of created / destroyed models, every
// host<->"device" copy's extent (device memory is heap memory here), the sample ring of run(), run_many's batching, and the
// in-process multi-rank rendezvous of comm.hip with one host thread per rank.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "bnmtf_hip.h"

extern "C" long hipstub_launches();
extern "C" long hipstub_live_allocs();
extern "C" void hipstub_throw_at_sync(long n);

#define OK(expr)                                                                                     \
  do {                                                                                               \
    const int rc_ = (expr);                                                                          \
    if (rc_ != BNMTF_OK) { fprintf(stderr, "FAILED %s -> %d (%s)\n", #expr, rc_, bnmtf_last_error()); exit(2); } \
  } while (0)
#define EXPECT_ERR(expr)                                                                             \
  do {                                                                                               \
    const int rc_ = (expr);                                                                          \
    if (rc_ == BNMTF_OK) { fprintf(stderr, "expected an error: %s\n", #expr); exit(2); }             \
  } while (0)

struct Data {
  int I, J;
  std::vector<float> R;
  std::vector<uint8_t> M;
};
static Data make_data(int I, int J, double missing, unsigned seed) {
  Data d{I, J, std::vector<float>((size_t)I * J), std::vector<uint8_t>((size_t)I * J, 1)};
  std::mt19937 g(seed);
  std::uniform_real_distribution<float> u(0.f, 10.f);
  std::uniform_real_distribution<double> p(0.0, 1.0);
  for (auto& x : d.R) x = u(g);
  for (auto& m : d.M) m = p(g) < missing ? 0 : 1;
  for (int i = 0; i < I; ++i) d.M[(size_t)i * J + (i % J)] = 1;          // no empty row ...
  for (int j = 0; j < J; ++j) d.M[(size_t)(j % I) * J + j] = 1;          // ... or column
  return d;
}

static bnmtf_handle create(const Data& d, int K, int L, int rank, int world, const uint8_t* cid, uint64_t seed = 7) {
  std::vector<double> lr((size_t)d.I * K, 0.1), lc((size_t)d.J * (L ? L : K), 0.1), ls((size_t)K * (L ? L : 1), 0.1);
  bnmtf_problem p;
  memset(&p, 0, sizeof p);
  p.I = d.I; p.J = d.J; p.K = K; p.L = L; p.R = d.R.data(); p.M = d.M.data();
  p.lambda_rows = lr.data(); p.lambda_cols = lc.data(); p.lambda_S = L ? ls.data() : nullptr;
  p.alpha = p.beta = 1.0; p.seed = seed; p.device = 0; p.rank = rank; p.world = world; p.comm_id = cid;
  bnmtf_handle h = nullptr;
  OK(bnmtf_create(&p, &h));
  return h;
}

static void bnmf_round_trip(const Data& d, int K, int iters, bool samples) {
  bnmtf_handle h = create(d, K, 0, 0, 1, nullptr);
  std::vector<double> U((size_t)d.I * K, 1.0), V((size_t)d.J * K, 1.0);
  OK(bnmf_set_state(h, U.data(), V.data(), 1.0));
  uint64_t total = 0;
  std::vector<uint32_t> row(d.I), col(d.J);
  OK(bnmtf_omega_counts(h, &total, row.data(), col.data()));
  uint64_t expect = 0;
  for (uint8_t m : d.M) expect += m;
  if (total != expect) { fprintf(stderr, "omega count %llu != %llu\n", (unsigned long long)total, (unsigned long long)expect); exit(2); }
  std::vector<float> Uo, Vo;
  if (samples) { Uo.resize((size_t)iters * d.I * K); Vo.resize((size_t)iters * d.J * K); }
  std::vector<double> tau(iters), perf((size_t)iters * 3), times(iters);
  OK(bnmf_gibbs_run(h, iters, BNMTF_UPDATE_DRAW, samples ? Uo.data() : nullptr, samples ? Vo.data() : nullptr, tau.data(), perf.data(), times.data()));
  OK(bnmf_gibbs_run(h, 2, BNMTF_UPDATE_MODE, nullptr, nullptr, nullptr, nullptr, nullptr));
  std::vector<double> num(d.I), tp(d.I);
  OK(bnmf_cond_params(h, 0, K - 1, num.data(), tp.data()));
  EXPECT_ERR(bnmf_cond_params(h, 0, K, num.data(), tp.data()));
  OK(bnmtf_set_expectation(h, 1, 2));
  OK(bnmf_gibbs_run(h, 5, BNMTF_UPDATE_DRAW, nullptr, nullptr, nullptr, nullptr, nullptr));
  double t = 0; uint64_t cnt = 0;
  OK(bnmtf_get_expectation(h, U.data(), nullptr, V.data(), &t, &cnt));
  OK(bnmf_get_state(h, U.data(), V.data(), &t));
  // the metric entry points: the handle's own operand copies (part of a small model's arena), and factors wider than 64 columns
  double sums[6];
  OK(bnmtf_beta_s(h, &t));
  OK(bnmtf_metric_sums(h, nullptr, U.data(), nullptr, V.data(), sums));
  {
    const int Kw = 96;
    std::vector<double> A((size_t)d.I * Kw, 0.5), B((size_t)d.J * Kw, 0.5);
    OK(bnmtf_metric_sums_wide(h, d.M.data(), A.data(), B.data(), Kw, sums));
    OK(bnmtf_metric_sums(h, nullptr, U.data(), nullptr, V.data(), sums));
  }
  char buf[2048];
  OK(bnmtf_describe(h, buf, sizeof buf));
  OK(bnmtf_destroy(h));
}

// a C++ exception inside an entry point comes back as a status with its message; the handle is still good afterwards
static void exception_stays_inside(const Data& d, int K) {
  bnmtf_handle h = create(d, K, 0, 0, 1, nullptr);
  std::vector<double> U((size_t)d.I * K, 1.0), V((size_t)d.J * K, 1.0);
  OK(bnmf_set_state(h, U.data(), V.data(), 1.0));
  hipstub_throw_at_sync(1);
  const int rc = bnmf_gibbs_run(h, 2, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rc != BNMTF_EINVAL || !strstr(bnmtf_last_error(), "injected by the stub")) { fprintf(stderr, "exception guard: rc %d, message '%s'\n", rc, bnmtf_last_error()); exit(1); }
  hipstub_throw_at_sync(0);
  OK(bnmf_gibbs_run(h, 2, 0, nullptr, nullptr, nullptr, nullptr, nullptr));
  OK(bnmtf_destroy(h));
}

static void vb_round_trip(const Data& d, int K) {
  bnmtf_handle h = create(d, K, 0, 0, 1, nullptr);
  std::vector<double> a((size_t)d.I * K, 1.0), b((size_t)d.J * K, 1.0);
  OK(bnmf_vb_set_state(h, a.data(), a.data(), a.data(), a.data(), b.data(), b.data(), b.data(), b.data(), 1.0));
  std::vector<double> et(4), perf(12), times(4), elbo(40);
  OK(bnmf_vb_run(h, 4, et.data(), perf.data(), elbo.data(), times.data()));
  double e = 0;
  OK(bnmf_vb_exp_square_diff(h, &e));
  OK(bnmtf_destroy(h));
}

// several variational models in lock-step (api_many.inc): the recorder, the argument lists, the per-model outputs; two shapes and a
// model given once only
static void vb_many(const Data& d1, const Data& d2, int K) {
  std::vector<bnmtf_handle> hs;
  const long live0 = hipstub_live_allocs();
  for (int m = 0; m < 5; ++m) {
    const Data& d = m == 3 ? d2 : d1;
    bnmtf_handle h = create(d, m == 1 ? K + 3 : K, 0, 0, 1, nullptr);
    const int Km = m == 1 ? K + 3 : K;
    std::vector<double> a((size_t)d.I * Km, 1.0), b((size_t)d.J * Km, 1.0);
    OK(bnmf_vb_set_state(h, a.data(), a.data(), a.data(), a.data(), b.data(), b.data(), b.data(), b.data(), 1.0));
    hs.push_back(h);
  }
  const int n = (int)hs.size(), it = 3;
  std::vector<double> et((size_t)n * it), perf((size_t)n * it * 3), elbo((size_t)n * it * 10), times((size_t)n * it);
  int info[2];
  OK(bnmf_vb_run_many(hs.data(), n, it, et.data(), perf.data(), elbo.data(), times.data(), info));
  OK(bnmf_vb_run_many(hs.data(), n, 2, nullptr, nullptr, nullptr, nullptr, nullptr));
  OK(bnmf_vb_run_many(hs.data(), 1, 2, et.data(), nullptr, nullptr, nullptr, info));
  const long live1 = hipstub_live_allocs();
  for (bnmtf_handle h : hs) OK(bnmtf_destroy(h));
  if (getenv("SAN_VERBOSE")) printf("vb_many: live allocations %ld before, %ld with five models, %ld after\n", live0, live1, hipstub_live_allocs());
}

// the variational tri-factorisation: set_state / run with shuffled orders / the direct exp_square_diff / single updates / get_state
static void tri_vb_round_trip(const Data& d, int K, int L, int iters) {
  bnmtf_handle h = create(d, K, L, 0, 1, nullptr);
  std::vector<double> F((size_t)d.I * K, 1.0), S((size_t)K * L, 1.0), G((size_t)d.J * L, 1.0);
  OK(bnmtf_vb_set_state(h, F.data(), F.data(), F.data(), F.data(), S.data(), S.data(), S.data(), S.data(), G.data(), G.data(), G.data(), G.data(), 1.0));
  const int per = K * L + K + L;
  std::vector<int32_t> orders((size_t)iters * per);
  for (int it = 0; it < iters; ++it) {
    int32_t* o = &orders[(size_t)it * per];
    for (int a = 0; a < K * L; ++a) o[a] = K * L - 1 - a;
    for (int k = 0; k < K; ++k) o[K * L + k] = (k + it) % K;
    for (int l = 0; l < L; ++l) o[K * L + K + l] = L - 1 - l;
  }
  std::vector<double> et(iters), perf((size_t)iters * 3), times(iters), elbo((size_t)iters * 10);
  OK(bnmtf_vb_run(h, iters, orders.data(), et.data(), perf.data(), elbo.data(), times.data()));
  double e = 0, sums[6];
  OK(bnmtf_vb_exp_square_diff(h, &e, sums));
  OK(bnmtf_vb_update(h, 0, K - 1, 0, 1));
  OK(bnmtf_vb_update(h, 1, K - 1, L - 1, 1));
  OK(bnmtf_vb_update(h, 2, 0, L - 1, 0));
  OK(bnmtf_vb_get_state(h, F.data(), F.data(), F.data(), F.data(), S.data(), S.data(), S.data(), S.data(), G.data(), G.data(), G.data(), G.data()));
  OK(bnmtf_destroy(h));
}

static void tri_round_trip(const Data& d, int K, int L, int iters) {
  bnmtf_handle h = create(d, K, L, 0, 1, nullptr);
  std::vector<double> F((size_t)d.I * K, 1.0), S((size_t)K * L, 1.0), G((size_t)d.J * L, 1.0);
  OK(bnmtf_set_state(h, F.data(), S.data(), G.data(), 1.0));
  std::vector<float> Fo((size_t)iters * d.I * K), So((size_t)iters * K * L), Go((size_t)iters * d.J * L);
  std::vector<double> tau(iters), perf((size_t)iters * 3);
  OK(bnmtf_gibbs_run(h, iters, BNMTF_UPDATE_DRAW, Fo.data(), So.data(), Go.data(), tau.data(), perf.data(), nullptr));
  double t;
  OK(bnmtf_get_state(h, F.data(), S.data(), G.data(), &t));
  OK(bnmtf_destroy(h));
}

// the one-launch path's batch entry points: models of different sizes, a duplicate handle (must be refused), a wrong-kind handle
static void batches() {
  std::vector<Data> ds;
  std::vector<bnmtf_handle> hs;
  for (int i = 0; i < 5; ++i) {
    ds.push_back(make_data(60 + 17 * i, 50 + 11 * i, 0.1, 100 + i));
  }
  for (int i = 0; i < 5; ++i) {
    hs.push_back(create(ds[i], 4 + i, 0, 0, 1, nullptr, 11 + i));
    std::vector<double> U((size_t)ds[i].I * (4 + i), 1.0), V((size_t)ds[i].J * (4 + i), 1.0);
    OK(bnmf_set_state(hs.back(), U.data(), V.data(), 1.0));
  }
  OK(bnmf_gibbs_run_many(hs.data(), (int)hs.size(), 6, BNMTF_UPDATE_DRAW, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
  std::vector<bnmtf_handle> dup = {hs[0], hs[1], hs[0]};
  (void)bnmf_gibbs_run_many(dup.data(), 3, 2, BNMTF_UPDATE_DRAW, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);   // refused when both take the one-launch path
  Data dt = make_data(70, 60, 0.1, 5);
  bnmtf_handle tri = create(dt, 4, 3, 0, 1, nullptr);
  std::vector<double> F(70 * 4, 1.0), S(12, 1.0), G(60 * 3, 1.0);
  OK(bnmtf_set_state(tri, F.data(), S.data(), G.data(), 1.0));
  std::vector<bnmtf_handle> mixed = {hs[0], tri};
  EXPECT_ERR(bnmf_gibbs_run_many(mixed.data(), 2, 2, BNMTF_UPDATE_DRAW, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
  EXPECT_ERR(bnmtf_gibbs_run_many(mixed.data(), 2, 2, BNMTF_UPDATE_DRAW, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
  std::vector<bnmtf_handle> tris = {tri};
  OK(bnmtf_gibbs_run_many(tris.data(), 1, 3, BNMTF_UPDATE_DRAW, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
  OK(bnmtf_destroy(tri));
  for (auto h : hs) OK(bnmtf_destroy(h));
  // pooled arenas: create / destroy in a loop, as a model search does
  for (int i = 0; i < 6; ++i) {
    bnmtf_handle h = create(ds[i % 5], 5, 0, 0, 1, nullptr);
    OK(bnmtf_destroy(h));
  }
}

// a factorisation wider than 64 columns as two column blocks (bnmtf_amd/_blocked.py's call sequence)
static void column_blocks(const Data& d) {
  bnmtf_handle a = create(d, 64, 0, 0, 1, nullptr), b = create(d, 17, 0, 0, 1, nullptr);
  OK(bnmf_set_column_block(a, 0)); OK(bnmf_set_column_block(b, 64));
  std::vector<double> Ua((size_t)d.I * 64, 1.0), Va((size_t)d.J * 64, 1.0), Ub((size_t)d.I * 17, 1.0), Vb((size_t)d.J * 17, 1.0);
  OK(bnmf_set_state(a, Ua.data(), Va.data(), 1.0)); OK(bnmf_set_state(b, Ub.data(), Vb.data(), 1.0));
  for (int it = 0; it < 2; ++it)
    for (int which = 0; which < 2; ++which) {
      OK(bnmf_set_residual_data(a, &b, 1)); OK(bnmf_half_sweep(a, which, BNMTF_UPDATE_DRAW));
      OK(bnmf_set_residual_data(b, &a, 1)); OK(bnmf_half_sweep(b, which, BNMTF_UPDATE_MODE));
    }
  OK(bnmtf_set_tau(a, 0.5)); OK(bnmtf_set_iteration(a, 3));
  EXPECT_ERR(bnmf_set_residual_data(a, &a, 1));
  OK(bnmf_set_residual_data(a, nullptr, 0));
  std::vector<double> num(d.I), tp(d.I);
  OK(bnmf_cond_params(b, 0, 16, num.data(), tp.data()));
  OK(bnmf_vb_set_state(a, Ua.data(), Ua.data(), Ua.data(), Ua.data(), Va.data(), Va.data(), Va.data(), Va.data(), 1.0));
  OK(bnmf_vb_set_state(b, Ub.data(), Ub.data(), Ub.data(), Ub.data(), Vb.data(), Vb.data(), Vb.data(), Vb.data(), 1.0));
  OK(bnmf_set_residual_data(b, &a, 1)); OK(bnmf_vb_half_sweep(b, 0)); OK(bnmf_vb_half_sweep(b, 1));
  double t2[2];
  OK(bnmf_vb_esd_terms(b, t2));
  OK(bnmtf_destroy(a)); OK(bnmtf_destroy(b));
}

// a block of the S of a wider tri-factorisation: a BNMTF handle on the data minus what two carriers (BNMF handles) explain
static void tri_blocks(const Data& d) {
  bnmtf_handle c0 = create(d, 40, 0, 0, 1, nullptr), c1 = create(d, 9, 0, 0, 1, nullptr), t = create(d, 33, 40, 0, 1, nullptr);
  OK(bnmf_set_column_block(c0, 0)); OK(bnmf_set_column_block(c1, 64));
  OK(bnmtf_set_s_block(t, 64, 0, 49));
  EXPECT_ERR(bnmtf_set_s_block(t, 0, 20, 49));                 // (the block would stick out of the wide S)
  EXPECT_ERR(bnmtf_set_s_block(c0, 0, 0, 40));                 // (a BNMF handle)
  std::vector<double> U0((size_t)d.I * 40, 0.5), V0((size_t)d.J * 40, 0.5), U1((size_t)d.I * 9, 0.5), V1((size_t)d.J * 9, 0.5);
  OK(bnmf_set_state(c0, U0.data(), V0.data(), 1.0)); OK(bnmf_set_state(c1, U1.data(), V1.data(), 1.0));
  std::vector<double> F((size_t)d.I * 33, 0.3), S((size_t)33 * 40, 0.3), G((size_t)d.J * 40, 0.3);
  OK(bnmtf_set_state(t, F.data(), S.data(), G.data(), 1.0));
  bnmtf_handle cs[2] = {c0, c1};
  OK(bnmf_set_residual_data(t, cs, 2));
  OK(bnmtf_set_iteration(t, 2));
  OK(bnmtf_s_rows(t, 0, 33, BNMTF_UPDATE_DRAW));
  OK(bnmtf_s_rows(t, 5, 6, BNMTF_UPDATE_ICM));
  EXPECT_ERR(bnmtf_s_rows(t, 3, 40, BNMTF_UPDATE_DRAW));
  EXPECT_ERR(bnmtf_s_rows(c0, 0, 1, BNMTF_UPDATE_DRAW));
  OK(bnmtf_get_state(t, nullptr, S.data(), nullptr, nullptr));
  double num, tp;
  OK(bnmtf_cond_params(t, 1, 32, 39, &num, &tp));
  OK(bnmtf_destroy(c0)); OK(bnmtf_destroy(c1)); OK(bnmtf_destroy(t));
}

// `world` ranks of this process, one thread each, joined by the in-process transport (communicator id "BNMTFLOC...")
static void sharded(const Data& d, int K, int L, int world, const char* token, int iters) {
  uint8_t cid[128];
  memset(cid, 0, sizeof cid);
  snprintf(reinterpret_cast<char*>(cid), sizeof cid, "BNMTFLOC%s", token);
  std::vector<std::thread> ts;
  std::vector<int> rc(world, 0);
  for (int r = 0; r < world; ++r)
    ts.emplace_back([&, r] {
      bnmtf_handle h = create(d, K, L, r, world, cid);
      if (L == 0) {
        std::vector<double> U((size_t)d.I * K, 1.0), V((size_t)d.J * K, 1.0);
        OK(bnmf_set_state(h, U.data(), V.data(), 1.0));
        std::vector<float> Uo((size_t)iters * d.I * K), Vo((size_t)iters * d.J * K);
        OK(bnmf_gibbs_run(h, iters, BNMTF_UPDATE_DRAW, Uo.data(), Vo.data(), nullptr, nullptr, nullptr));
        OK(bnmf_vb_set_state(h, U.data(), U.data(), U.data(), U.data(), V.data(), V.data(), V.data(), V.data(), 1.0));
        OK(bnmf_vb_run(h, 2, nullptr, nullptr, nullptr, nullptr));
      } else {
        std::vector<double> F((size_t)d.I * K, 1.0), S((size_t)K * L, 1.0), G((size_t)d.J * L, 1.0);
        OK(bnmtf_set_state(h, F.data(), S.data(), G.data(), 1.0));
        OK(bnmtf_gibbs_run(h, iters, BNMTF_UPDATE_MODE, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
        // ... and the variational tri-factorisation over the same ranks (round 6)
        OK(bnmtf_vb_set_state(h, F.data(), F.data(), F.data(), F.data(), S.data(), S.data(), S.data(), S.data(), G.data(), G.data(), G.data(), G.data(), 1.0));
        const int per = K * L + K + L;
        std::vector<int32_t> orders((size_t)2 * per);
        for (int it = 0; it < 2; ++it) {
          int32_t* o = &orders[(size_t)it * per];
          for (int a = 0; a < K * L; ++a) o[a] = (a + it) % (K * L);
          for (int k = 0; k < K; ++k) o[K * L + k] = K - 1 - k;
          for (int l = 0; l < L; ++l) o[K * L + K + l] = l;
        }
        OK(bnmtf_vb_run(h, 2, orders.data(), nullptr, nullptr, nullptr, nullptr));
        EXPECT_ERR(bnmtf_vb_update(h, 0, 0, 0, 1));          // (single updates are single-GPU hooks)
      }
      OK(bnmtf_destroy(h));
      rc[r] = 1;
  exception_stays_inside(make_data(515, 389, 0.12, 27), 24);
  vb_many(make_data(300, 120, 0.15, 21), make_data(210, 150, 0.1, 22), 12);
  {   // two host threads, a list of models each (the recorder is thread-local)
    const Data da = make_data(280, 110, 0.15, 23), db = make_data(190, 160, 0.1, 24), dc = make_data(260, 100, 0.2, 25), dd = make_data(150, 170, 0.1, 26);
    std::thread t1([&] { vb_many(da, db, 10); });
    std::thread t2([&] { vb_many(dc, dd, 14); });
    t1.join(); t2.join();
  }
    });
  for (auto& t : ts) t.join();
  for (int r = 0; r < world; ++r) if (!rc[r]) { fprintf(stderr, "rank %d did not finish\n", r); exit(2); }
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  int n = 0;
  OK(bnmtf_device_count(&n));
  int64_t first, count;
  for (int w = 1; w <= 8; ++w) for (int r = 0; r < w; ++r) OK(bnmtf_shard_range(1000 + w, r, w, &first, &count));
  EXPECT_ERR(bnmtf_shard_range(10, 3, 2, &first, &count));

  // small models (the one-launch path's arena), then shapes that take the multi-launch structures: the 8-wave and -- with
  // enough units -- the 16-wave slot layout, the two-chunk inner extent, a mask with rows the on-chip kernels refuse
  bnmf_round_trip(make_data(100, 80, 0.1, 1), 10, 6, true);
  bnmf_round_trip(make_data(640, 512, 0.12, 2), 24, 9, true);            // sample ring: more iterations than its depth
  bnmf_round_trip(make_data(1, 40, 0.0, 3), 3, 3, true);
  bnmf_round_trip(make_data(40, 1, 0.0, 4), 1, 3, false);
  bnmf_round_trip(make_data(300, 200, 0.9, 5), 8, 3, true);              // 90 % missing
  vb_round_trip(make_data(515, 389, 0.12, 6), 40);
  tri_round_trip(make_data(100, 80, 0.1, 7), 5, 5, 4);
  tri_round_trip(make_data(400, 300, 0.1, 8), 32, 17, 3);
  tri_vb_round_trip(make_data(90, 70, 0.1, 15), 4, 5, 3);
  tri_vb_round_trip(make_data(1200, 1100, 0.1, 16), 12, 9, 2);         // the on-chip F / G sweeps' host side, the blocked chain's (K L >= 64)
  batches();
  column_blocks(make_data(150, 120, 0.15, 21));
  column_blocks(make_data(70, 60, 0.1, 22));             // (blocks that qualify for the one-launch arena)
  tri_blocks(make_data(130, 110, 0.12, 23));
  tri_blocks(make_data(60, 50, 0.1, 24));
  sharded(make_data(640, 512, 0.12, 9), 24, 0, 2, "a2", 5);
  sharded(make_data(515, 389, 0.12, 10), 40, 0, 3, "a3", 5);
  sharded(make_data(300, 260, 0.1, 11), 8, 6, 2, "t2", 3);
  if (!quick) {
    bnmf_round_trip(make_data(6200, 6144, 0.1, 12), 64, 3, false);       // >= 192 blocks of 32 units: the 16-wave layout + hand-over tables
    bnmf_round_trip(make_data(700, 12288, 0.1, 13), 16, 2, false);       // inner extent of two LDS panels for the rows direction
    sharded(make_data(2048, 2048, 0.1, 14), 64, 0, 8, "a8", 2);
  }
  printf("sanitize driver: ok (%ld stub launches, %ld live stub allocations)\n", hipstub_launches(), hipstub_live_allocs());
  return 0;
}
