// C ABI of libbnmtf_hip.so (include/bnmtf_hip.h): model construction (host-side
// layout of the masked matrix for the two sweep directions), state hand-off, the
// Gibbs / VB drivers that enqueue the kernels on the handle's stream.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <exception>
#include <new>
#include <system_error>
#include <thread>
#include <tuple>
#include <vector>

#include "model.h"
#include "comm.h"
#include "many.h"
#include "device_rng.h"

namespace bnmtf {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
}  // namespace bnmtf
// No C++ exception leaves the library: every entry point that returns a status is a function-try-block ending in this handler
// (round 6: a std::system_error from a thread that could not be had crossed the ABI once -- std::terminate, the caller's process
// gone; HISTORY 8.4).  Out of host memory -> BNMTF_ENOMEM, anything else -> BNMTF_EINVAL, the message in bnmtf_last_error().
#define BNMTF_ABI_GUARD                                                                                                              \
  catch (const std::bad_alloc&) { bnmtf::set_error("out of host memory"); return BNMTF_ENOMEM; }                                     \
  catch (const std::exception& e) { bnmtf::set_error("host exception: %s", e.what()); return BNMTF_EINVAL; }                         \
  catch (...) { bnmtf::set_error("host exception"); return BNMTF_EINVAL; }
extern "C" int bnmtf_shard_range(int64_t n, int rank, int world, int64_t* first, int64_t* count);
namespace bnmtf {

template <typename T>
static int dalloc(T** p, size_t count, bool zero = true) {
  if (count == 0) count = 1;
  HIPCHK(hipMalloc((void**)p, count * sizeof(T)));
  if (zero) { HIPCHK(hipMemset(*p, 0, count * sizeof(T))); HIPCHK(hipDeviceSynchronize()); }
  return BNMTF_OK;
}
template <typename T>
static void dfree(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

// A model search builds and destroys a model per candidate (fold x rank x restart), all of the same few sizes: the one device
// allocation of a small model and its stream are handed back to a per-process pool instead of to the driver (hipMalloc / hipFree /
// stream creation are 0.3-1.5 ms each, a 1 000-iteration run of a toy model 15 ms).  Bounded: 256 MB of buffers, 64 streams.
struct DevicePool {
  std::mutex mu;
  std::vector<std::tuple<int, size_t, void*>> bufs;     // (device, bytes, pointer)
  std::vector<std::pair<int, hipStream_t>> streams;
  size_t held = 0;
  void* take(int dev, size_t bytes) {
    std::lock_guard<std::mutex> g(mu);
    for (size_t i = 0; i < bufs.size(); ++i)
      if (std::get<0>(bufs[i]) == dev && std::get<1>(bufs[i]) == bytes) { void* p = std::get<2>(bufs[i]); held -= bytes; bufs.erase(bufs.begin() + i); return p; }
    return nullptr;
  }
  bool give(int dev, size_t bytes, void* p) {
    std::lock_guard<std::mutex> g(mu);
    if (held + bytes > ((size_t)256 << 20) || bufs.size() >= 512) return false;
    bufs.emplace_back(dev, bytes, p); held += bytes;
    return true;
  }
  hipStream_t take_stream(int dev) {
    std::lock_guard<std::mutex> g(mu);
    for (size_t i = 0; i < streams.size(); ++i)
      if (streams[i].first == dev) { hipStream_t s = streams[i].second; streams.erase(streams.begin() + i); return s; }
    return nullptr;
  }
  bool give_stream(int dev, hipStream_t s) {
    std::lock_guard<std::mutex> g(mu);
    if (streams.size() >= 64) return false;
    streams.emplace_back(dev, s);
    return true;
  }
};
static DevicePool& device_pool() { static DevicePool* p = new DevicePool(); return *p; }      // (never destroyed: the driver may be gone at exit)

// events / scratch buffers of one call: released on every return path
struct EventList {
  std::vector<hipEvent_t> ev;
  int create(size_t n) {
    ev.reserve(n);
    for (size_t i = 0; i < n; ++i) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); ev.push_back(e); }
    return BNMTF_OK;
  }
  hipEvent_t operator[](size_t i) const { return ev[i]; }
  ~EventList() { for (auto e : ev) (void)hipEventDestroy(e); }
};
template <typename T>
struct DevBuf {
  T* p = nullptr;
  int alloc(size_t count, bool zero = false) { return dalloc(&p, count, zero); }
  ~DevBuf() { dfree(p); }
};

// Host threads for the O(I*J) layout passes of bnmtf_create (a cross-validation driver pays them once per model):
// fn(begin, end) over fixed-size chunks of [0, n), so results never depend on the thread count.
template <typename Fn>
static void parallel_chunks(int n, int chunk, Fn fn) {
  const int nchunks = (n + chunk - 1) / chunk;
  int nt = (int)std::thread::hardware_concurrency();
  if (const char* e = getenv("BNMTF_HOST_THREADS")) nt = atoi(e);
  nt = std::max(1, std::min({nt, 32, nchunks}));
  if ((size_t)n * (size_t)chunk < 4096) nt = 1;           // (a few thousand rows: starting the threads costs more than the pass)
  std::atomic<int> next{0};
  auto work = [&]() {
    for (int c = next.fetch_add(1); c < nchunks; c = next.fetch_add(1)) fn(c * chunk, std::min(n, (c + 1) * chunk));
  };
  if (nt == 1) { work(); return; }
  // nt - 1 helpers and the calling thread.  A thread that cannot be had (EAGAIN: the user's process / thread limit -- a long test
  // session with worker pools, three ranks building their layouts at once) must not leave this function as an exception: it
  // would cross the C ABI and end the process (std::terminate).  The chunks are claimed one by one, so whoever is there does them.
  std::vector<std::thread> ts;
  ts.reserve(nt);
  for (int t = 0; t + 1 < nt; ++t) {
    try { ts.emplace_back(work); } catch (const std::system_error&) { break; }
  }
  work();
  for (auto& t : ts) t.join();
}

// BNMTF_CREATE_TIMING=1: wall-clock laps of bnmtf_create's phases on stderr
struct CreateLaps {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  bool on = getenv("BNMTF_CREATE_TIMING") != nullptr;
  void lap(const char* what) {
    const auto now = std::chrono::steady_clock::now();
    if (on) fprintf(stderr, "bnmtf_create: %-44s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};
// ------------------------------------------------------------------ layout
// Fill one direction.  get(u, r) returns (observed, value) of unit u (global) at
// inner index r.
// dR / dM: the data and the mask on the device ([I][J] row major); by_rows: a unit is a row of R (else a column); obs: observed
// entries per unit (global).  The I x J passes -- masked / transposed contraction operand, missing lists -- run on the device
// (kernel_layout.hip); the slot layout below is built on the host from the downloaded lists.
static int build_dir(Dir& d, int nglob, int m, int W, int rank, int world, const double* lambda,
                     const float* dR, const uint8_t* dM, int I, int J, bool by_rows, const std::vector<uint32_t>& obs, hipStream_t st) {
  d.nglob = nglob; d.m = m; d.W = W; d.KP = W <= 32 ? 32 : 64;
  {
    int64_t first = 0, count = 0;
    (void)bnmtf_shard_range(nglob, rank, world, &first, &count);
    d.n0 = (int)first; d.n = (int)count;
  }
  d.n_pad = round_up(std::max(d.n, 1), 128);
  // split of the inner dimension: aim at >= 2 blocks per CU, >= 64 inner rows per wave
  // a wave covers 128 output columns (four 32-column tiles); when the output side is short (a shard of a multi-GPU run)
  // 64 columns, so that the chip is filled with half as many inner slices, each twice as long (BNMTF_GEMM_TW=2/4 forces)
  d.gemm_tw = (d.KP == 64 && d.n_pad <= 2048) ? 2 : 4;
#ifdef BNMTF_EXPERIMENTS       // (A/B switches of decisions that are made: make EXPERIMENTS=1)
  if (const char* e = getenv("BNMTF_GEMM_TW")) d.gemm_tw = atoi(e) == 2 && d.KP == 64 ? 2 : 4;
#endif
  const int tiles = d.n_pad / (32 * d.gemm_tw);
  // one block per CU: at KP = 64 the GEMM holds 332 registers per lane (one resident block), and at KP = 32 -- where two
  // would fit -- 256 blocks with twice the inner slice per wave beat 512 (4096^2, K = 32: 20.3 us against 23.2; round 3)
  int split = std::max(1, 256 / tiles);
#ifdef BNMTF_EXPERIMENTS
  if (const char* e = getenv("BNMTF_GEMM_SPLIT")) split = std::max(1, atoi(e));
#endif
  const int max_split = std::max(1, m / (4 * 64));
  d.split = std::min(split, max_split);
  d.ipw = round_up((m + d.split * 4 - 1) / (d.split * 4), 32);   // multiple of the GEMM's 2*U register-pipeline group
  d.inner_pad = d.split * 4 * d.ipw;

  CreateLaps laps;
  std::vector<uint32_t> ptr(d.n + 1, 0), idx;
  d.obs_count.assign(nglob, 0);
  d.nmiss = 0;
  for (int ul = 0; ul < d.n; ++ul) {
    const uint32_t cnt = (uint32_t)m - obs[d.n0 + ul];
    d.nmiss += cnt;
    ptr[ul + 1] = ptr[ul] + (cnt + 63u) / 64u * 64u;                 // 64-wide slots, padded with the zero sentinel m
  }
  d.nslots = ptr[d.n];
  CHK(dalloc(&d.big, (size_t)d.inner_pad * d.n_pad));               // zero-filled: the pads stay zero
  CHK(dalloc(&d.slot_ptr, ptr.size(), false));
  HIPCHK(hipMemcpyAsync(d.slot_ptr, ptr.data(), ptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  CHK(dalloc(&d.idx, std::max<size_t>(d.nslots, 1), false));
  launch_masked_operand(dR, dM, I, J, by_rows ? 1 : 0, d.n0, d.n, m, d.big, d.n_pad, st);
  launch_missing_lists(dM, I, J, by_rows ? 1 : 0, d.n0, d.n, m, d.slot_ptr, d.idx, st);
  idx.resize(d.nslots);
  if (d.nslots) HIPCHK(hipMemcpyAsync(idx.data(), d.idx, d.nslots * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  laps.lap("  build_dir: R~ operand + missing lists (device)");
  auto miss_begin = [&](int ul) { return idx.data() + ptr[ul]; };
  auto miss_end = [&](int ul) { return idx.data() + ptr[ul] + ((uint32_t)m - obs[d.n0 + ul]); };

  // fast layout: a unit owns a 32-lane half wave.  Lane r prefers the entries with j mod 32 == r (bank-conflict-free
  // LDS gathers).  Residue classes are binomially unbalanced, so instead of padding every lane to the fullest class
  // a unit gets E = ceil(cnt / 32) slots per lane (rounded up to even) and the entries of over-full classes are
  // parked in lanes with room.  A parked entry shares its row's LDS read with the entry of its own residue lane (a
  // 2-way bank conflict: one extra LDS cycle for that row); parked entries are packed into the last rows, distinct
  // residues per row, so that few rows pay it.  BNMTF_BALANCE=0 restores the padded conflict-free layout.
  d.mz = round_up(m, 32);
  d.pw = round_up(d.mz + 32, 256);
  // an inner extent that does not fit one LDS panel is cut in two chunks (kernel_sweep_fast.hip: sweep_two_chunks_plan); a
  // unit's entries are then laid out chunk by chunk, with inner indices local to the chunk
  d.nch = 1; d.mh = 0; d.pw_chunk = d.pw; d.pw1 = 0;
  if (!sweep_fast_supported(d.KP, d.pw) && !getenv("BNMTF_NO_CHUNKS") && sweep_two_chunks_plan(d.KP, m, &d.mh, &d.pw_chunk, &d.pw1)) d.nch = 2;
  const int nch = d.nch;
  {
#ifdef BNMTF_EXPERIMENTS
    const bool balance = !(getenv("BNMTF_BALANCE") && atoi(getenv("BNMTF_BALANCE")) == 0);
#else
    constexpr bool balance = true;
#endif
    const uint32_t kNone = 0xFFFFFFFFu;
    std::vector<int> Eu(d.n, 0);
    std::vector<std::vector<uint32_t>> lanes((size_t)d.n * 32 * nch);     // per unit, per chunk, per lane: slot contents (kNone = empty), chunk-local inner indices
    // E slots per lane for the 32 lane lists of one half wave (L[r]: the entries of residue class r, `cnt` in all), balanced:
    // E = ceil(cnt / 32) rounded up to even, the entries of over-full classes parked in lanes with room (see above)
    auto balance_lanes = [&](std::vector<uint32_t>* L, size_t cnt, std::vector<std::vector<uint32_t>>& over) -> int {
      int emax = 0;
      for (int r = 0; r < 32; ++r) emax = std::max(emax, (int)L[r].size());
      int E = std::max(2, (emax + 1) & ~1);
      const int Eb = std::max(2, ((int)((cnt + 31) / 32) + 1) & ~1);
      if (balance && Eb < E) {
        E = Eb;
        size_t nover = 0;
        for (int r = 0; r < 32; ++r) {
          over[r].clear();
          while ((int)L[r].size() > E) { over[r].push_back(L[r].back()); L[r].pop_back(); ++nover; }
        }
        std::vector<int> own(32);
        for (int r = 0; r < 32; ++r) { own[r] = (int)L[r].size(); L[r].resize(E, kNone); }
        // last rows first; in a row every free lane takes a parked entry of a residue not yet parked in that row
        int rr = 0;
        for (int row = E - 1; row >= 0 && nover > 0; --row) {
          uint32_t used = 0;
          for (int lane = 0; lane < 32 && nover > 0; ++lane) {
            if (own[lane] > row) continue;
            int pick = -1;
            for (int t = 0; t < 32; ++t) { const int r = (rr + t) & 31; if (!over[r].empty() && !((used >> r) & 1u)) { pick = r; break; } }
            if (pick < 0) break;
            L[lane][row] = over[pick].back(); over[pick].pop_back(); --nover;
            used |= 1u << pick; rr = (pick + 1) & 31;
          }
        }
        for (int row = E - 1; row >= 0 && nover > 0; --row)      // leftovers (same residue twice in a row): any free slot
          for (int lane = 0; lane < 32 && nover > 0; ++lane) {
            if (L[lane][row] != kNone) continue;
            for (int r = 0; r < 32; ++r) if (!over[r].empty()) { L[lane][row] = over[r].back(); over[r].pop_back(); --nover; break; }
          }
      }
      return E;
    };
    parallel_chunks(d.n, 64, [&](int ua, int ub) {
    std::vector<std::vector<uint32_t>> over(32);
    for (int ul = ua; ul < ub; ++ul) {
      int ehalf = 0;
      for (int ch = 0; ch < nch; ++ch) {
      std::vector<uint32_t>* L = &lanes[((size_t)ul * nch + ch) * 32];
      const uint32_t lo = ch == 0 ? 0u : (uint32_t)d.mh, hi = (nch == 2 && ch == 0) ? (uint32_t)d.mh : 0xFFFFFFFFu;
      size_t cnt = 0;
      {   // one allocation per lane instead of a doubling chain (32 lanes x 8192 units: the layout pass was mostly malloc)
        const size_t guess = (size_t)(miss_end(ul) - miss_begin(ul)) / (32 * (size_t)nch) + 8;
        for (int r = 0; r < 32; ++r) L[r].reserve(guess);
      }
      for (const uint32_t* pj = miss_begin(ul); pj != miss_end(ul); ++pj) { const uint32_t j = *pj; if (j >= lo && j < hi) { L[(j - lo) & 31].push_back(j - lo); ++cnt; } }
      const int E = balance_lanes(L, cnt, over);
      ehalf = std::max(ehalf, E);
      }
      Eu[ul] = nch * ehalf;                  // two chunks: each gets half of the unit's slot rows (a multiple of 4 in all)
    }
    });
    laps.lap("    layout: lanes filled and balanced");
    auto res = [&](int ul, int ch, int r) -> const std::vector<uint32_t>& { return lanes[((size_t)ul * nch + ch) * 32 + r]; };
    std::vector<int> order(d.n);
    for (int i = 0; i < d.n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return Eu[x] > Eu[y]; });
    // Units pair up in descending slot-count order.  The 2/4/8-wave blocks take the pairs in that order.  The on-chip
    // kernel picks the slot class per WAVE, so for the 16-wave shape the pairs are dealt to the
    // blocks boustrophedon (and to the four SIMDs of a block likewise): every block, and every SIMD, gets the same mix
    // of full and light units, and one round of blocks ends together.
    const int npairs_real = (d.n + 1) / 2;
    const int emax_all = d.n > 0 ? Eu[order[0]] : 0;
    const int wide_blocks = (npairs_real + 15) / 16;
    const bool wide_can = nch == 1 && sweep_wide_supported(d.KP, d.pw) && emax_all <= kWideMaxSlots && d.n > 0;
    d.wide_can = wide_can;
    d.use_wide = wide_can && wide_blocks >= 192;
    if (const char* e = getenv("BNMTF_WIDE")) d.use_wide = wide_can && atoi(e) != 0;      // 0: never, 1: whenever it can run
    if (const char* e = getenv("BNMTF_VB_PATH")) d.vb_path = !strcmp(e, "masked") ? 1 : !strcmp(e, "pairs") ? 2 : 0;
    d.use_turns = false;
#ifdef BNMTF_EXPERIMENTS
    if (const char* e = getenv("BNMTF_TURNS")) d.use_turns = d.use_wide && sweep_turns_supported(d.KP, d.pw) && atoi(e) != 0;
#endif
    d.f_npairs = d.use_wide ? wide_blocks * 16 : npairs_real;
    auto slot_of = [&](int pi) {
      if (!d.use_wide) return pi;
      const int r = pi / wide_blocks, c = pi % wide_blocks;
      const int blk = (r & 1) ? wide_blocks - 1 - c : c;
      // rank 0 = the pairs with the most slots.  The lightest quarter goes to waves 0-3, the waves of the sampler window (sampler
      // and table filler: sweep_chip.inc, RL) -- their role needs registers the heaviest slot class does not have, and a light
      // wave reaches the column's first barrier early, with the word-only half of its candidate done by the time the others arrive
      const int t = r >> 2, sx = r & 3;
      return blk * 16 + 4 * (3 - t) + ((t & 1) ? 3 - sx : sx);
    };
    std::vector<int> umap((size_t)d.f_npairs * 2, -1);
    std::vector<uint32_t> pE(d.f_npairs, 0u), pB(d.f_npairs, 0u);
    d.f_emax = 0;
    for (int pi = 0; pi < npairs_real; ++pi) {
      int e = 0;
      const int sl = slot_of(pi);
      for (int hh = 0; hh < 2; ++hh) {
        const int pos = 2 * pi + hh;
        if (pos < d.n) { umap[2 * sl + hh] = order[pos]; e = std::max(e, Eu[order[pos]]); }
      }
      pE[sl] = (uint32_t)e;
      d.f_emax = std::max(d.f_emax, e);
    }
    size_t rows_total = 0;
    for (int sl = 0; sl < d.f_npairs; ++sl) { pB[sl] = (uint32_t)rows_total; rows_total += pE[sl]; }
    std::vector<uint32_t> off(std::max<size_t>(rows_total, 1) * 64);
    parallel_chunks(d.f_npairs, 64, [&](int pa, int pb) {
    for (int pi = pa; pi < pb; ++pi)
      for (int hh = 0; hh < 2; ++hh) {
        const int ul = umap[2 * pi + hh];
        for (uint32_t sidx = 0; sidx < pE[pi]; ++sidx) {
          // two chunks: the pair's first pE / 2 rows are chunk 0 (zero words behind index mh), the rest chunk 1 (behind mz - mh)
          const int ch = (nch == 2 && sidx >= pE[pi] / 2) ? 1 : 0;
          const uint32_t sl = sidx - (uint32_t)ch * (pE[pi] / 2);
          const uint32_t sent0 = nch == 2 ? (uint32_t)(ch ? d.mz - d.mh : d.mh) : (uint32_t)d.mz;
          for (int r = 0; r < 32; ++r) {
            uint32_t v = sent0 + (uint32_t)r;
            if (ul >= 0) { const auto& lst = res(ul, ch, r); if (sl < lst.size() && lst[sl] != kNone) v = lst[sl]; }
            off[((size_t)pB[pi] + sidx) * 64 + hh * 32 + r] = v;
          }
        }
      }
    });
    d.f_slots = rows_total;
    laps.lap("    layout: slot table filled");
    CHK(dalloc(&d.f_unit_map, umap.size(), false));
    HIPCHK(hipMemcpy(d.f_unit_map, umap.data(), umap.size() * sizeof(int), hipMemcpyHostToDevice));
    CHK(dalloc(&d.f_pair_E, pE.size(), false));
    HIPCHK(hipMemcpy(d.f_pair_E, pE.data(), pE.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    CHK(dalloc(&d.f_pair_base, pB.size(), false));
    HIPCHK(hipMemcpy(d.f_pair_base, pB.data(), pB.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    CHK(dalloc(&d.f_off, off.size(), false));
    HIPCHK(hipMemcpy(d.f_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    {   // 16-bit packed slot pairs (two inner indices per word): the slot table the on-chip kernels load
      d.pair_ok = d.mz + 32 < 65536;
      std::vector<uint32_t> off16(std::max<size_t>(rows_total / 2, 1) * 64, 0);
      if (d.pair_ok)
        parallel_chunks((int)(rows_total / 2), 4096, [&](int ra, int rb) {
          for (size_t r2 = (size_t)ra; r2 < (size_t)rb; ++r2)
            for (int l = 0; l < 64; ++l) off16[r2 * 64 + l] = (off[(2 * r2) * 64 + l] & 0xFFFFu) | (off[(2 * r2 + 1) * 64 + l] << 16);
        });
      CHK(dalloc(&d.f_off16, off16.size(), false));
      HIPCHK(hipMemcpy(d.f_off16, off16.data(), off16.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    // pairs with more slots per lane than the block shape holds (kFastMaxSlots; the 16-wave shape is only chosen when
    // every pair fits) are left to the generic kernel: their waves idle in the on-chip kernel
    {
      std::vector<int> gen;
      // waves per block: 8 when there are enough units for >= 256 blocks, else 4 or 2 (multi-GPU shards, small problems)
      d.f_nw = d.f_npairs >= 8 * 256 ? 8 : (d.f_npairs >= 4 * 256 ? 4 : (d.f_npairs >= 2 * 64 ? 2 : 8));
      if (const char* e = getenv("BNMTF_FAST_NW")) d.f_nw = atoi(e) == 2 ? 2 : (atoi(e) == 4 ? 4 : 8);
      if (d.use_wide) d.f_nw = 16;
      // the twin shape (BNMTF_TWIN=1): the 16-wave layout run by 8-wave blocks, two to a CU (sweep_chip.inc, TW = 1)
      d.use_twin = false;
#ifdef BNMTF_EXPERIMENTS
      if (const char* e = getenv("BNMTF_TWIN")) d.use_twin = d.use_wide && !d.use_turns && world == 1 && d.pw <= kTwinPanelStride && atoi(e) != 0;
#endif
      if (d.use_twin) d.f_nw = 8;
      if (nch == 2) d.f_nw = 8;                       // the two-chunk variant is an 8-wave kernel
      for (int pi = 0; pi < d.f_npairs && !d.use_wide; ++pi)
        if ((int)pE[pi] > kFastMaxSlots)
          for (int t = 2 * pi; t < 2 * pi + 2; ++t) if (umap[t] >= 0) gen.push_back(umap[t]);
      d.fast_ok = true;
      d.f_gen_count = (int)gen.size();
      CHK(dalloc(&d.f_gen_units, std::max<size_t>(gen.size(), 1), false));
      if (!gen.empty()) HIPCHK(hipMemcpy(d.f_gen_units, gen.data(), gen.size() * sizeof(int), hipMemcpyHostToDevice));
      d.stats_blocks = std::max((d.f_npairs + d.f_nw - 1) / d.f_nw, sweep_vb_blocks(d.f_npairs)) + 2;   // the VB sweep writes its own block count of rows
      // the block of every slot row: what build_handover needs beside the layout itself (one GPU, the 16-wave shape or the
      // plain 8-wave shape, every unit on the on-chip kernel)
      if (world == 1 && !d.use_turns && d.pair_ok && d.f_gen_count == 0 && nch == 1 && (d.f_nw == 16 || d.f_nw == 8) && d.f_npairs / d.f_nw < 65535) {
        d.ho_ppb = d.f_nw;
        std::vector<uint16_t> row_blk(std::max<size_t>(rows_total, 1), 0);
        for (int pi = 0; pi < d.f_npairs; ++pi)
          for (uint32_t sidx = 0; sidx < pE[pi]; ++sidx) row_blk[pB[pi] + sidx] = (uint16_t)(pi / d.ho_ppb);
        CHK(dalloc(&d.f_row_blk, row_blk.size(), false));
        HIPCHK(hipMemcpy(d.f_row_blk, row_blk.data(), row_blk.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      }
    }
    // ---- the unit-per-wave layout (kernel_sweep_unit.hip, round 6): few units per CU -- a shard of a multi-GPU run, a small
    // problem.  A pair's two halves belong to the SAME unit: residue class r of the unit's missing entries is dealt in turn to
    // lanes r and r + 32 (each half of a 64-lane LDS read touches 32 distinct banks), each half balanced like a 32-lane unit.
    // Pair p = local unit p (no sorting: a block's waves do not share slot work).  Beside the layout above (the variational
    // sweeps keep it), a few MB at these sizes.
    d.uw_ok = false;
    // (a test that forces another block shape -- BNMTF_WIDE, BNMTF_FAST_NW -- gets that shape; BNMTF_UNIT=0 switches this one off)
    if (d.n > 0 && d.n <= kUnitMaxUnits && nch == 1 && d.pair_ok && sweep_unit_supported(d.KP, d.pw) && !getenv("BNMTF_WIDE") && !getenv("BNMTF_FAST_NW") &&
        !(getenv("BNMTF_UNIT") && atoi(getenv("BNMTF_UNIT")) == 0)) {
      std::vector<std::vector<uint32_t>> ul((size_t)d.n * 64);
      std::vector<uint32_t> uE(d.n, 0u), uB(d.n, 0u);
      parallel_chunks(d.n, 64, [&](int ua, int ub) {
        std::vector<std::vector<uint32_t>> over(32);
        for (int u = ua; u < ub; ++u) {
          std::vector<uint32_t>* L = &ul[(size_t)u * 64];
          size_t cnt[2] = {0, 0};
          uint8_t turn[32] = {};
          for (const uint32_t* pj = miss_begin(u); pj != miss_end(u); ++pj) {
            const int r = (int)(*pj & 31u), hh = turn[r]; turn[r] ^= 1;
            L[hh * 32 + r].push_back(*pj); ++cnt[hh];
          }
          const int e0 = balance_lanes(L, cnt[0], over), e1 = balance_lanes(L + 32, cnt[1], over);
          uE[u] = (uint32_t)std::max(e0, e1);
        }
      });
      size_t rows = 0; uint32_t emax = 0;
      for (int u = 0; u < d.n; ++u) { uB[u] = (uint32_t)rows; rows += uE[u]; emax = std::max(emax, uE[u]); }
      if ((int)emax <= kUnitMaxSlots) {
        std::vector<uint32_t> o16(std::max<size_t>(rows / 2, 1) * 64, 0);
        parallel_chunks(d.n, 64, [&](int ua, int ub) {
          for (int u = ua; u < ub; ++u)
            for (uint32_t sidx = 0; sidx < uE[u]; ++sidx)
              for (int l = 0; l < 64; ++l) {
                const auto& lst = ul[(size_t)u * 64 + l];
                uint32_t v = (uint32_t)d.mz + (uint32_t)(l & 31);
                if (sidx < lst.size() && lst[sidx] != kNone) v = lst[sidx];
                uint32_t& w = o16[((size_t)(uB[u] + sidx) / 2) * 64 + l];
                w = ((uB[u] + sidx) & 1u) ? (w & 0xFFFFu) | (v << 16) : (w & 0xFFFF0000u) | v;
              }
        });
        std::vector<int> um((size_t)d.n * 2);
        for (int u = 0; u < d.n; ++u) um[2 * u] = um[2 * u + 1] = u;
        CHK(dalloc(&d.u_unit_map, um.size(), false));
        HIPCHK(hipMemcpy(d.u_unit_map, um.data(), um.size() * sizeof(int), hipMemcpyHostToDevice));
        CHK(dalloc(&d.u_pair_E, uE.size(), false));
        HIPCHK(hipMemcpy(d.u_pair_E, uE.data(), uE.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        CHK(dalloc(&d.u_pair_base, uB.size(), false));
        HIPCHK(hipMemcpy(d.u_pair_base, uB.data(), uB.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        CHK(dalloc(&d.u_off16, o16.size(), false));
        HIPCHK(hipMemcpy(d.u_off16, o16.data(), o16.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        d.u_nw = d.n <= 4 * 256 ? 4 : 8;
        if (const char* e = getenv("BNMTF_UNIT_NW")) d.u_nw = atoi(e) == 8 ? 8 : 4;
        d.u_emax = (int)emax;
        d.uw_ok = true;
        d.stats_blocks = std::max(d.stats_blocks, (d.n + d.u_nw - 1) / d.u_nw + 2);
      }
      laps.lap("    layout: unit-per-wave tables");
    }
    CHK(dalloc(&d.stats, (size_t)d.stats_blocks * 4));
    laps.lap("    layout: tables uploaded");
  }

  laps.lap("  build_dir: slot layout + its uploads");
  CHK(dalloc(&d.slabs, (size_t)d.split * d.n_pad * d.KP));
  CHK(dalloc(&d.q, idx.size()));
  std::vector<float> lam((size_t)std::max(d.n, 1) * d.KP, 0.0f);
  for (int ul = 0; ul < d.n; ++ul)
    for (int k = 0; k < W; ++k) lam[(size_t)ul * d.KP + k] = (float)lambda[(size_t)(d.n0 + ul) * W + k];
  CHK(dalloc(&d.lambda, lam.size(), false));
  HIPCHK(hipMemcpy(d.lambda, lam.data(), lam.size() * sizeof(float), hipMemcpyHostToDevice));
  // C64 | colsum | colsum2 in one piece: what a multi-GPU run sums over the ranks with ONE all-reduce (exchange_factor)
  CHK(dalloc(&d.C64, (size_t)64 * 64 + 128));
  d.colsum = d.C64 + 64 * 64; d.colsum2 = d.colsum + 64; d.gram_packed = true;
  CHK(dalloc(&d.C32, (size_t)64 * 64));
  CHK(dalloc(&d.numer, (size_t)std::max(d.n, 1)));
  CHK(dalloc(&d.taup, (size_t)std::max(d.n, 1)));
  return BNMTF_OK;
}

// q hand-over tables for one writer / reader pair of directions (model.h Dir::ho_*), built on the device from the two slot
// layouts (kernel_handover.hip).  Returns false (and leaves the pair without hand-over) if a staging area or a region would
// not fit the block's LDS or its 16-bit offsets.
static bool build_handover(bnmtf_model* h, Dir& W, Dir& Rd) {
  if (!W.f_row_blk || !Rd.f_row_blk || W.f_npairs == 0 || Rd.f_npairs == 0) return false;
  const int nbW = (W.f_npairs + W.ho_ppb - 1) / W.ho_ppb, nbR = (Rd.f_npairs + Rd.ho_ppb - 1) / Rd.ho_ppb;
  if (nbR > 8192) return false;                         // (the destination kernel counts per reader block in LDS)
  const size_t rowsW = W.f_slots, rowsR = Rd.f_slots;
  DevBuf<uint32_t> inv, dst_slot, dst_rank, count, sbase, stotal, rbase, rdata, rsize, limits;
  if (inv.alloc(std::max<size_t>(Rd.nslots, 1)) || dst_slot.alloc(std::max<size_t>(rowsW, 1) * 64) || dst_rank.alloc(std::max<size_t>(rowsW, 1) * 64) ||
      count.alloc((size_t)nbW * nbR) || sbase.alloc((size_t)nbW * nbR) || stotal.alloc(nbW) || rbase.alloc((size_t)nbW * nbR) || rdata.alloc(nbR) ||
      rsize.alloc(nbR) || limits.alloc(4, true)) return false;
  uint16_t *t_out = nullptr, *t_in = nullptr;
  // (two 16-bit places per word; an odd row count leaves the last word's upper half unused)
  if (dalloc(&t_out, ((rowsW + 1) / 2) * 128, true) || dalloc(&t_in, ((rowsR + 1) / 2) * 128, true)) return false;
  W.ho_out = reinterpret_cast<uint32_t*>(t_out); Rd.ho_in = reinterpret_cast<uint32_t*>(t_in);
  if (dalloc(&W.ho_pk, (size_t)nbW * nbR * 3, false) || dalloc(&Rd.ho_region_ofs, (size_t)nbR + 1, false)) return false;
  HandoverArgs a;
  a.w_off = W.f_off; a.w_pB = W.f_pair_base; a.w_pE = W.f_pair_E; a.w_umap = W.f_unit_map; a.w_npairs = W.f_npairs; a.w_ppb = W.ho_ppb; a.w_nb = nbW; a.w_inner = (uint32_t)Rd.nglob;
  a.r_off = Rd.f_off; a.r_pB = Rd.f_pair_base; a.r_pE = Rd.f_pair_E; a.r_umap = Rd.f_unit_map; a.r_npairs = Rd.f_npairs; a.r_ppb = Rd.ho_ppb; a.r_nb = nbR; a.r_inner = (uint32_t)W.nglob;
  a.r_row_blk = Rd.f_row_blk; a.r_ptr = Rd.slot_ptr; a.r_idx = Rd.idx;
  a.inv = inv.p; a.dst_slot = dst_slot.p; a.dst_rank = dst_rank.p; a.count = count.p;
  a.sbase = sbase.p; a.stotal = stotal.p; a.rbase = rbase.p; a.rdata = rdata.p; a.rsize = rsize.p;
  a.rofs = Rd.ho_region_ofs; a.t_out = t_out; a.t_in = t_in; a.pk = W.ho_pk; a.limits = limits.p;
  launch_handover_build(a, h->stream);
  uint32_t lim[4];
  if (hipMemcpyAsync(lim, limits.p, sizeof(lim), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return false;
  // staging area and region live behind the Gram in the block's LDS (160 KiB in all; the launch asks for what they need)
  // (8-wave blocks run two to a CU: half the LDS each.  256: the VB kernel's small arrays ahead of the panels)
  const uint32_t lds_w = (W.ho_ppb == 8 ? 80u : 160u) * 1024u / 4u - (uint32_t)(W.KP * W.KP) - 256u;
  const uint32_t lds_r = (Rd.ho_ppb == 8 ? 80u : 160u) * 1024u / 4u - (uint32_t)(W.KP * W.KP) - 256u;
  if (lim[2] != 0 || lim[0] + 256u + 32u > lds_w || lim[0] + 32u > 65536u || lim[1] > lds_r || lim[1] > 65536u) return false;
  W.ho_lds_floats = std::max(W.ho_lds_floats, (int)lim[0] + 256 + 32);
  Rd.ho_lds_floats = std::max(Rd.ho_lds_floats, (int)lim[1]);
  if (dalloc(&Rd.ho_region, (size_t)lim[3] + 256) != BNMTF_OK) return false;      // zero-filled: the zeros behind the runs are never written
  Rd.ho_blocks = nbR;
  return true;
}

static int alloc_factor(Dir& d, int other_inner_pad) {
  d.xrows = std::max(d.nglob + 1, other_inner_pad);
  d.ldT = round_up(round_up(d.nglob, 32) + 32, 256);   // = the other direction's panel size pw; zero beyond nglob
  CHK(dalloc(&d.X, (size_t)d.xrows * d.KP));
  CHK(dalloc(&d.XT, (size_t)d.KP * d.ldT));
  CHK(dalloc(&d.XT2, (size_t)d.KP * d.ldT));
  const int nb = post_blocks(d.nglob);
  CHK(dalloc(&d.Cpart, (size_t)nb * d.KP * d.KP));
  CHK(dalloc(&d.spart, (size_t)nb * d.KP));
  CHK(dalloc(&d.s2part, (size_t)nb * d.KP));
  return BNMTF_OK;
}

static void free_dir(Dir& d) {
  dfree(d.big); dfree(d.slabs); dfree(d.lambda); dfree(d.slot_ptr); dfree(d.idx); dfree(d.q);
  dfree(d.X); dfree(d.XT); dfree(d.C64); dfree(d.C32);
  if (!d.gram_packed) { dfree(d.colsum); dfree(d.colsum2); }
  if (d.ev_sweep) (void)hipEventDestroy(d.ev_sweep);
  if (d.ev_gram) (void)hipEventDestroy(d.ev_gram);
  if (d.ev_gathered) (void)hipEventDestroy(d.ev_gathered);
  if (d.ev_gram_all) (void)hipEventDestroy(d.ev_gram_all);
  dfree(d.XT2); dfree(d.Cpart); dfree(d.spart); dfree(d.s2part); dfree(d.f_unit_map); dfree(d.f_pair_E);
  dfree(d.f_pair_base); dfree(d.f_off); dfree(d.f_off16); dfree(d.stats); dfree(d.f_gen_units);
  dfree(d.u_unit_map); dfree(d.u_pair_E); dfree(d.u_pair_base); dfree(d.u_off16);
  dfree(d.vb_stats);
  dfree(d.ho_in); dfree(d.ho_out); dfree(d.ho_pk); dfree(d.ho_region_ofs); dfree(d.ho_region); dfree(d.f_row_blk);
  dfree(d.mu); dfree(d.tauq); dfree(d.var); dfree(d.S2); dfree(d.S2T); dfree(d.XS); dfree(d.vb_asq); dfree(d.vb_vsq); dfree(d.mbits); dfree(d.XB); dfree(d.mslabs); dfree(d.xb_umax); dfree(d.xb_cexp); dfree(d.xb_mpart); dfree(d.numer); dfree(d.taup);
}

// ---------------------------------------------------------------- profiling
struct ScopedKernelTimer {
  bnmtf_model* h; int id; hipEvent_t a = nullptr, b = nullptr;
  bool on;
  ScopedKernelTimer(bnmtf_model* h_, int id_) : h(h_), id(id_), on(((h_->profiling >> id_) & 1u) && h_->iteration % h_->profile_stride == 0) {
    if (!on) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!h->event_pool.empty()) { e = h->event_pool.back(); h->event_pool.pop_back(); }
      else (void)hipEventCreate(&e);
      return e;
    };
    a = get(); b = get();
    (void)hipEventRecord(a, h->stream);
  }
  ~ScopedKernelTimer() {
    if (!on) return;
    (void)hipEventRecord(b, h->stream);
    h->pending_events.push_back({id, {a, b}});
  }
};
static void drain_events(bnmtf_model* h) {
  for (auto& pe : h->pending_events) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pe.second.first, pe.second.second) == hipSuccess) {
      h->kernel_ms[pe.first] += ms;
      h->kernel_launches[pe.first] += 1;
    }
    h->event_pool.push_back(pe.second.first);
    h->event_pool.push_back(pe.second.second);
  }
  h->pending_events.clear();
}

// q hand-over between the half sweeps, for the duration of one run call: from its first rows sweep (a pre-pass) on
struct HandoverScope {
  bnmtf_model* h;
  // (a call that follows another one of the same handle with the state untouched in between picks the regions up where that
  // call left them: run(50) + run(50) is the chain of run(100), bit for bit)
  explicit HandoverScope(bnmtf_model* h_) : h(h_) {
    h->ho_active = h->ho_enabled && !h->comm && h->use_fast && !h->block_mode;
    if (!h->ho_active || !h->ho_regions_current) h->rows.ho_filled = h->cols.ho_filled = false;
    h->ho_regions_current = false;            // (an error return out of the loop leaves them unknown)
  }
  // the run loop has finished and its stream is drained: the regions hold q of the state the call leaves behind
  void commit() { h->ho_regions_current = h->ho_active && h->rows.ho_filled; }
  ~HandoverScope() { h->ho_active = false; }        // (an error return never got to commit(): the regions stay "unknown")
};
// ------------------------------------------------------------- step pieces
static void enqueue_gemm(bnmtf_model* h, Dir& d, const Dir& other, int kid, int part = 0);
// part: 0 the whole contraction; 1 only the inner slices that lie inside the rank's OWN block of the other factor (rows
// [other.n0, other.n0 + other.n): final the moment the rank's sweep ends, before any exchange); 2 the remaining slices
static void enqueue_gemm(bnmtf_model* h, Dir& d, const Dir& other, int kid, int part) {
  GemmArgs g;
  g.big = d.big; g.ld = d.n_pad; g.X = other.X; g.slabs = d.slabs;
  g.n_pad = d.n_pad; g.split = d.split; g.inner_per_wave = d.ipw; g.tw = d.gemm_tw;
  const int rows_per_slice = 4 * d.ipw;
  const int s0 = (other.n0 + rows_per_slice - 1) / rows_per_slice, s1 = std::min(d.split, (other.n0 + other.n) / rows_per_slice);
  const bool have_local = h->comm && s1 > s0;
  if (part == 1) {
    if (!have_local) return;
    ScopedKernelTimer t(h, kid);
    g.split0 = s0; g.nsplit = s1 - s0;
    launch_gemm(g, d.KP, h->stream);
    return;
  }
  ScopedKernelTimer t(h, kid);
  if (part == 2 && have_local) {
    if (s0 > 0) { g.split0 = 0; g.nsplit = s0; launch_gemm(g, d.KP, h->stream); }
    if (s1 < d.split) { g.split0 = s1; g.nsplit = d.split - s1; launch_gemm(g, d.KP, h->stream); }
    return;
  }
  launch_gemm(g, d.KP, h->stream);
}
// relayout (XT, XT2) + Gram of a factor that was just written
// the variational half sweep of direction d (other factor o) runs on the on-chip kernels (api_models.inc: enqueue_vb_sweep)
// Policy (same-box A/Bs): the product's fixed cost (bits x digit planes, column maxima, planes, slab reads) pays when
// a column loop is 64 columns of a 16-wave block -- 8192^2, K = 64: +7 % -- and does not at K <= 32 (4096^2: -15 %, 8192^2: -5 %),
// where the pair-panel kernel (kernel_sweep_vb.hip) stays.  BNMTF_VB_PATH=masked / pairs forces one of them -- read ONCE, when the
// model's layout is built: the relayout of one half sweep builds what the next one's kernel reads (the pair panels or not), so a
// switch flipped between two calls must not change the path of a model that exists (round 4's advice).
static bool vb_chip_ok(const Dir& d, const Dir& o) {
  if (!(d.mbits && o.XB && d.mslabs && d.nch == 1 && sweep_fast_supported(d.KP, d.pw))) return false;
  if (d.vb_path != 0) return d.vb_path == 1;
  return d.KP == 64 && d.use_wide;
}
static void enqueue_post(bnmtf_model* h, Dir& d, bool vb = false) {
  PostArgs g;
  memset(&g, 0, sizeof(g));
  g.X = d.X; g.rows = d.nglob; g.KP = d.KP; g.XT = d.XT; g.ldT = d.ldT; g.XT2 = d.XT2; g.ld2 = d.ldT;
  g.Cpart = d.Cpart; g.spart = d.spart; g.C64 = d.C64; g.C32 = d.C32; g.colsum = d.colsum;
  if (vb) { g.S2 = d.S2; g.S2T = d.S2T; g.s2part = d.s2part; g.colsum2 = d.colsum2; g.XS = d.XS; }
  // (the (E, S2) pair panels are read by the pair-panel VB kernel only: the on-chip path takes its masked sums from kernel_maskgemm.hip)
  if (vb && vb_chip_ok(h->rows, h->cols) && vb_chip_ok(h->cols, h->rows)) {
    g.XS = nullptr;
    // ... and needs the column maxima of [S2 | E^2] (the fixed-point grid of its digit planes): the Gram blocks see every row anyway
    g.mpart = d.xb_mpart; g.umax = d.xb_umax;
    d.xb_umax_posted = d.xb_mpart != nullptr;
  } else d.xb_umax_posted = false;
  g.snap = d.snap_dst; g.snapW = d.W; d.snap_dst = nullptr;
  launch_post(g, h->stream);
}
// Several GPUs, after a half sweep: the block of X this rank has just drawn goes to the other ranks (all-gather on the
// exchange stream) WHILE the rank forms the Gram partial and the column sums of its own rows; the partials -- one buffer,
// C64 | colsum -- are then summed over the ranks by one all-reduce (north_star's "all-reduce on the K x K Gram matrices";
// every rank used to form the whole Gram from the gathered factor), which runs beside the relayout of the gathered factor
// and the next contraction.  The compute stream waits for the gathered factor here and for the summed Gram in await_gram(),
// just ahead of its first reader.  Every collective is issued on the ONE exchange stream, in the same order on all ranks.
// next / next_kid: the direction whose contraction reads d's factor next.  Its inner slices over this rank's OWN rows of the
// factor are launched here, ahead of the wait for the other ranks' blocks (they are final, whatever the others send); the
// remaining slices behind the relayout of the gathered factor.  Every slice writes its own slab and the sweep adds the slabs
// in slab order, so the chain is the same bits as with one launch.
static int exchange_factor(bnmtf_model* h, Dir& d, Dir* next = nullptr, int next_kid = 0) {
  if (!h->comm) { enqueue_post(h, d); return BNMTF_OK; }
  // BNMTF_EXCHANGE=serial: every collective on the compute stream, in program order, nothing overlapped -- the fall-back
  // while the two-stream ordering below has not run on a node with several GPUs (round 3's advice; tests/test_rccl_two_process_gpu.py)
  const char* xe = getenv("BNMTF_EXCHANGE");            // (read per call: the sharded tests run both orderings in one process)
  if (xe && !strcmp(xe, "serial")) {
    PostArgs g;
    memset(&g, 0, sizeof(g));
    g.X = d.X; g.rows = d.nglob; g.KP = d.KP; g.XT = d.XT; g.ldT = d.ldT; g.XT2 = d.XT2; g.ld2 = d.ldT;
    g.Cpart = d.Cpart; g.spart = d.spart; g.C64 = d.C64; g.C32 = d.C32; g.colsum = d.colsum;
    launch_post_gram_rows(g, d.n0, d.n0 + d.n, h->stream);
    CHK(comm_allgather_factor(h->comm, d.X, d.KP, d.nglob, h->world, h->stream));
    CHK(comm_allreduce_sum(h->comm, d.C64, 64 * 64 + 64, h->stream));
    launch_gram_cast(d.C64, d.C32, d.KP * d.KP, h->stream);
    g.snap = d.snap_dst; g.snapW = d.W; d.snap_dst = nullptr;
    launch_post_layout(g, h->stream);
    d.gram_pending = false;
    return BNMTF_OK;
  }
  if (!h->xchg_stream) HIPCHK(hipStreamCreateWithFlags(&h->xchg_stream, hipStreamNonBlocking));
  for (hipEvent_t* e : {&d.ev_sweep, &d.ev_gram, &d.ev_gathered, &d.ev_gram_all})
    if (!*e) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  PostArgs g;
  memset(&g, 0, sizeof(g));
  g.X = d.X; g.rows = d.nglob; g.KP = d.KP; g.XT = d.XT; g.ldT = d.ldT; g.XT2 = d.XT2; g.ld2 = d.ldT;
  g.Cpart = d.Cpart; g.spart = d.spart; g.C64 = d.C64; g.C32 = d.C32; g.colsum = d.colsum;
  HIPCHK(hipEventRecord(d.ev_sweep, h->stream));
  launch_post_gram_rows(g, d.n0, d.n0 + d.n, h->stream);             // own rows only: they are final, whatever the others send
  HIPCHK(hipEventRecord(d.ev_gram, h->stream));
  if (next) enqueue_gemm(h, *next, d, next_kid, 1);
  HIPCHK(hipStreamWaitEvent(h->xchg_stream, d.ev_sweep, 0));
  CHK(comm_allgather_factor(h->comm, d.X, d.KP, d.nglob, h->world, h->xchg_stream));
  HIPCHK(hipEventRecord(d.ev_gathered, h->xchg_stream));
  HIPCHK(hipStreamWaitEvent(h->xchg_stream, d.ev_gram, 0));
  CHK(comm_allreduce_sum(h->comm, d.C64, 64 * 64 + 64, h->xchg_stream));          // C64 | colsum
  launch_gram_cast(d.C64, d.C32, d.KP * d.KP, h->xchg_stream);
  HIPCHK(hipEventRecord(d.ev_gram_all, h->xchg_stream));
  d.gram_pending = true;
  HIPCHK(hipStreamWaitEvent(h->stream, d.ev_gathered, 0));
  g.snap = d.snap_dst; g.snapW = d.W; d.snap_dst = nullptr;
  launch_post_layout(g, h->stream);
  if (next) { enqueue_gemm(h, *next, d, next_kid, 2); next->gemm_ahead = true; }
  return BNMTF_OK;
}
// the summed Gram of d (C32, C64, colsum) is about to be read on the compute stream
static int await_gram(bnmtf_model* h, Dir& d) {
  if (d.gram_pending) { HIPCHK(hipStreamWaitEvent(h->stream, d.ev_gram_all, 0)); d.gram_pending = false; }
  return BNMTF_OK;
}
// q hand-over between the half sweeps (the run loops switch it on: bnmf_gibbs_run, bnmf_vb_run): read this direction's region
// if the other direction's last sweep filled it, fill the other's.  Every ho_refresh-th iteration the rows sweep runs its
// pre-pass all the same: q handed back and forth collects one fp32 rounding per column update, the pre-pass starts from X
// again (tools/handover_drift.py: no drift to see at 64).
static void set_handover(bnmtf_model* h, Dir& d, const Dir& other, FastArgs& f, bool kernel_can) {
  f.ho_read = f.ho_write = 0; f.ho_nb_other = 0; f.ho_lds_floats = 0; f.ho_rows_total = (int)d.f_slots;
  f.ho_in = f.ho_out = f.ho_pk = f.ho_region_ofs = nullptr; f.ho_region = nullptr; f.ho_dst = nullptr;
  if (!(h->ho_active && d.ho_ready && other.ho_ready && kernel_can)) return;
  const bool refresh = &d == &h->rows && h->iteration % h->ho_refresh == 0;
  f.ho_read = d.ho_filled && !refresh;
  f.ho_in = d.ho_in; f.ho_region = d.ho_region; f.ho_region_ofs = d.ho_region_ofs;
  f.ho_write = 1; f.ho_nb_other = other.ho_blocks; f.ho_lds_floats = d.ho_lds_floats;
  f.ho_out = d.ho_out; f.ho_pk = d.ho_pk; f.ho_dst = other.ho_region;
  other.ho_filled = true; d.ho_filled = false;
}
static void enqueue_sweep(bnmtf_model* h, Dir& d, const Dir& other, SweepArgs& s, bool want_stats) {
  s.unit_list = nullptr;
  h->last_sweep_fast = false;
  if (h->use_fast && d.uw_ok && other.X && s.cond_k < 0 && s.mode != kSweepVB && !s.cov_S && !s.order && (!h->ho_enabled || h->uw_force)) {
    // few units per CU: one unit per wave (kernel_sweep_unit.hip); every unit of the direction (a layout with a unit beyond
    // kFastMaxSlots slots per lane is not built)
    FastArgs f;
    memset(&f, 0, sizeof(f));
    f.unit_map = d.u_unit_map; f.pair_E = d.u_pair_E; f.pair_base = d.u_pair_base; f.off16 = d.u_off16;
    f.npairs = d.n; f.mz = d.mz; f.pw = d.pw; f.nw = d.u_nw; f.nch = 1;
    f.XoT = other.XT; f.ldT_o = other.ldT; f.Xo = other.X; f.Xo_rows = other.nglob;
    f.stats = want_stats ? d.stats : nullptr;
    SweepArgs s2 = s;
    s2.acc = nullptr;
    if (h->ho_active) { d.ho_filled = false; }        // (q is not handed over by this shape: the next reader of a region rebuilds)
    launch_sweep_unit(s2, f, h->stream);
    h->last_sweep_fast = true;
    return;
  }
  if (h->use_fast && d.fast_ok && s.cond_k < 0 && s.mode != kSweepVB && (d.nch == 2 || sweep_fast_supported(d.KP, d.pw))) {
    FastArgs f;
    f.unit_map = d.f_unit_map; f.pair_E = d.f_pair_E; f.pair_base = d.f_pair_base; f.off = d.f_off;
    f.npairs = d.f_npairs; f.mz = d.mz; f.pw = d.nch == 2 ? d.pw_chunk : d.pw; f.nw = d.f_nw;
    f.nch = d.nch; f.mh = d.mh; f.pw1 = d.pw1; f.twin = d.use_twin ? 1 : 0;
    f.XoT = other.XT; f.ldT_o = other.ldT; f.XoT2 = other.XT2; f.ld2_o = other.ldT; f.Xo = other.X; f.Xo_rows = other.nglob;
    f.stats = want_stats ? d.stats : nullptr;
    SweepArgs s2 = s;
    s2.acc = nullptr;
    f.off16 = d.pair_ok ? d.f_off16 : nullptr;
    set_handover(h, d, other, f, !d.use_turns && d.nch == 1 && d.f_nw == d.ho_ppb && s.mode != kSweepVB);
#ifdef BNMTF_EXPERIMENTS
    if (d.use_wide && d.use_turns) launch_sweep_turns(s2, f, h->stream);
    else
#endif
    if (d.use_wide && !d.use_twin) launch_sweep_wide(s2, f, h->stream);
    else launch_sweep_fast(s2, f, h->stream);
    h->last_sweep_fast = true;
    if (d.f_gen_count == 0) return;
    s.unit_list = d.f_gen_units;           // the few units with more than kFastMaxSlots slots per lane
    s.n = d.f_gen_count;
  }
  launch_sweep(s, h->stream);
}
static SweepArgs sweep_args(bnmtf_model* h, Dir& d, const Dir& other, int mode, uint32_t stream_id) {
  SweepArgs s;
  memset(&s, 0, sizeof(s));
  s.min_x = mode == kSweepMode ? h->cur_min_x : 0.f;
  s.n = d.n; s.n0 = d.n0; s.K = d.W; s.KP = d.KP; s.mode = mode; s.cond_k = -1; s.qinit_only = 0; s.only_k = -1; s.vb_moments = 1; s.vb_stats = nullptr;
  s.col0 = h->col0;
  s.slabs = d.slabs; s.split = d.split; s.n_pad = d.n_pad; s.lambda = d.lambda;
  s.Xself = d.X; s.XselfT = nullptr; s.ldT_self = d.ldT;
  s.XoT = other.XT; s.ldT_o = other.ldT; s.C32 = other.C32;
  s.slot_ptr = d.slot_ptr; s.idx = d.idx; s.q = d.q;
  s.tau = h->tau_f;
  s.key0 = (uint32_t)h->seed; s.key1 = (uint32_t)(h->seed >> 32); s.it = (uint32_t)h->iteration; s.stream = stream_id;
  s.acc = nullptr;
  s.numer_out = d.numer; s.tau_out = d.taup;
  s.mu_self = d.mu; s.tau_self = d.tauq; s.var_self = d.var; s.S2self = d.S2; s.S2selfT = d.S2T;
  s.S2oT = other.S2T; s.colsum2_o = other.colsum2;
  return s;
}

static int upload_factor(bnmtf_model* h, Dir& d, const double* src) {
  std::vector<float> tmp((size_t)d.nglob * d.KP, 0.0f);
  for (int r = 0; r < d.nglob; ++r)
    for (int k = 0; k < d.W; ++k) tmp[(size_t)r * d.KP + k] = (float)src[(size_t)r * d.W + k];
  HIPCHK(hipMemcpyAsync(d.X, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  enqueue_post(h, d);
  return BNMTF_OK;
}
static int download_matrix(bnmtf_model* h, const float* dev, int rows, int W, int KP, double* dst) {
  std::vector<float> tmp((size_t)rows * KP);
  HIPCHK(hipMemcpyAsync(tmp.data(), dev, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < W; ++k) dst[(size_t)r * W + k] = (double)tmp[(size_t)r * KP + k];
  return BNMTF_OK;
}
static int set_tau(bnmtf_model* h, double tau) {
  const float tf = (float)tau;
  HIPCHK(hipMemcpyAsync(h->tau_d, &tau, sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->tau_f, &tf, sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return BNMTF_OK;
}

static int bnmtf_alloc_extras(bnmtf_model* h, const double* lambdaS);
static int build_standard(bnmtf_model* h, const double* lambda_S, const uint8_t* comm_id);
static void describe_model(bnmtf_model* h);
static int small_build(bnmtf_model* h, const float* R, const uint8_t* M);
static void small_free(bnmtf_model* h);
static int ensure_std(bnmtf_model* h);

// ------------------------------------------------------------ sample hand-off
// run() hands every sample to the host (all_U[it], all_V[it]; bnmf_gibbs_optimised.py:146-148).  The factor is packed
// ([rows][KP] -> [rows][W]) into a device snapshot slot on the compute stream -- a few microseconds -- and the compute
// stream moves on to the next iteration; a copy stream takes the slot to the host behind it.  Caller buffers that are
// pinned (bnmtf_host_alloc, or registered by the caller) receive the copy directly, at PCIe rate and with no host work;
// pageable ones go through a pinned ring that the calling thread empties kDepth iterations behind the enqueue front.
struct SampleSink {
  // Slots are handed over in GROUPS of kGroup iterations: one event recorded on the compute stream and one stream wait per
  // group, not per iteration -- an event record drains the compute queue for a few microseconds each time (round 1: eight
  // records cost 30 us per iteration).  Two groups of slots: one being filled while the other goes to the host.
#ifndef BNMTF_SAMPLE_GROUP
#define BNMTF_SAMPLE_GROUP 8
#endif
  static constexpr int kGroup = BNMTF_SAMPLE_GROUP, kGroups = 2, kDepth = kGroup * kGroups;
  struct Mat { const float* src; int rows, W, KP; float* dst; size_t off; };
  bnmtf_model* h = nullptr;
  Mat m[3]; int nmat = 0;
  size_t per_it = 0;            // floats per iteration over all matrices
  bool active = false, direct = true;
  int n_iter = 0, group_first = 0, drained = 0;     // first iteration of the group being filled; ring path: iterations already in the caller's arrays
  int next_copy = 0;                                // first iteration whose slot has not been given to the copy stream yet
  int pending_last[kGroups] = {-1, -1};             // ring path: last iteration of the group whose copy is in flight in event slot g

  void add(const float* src, int rows, int W, int KP, float* dst) {
    if (!dst) return;
    m[nmat++] = Mat{src, rows, W, KP, dst, per_it};
    per_it += (size_t)rows * W;
  }
  static bool pinned(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
  }
  int begin(bnmtf_model* h_, int n_iter_) {
    h = h_; n_iter = n_iter_; next_copy = 0;
    if (nmat == 0) return BNMTF_OK;
    active = true;
    for (int i = 0; i < nmat; ++i) direct = direct && pinned(m[i].dst) && pinned(m[i].dst + (size_t)n_iter * m[i].rows * m[i].W - 1);
    if (getenv("BNMTF_SAMPLES_RING")) direct = false;       // test hook: force the pageable path
    if (!h->copy_stream) {
      HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
      for (int s = 0; s < kGroups; ++s) {
        HIPCHK(hipEventCreateWithFlags(&h->snap_ready[s], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->copy_done[s], hipEventDisableTiming));
      }
    }
    if (h->snap_dev_cap < per_it * kDepth) {
      dfree(h->snap_dev);
      CHK(dalloc(&h->snap_dev, per_it * kDepth, false));
      h->snap_dev_cap = per_it * kDepth;
    }
    if (!direct && h->snap_host_cap < per_it * kDepth) {
      if (h->snap_host) (void)hipHostFree(h->snap_host);
      h->snap_host = nullptr; h->snap_host_cap = 0;
      HIPCHK(hipHostMalloc((void**)&h->snap_host, per_it * kDepth * sizeof(float), hipHostMallocDefault));
      h->snap_host_cap = per_it * kDepth;
    }
    return BNMTF_OK;
  }
  // ring path: the group in event slot g has landed in the pinned ring -> the caller's arrays
  int drain_group(int g) {
    if (pending_last[g] < 0) return BNMTF_OK;
    HIPCHK(hipEventSynchronize(h->copy_done[g]));
    for (; drained <= pending_last[g]; ++drained) {
      const float* slot = h->snap_host + (size_t)(drained % kDepth) * per_it;
      for (int i = 0; i < nmat; ++i)
        memcpy(m[i].dst + (size_t)drained * m[i].rows * m[i].W, slot + m[i].off, (size_t)m[i].rows * m[i].W * sizeof(float));
    }
    pending_last[g] = -1;
    return BNMTF_OK;
  }
  // before iteration `it` is enqueued: at the start of a group its slots must have been taken to the host (the group two back)
  int open_slot(int it) {
    if (!active || it % kGroup != 0 || it < kDepth) return BNMTF_OK;
    const int g = (it / kGroup) % kGroups;
    if (!direct) CHK(drain_group(g));
    HIPCHK(hipStreamWaitEvent(h->stream, h->copy_done[g], 0));
    return BNMTF_OK;
  }
  // where matrix `src` goes in the slot of iteration `it` (the relayout kernel that follows its sweep writes it there), or null
  float* slot_for(int it, const float* src) const {
    if (!active) return nullptr;
    for (int i = 0; i < nmat; ++i)
      if (m[i].src == src) return h->snap_dev + (size_t)(it % kDepth) * per_it + m[i].off;
    return nullptr;
  }
  // matrix `src` is final for iteration `it`: pack it into the slot (compute stream)
  void snapshot(int it, const float* src) {
    if (!active) return;
    for (int i = 0; i < nmat; ++i)
      if (m[i].src == src) launch_compact_rows(src, m[i].rows, m[i].W, m[i].KP, h->snap_dev + (size_t)(it % kDepth) * per_it + m[i].off, h->stream);
  }
  // every matrix of iteration `it` is in its slot; at the end of a group the group goes to the copy stream -- and in the LAST group
  // of the run every iteration on its own: what is still to be copied when the last sweep has finished is then one iteration, not
  // four (16 MB at cfg3 = 0.6 ms behind the compute stream, 7 % of a 20-iteration call)
  int close_slot(int it) {
    if (!active) return BNMTF_OK;
    const bool last_group = it / kGroup == (n_iter - 1) / kGroup;
    if (it % kGroup != kGroup - 1 && it != n_iter - 1 && !last_group) return BNMTF_OK;
    const int first = next_copy, g = (it / kGroup) % kGroups;
    next_copy = it + 1;
    HIPCHK(hipEventRecord(h->snap_ready[g], h->stream));
    HIPCHK(hipStreamWaitEvent(h->copy_stream, h->snap_ready[g], 0));
#ifdef BNMTF_EXPERIMENT_SAMPLES_NOCOPY
    constexpr bool nocopy = true;      // measurement build only (wrong results): the hand-off without its copies
#else
    constexpr bool nocopy = false;
#endif
    for (int j = first; j <= it && !nocopy; ++j) {
      const float* slot = h->snap_dev + (size_t)(j % kDepth) * per_it;
      if (direct) {
        for (int i = 0; i < nmat; ++i)
          HIPCHK(hipMemcpyAsync(m[i].dst + (size_t)j * m[i].rows * m[i].W, slot + m[i].off, (size_t)m[i].rows * m[i].W * sizeof(float),
                                hipMemcpyDeviceToHost, h->copy_stream));
      } else {
        HIPCHK(hipMemcpyAsync(h->snap_host + (size_t)(j % kDepth) * per_it, slot, per_it * sizeof(float), hipMemcpyDeviceToHost, h->copy_stream));
      }
    }
    HIPCHK(hipEventRecord(h->copy_done[g], h->copy_stream));
    pending_last[g] = it;
    return BNMTF_OK;
  }
  int finish() {
    if (!active) return BNMTF_OK;
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    if (!direct) {                                   // the groups still in the ring, oldest first
      const int g_last = ((n_iter - 1) / kGroup) % kGroups;
      CHK(drain_group((g_last + 1) % kGroups));
      CHK(drain_group(g_last));
    }
    active = false;
    return BNMTF_OK;
  }
  // an error return out of the iteration loop leaves copies in flight into the caller's arrays (or the ring): they are
  // waited for on EVERY way out, so nothing is written after run() has returned
  ~SampleSink() {
    if (active && h && h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
  }
};

static int ensure_rec(bnmtf_model* h, size_t n) {
  if (h->rec_cap >= n) return BNMTF_OK;
  dfree(h->rec); dfree(h->gunit);
  CHK(dalloc(&h->rec, n * 5));
  CHK(dalloc(&h->gunit, n));
  h->rec_cap = n;
  return BNMTF_OK;
}
// Gamma(alpha_s, 1) variates of the next n iterations (tau = variate / beta_s): they depend on the seed and the iteration
// number only, so the fp64 Marsaglia-Tsang loop runs on the host, off the device's critical path.
static int stage_gamma_variates(bnmtf_model* h, int n) {
  h->gunit_host.resize((size_t)n);
  const double shape = h->alpha + 0.5 * h->n_obs;
  for (int it = 0; it < n; ++it)
    h->gunit_host[it] = gamma_unit_draw(shape, (uint32_t)(h->iteration + it), kStreamTau, (uint32_t)h->seed, (uint32_t)(h->seed >> 32));
  HIPCHK(hipMemcpyAsync(h->gunit, h->gunit_host.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return BNMTF_OK;
}

// approx_expectation on the device: reset at the start of a run() call, add the sample of iteration `it` of that call
static int expectation_begin(bnmtf_model* h) {
  if (h->exp_burn < 0) return BNMTF_OK;
  if (!h->exp_rows) {
    CHK(dalloc(&h->exp_rows, (size_t)h->rows.xrows * h->rows.KP));
    CHK(dalloc(&h->exp_cols, (size_t)h->cols.xrows * h->cols.KP));
    CHK(dalloc(&h->exp_S, (size_t)std::max(h->K * h->L, 1)));
    CHK(dalloc(&h->exp_tau, 1));
  }
  HIPCHK(hipMemsetAsync(h->exp_rows, 0, sizeof(double) * h->rows.nglob * h->rows.KP, h->stream));
  HIPCHK(hipMemsetAsync(h->exp_cols, 0, sizeof(double) * h->cols.nglob * h->cols.KP, h->stream));
  HIPCHK(hipMemsetAsync(h->exp_S, 0, sizeof(double) * std::max(h->K * h->L, 1), h->stream));
  HIPCHK(hipMemsetAsync(h->exp_tau, 0, sizeof(double), h->stream));
  h->exp_count = 0;
  return BNMTF_OK;
}
static void expectation_add(bnmtf_model* h, int it) {
  if (h->exp_burn < 0 || it < h->exp_burn || (it - h->exp_burn) % h->exp_thin != 0) return;
  launch_accumulate(h->rows.X, (size_t)h->rows.nglob * h->rows.KP, h->exp_rows, h->tau_d, h->exp_tau, h->stream);
  launch_accumulate(h->cols.X, (size_t)h->cols.nglob * h->cols.KP, h->exp_cols, nullptr, nullptr, h->stream);
  if (h->L > 0) launch_accumulate(h->S, (size_t)h->K * h->L, h->exp_S, nullptr, nullptr, h->stream);
  h->exp_count++;
}

}  // namespace bnmtf

#include "api_small.inc"

using namespace bnmtf;

// ======================================================================= C ABI
extern "C" {

int bnmtf_version(void) { return 100; }
const char* bnmtf_last_error(void) { return g_err; }

int bnmtf_device_count(int* count) try {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { n = 0; (void)hipGetLastError(); }
  *count = n;
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmtf_comm_unique_id(uint8_t out[128]) { return comm_unique_id(out); }

int bnmtf_shard_range(int64_t n, int rank, int world, int64_t* first, int64_t* count) try {
  if (world < 1 || rank < 0 || rank >= world || n < 0) { set_error("bad shard request"); return BNMTF_EINVAL; }
  *first = n * rank / world;
  *count = n * (rank + 1) / world - *first;
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmtf_create(const bnmtf_problem* p, bnmtf_handle* out) try {
  *out = nullptr;
  if (!p || !p->R || !p->M || !p->lambda_rows || !p->lambda_cols) { set_error("bnmtf_create: null argument"); return BNMTF_EINVAL; }
  if (p->I < 1 || p->J < 1 || p->K < 1 || p->K > BNMTF_MAX_RANK || p->L < 0 || p->L > BNMTF_MAX_RANK) {
    set_error("bnmtf_create: unsupported shape I=%d J=%d K=%d L=%d (1 <= K,L <= %d)", p->I, p->J, p->K, p->L, BNMTF_MAX_RANK);
    return BNMTF_EINVAL;
  }
  if (p->L > 0 && !p->lambda_S) { set_error("bnmtf_create: lambda_S required when L > 0"); return BNMTF_EINVAL; }
  if (p->world < 1 || p->rank < 0 || p->rank >= p->world) { set_error("bnmtf_create: bad rank/world %d/%d", p->rank, p->world); return BNMTF_EINVAL; }
  if (p->world > 1 && !p->comm_id) { set_error("bnmtf_create: comm_id required when world > 1"); return BNMTF_EINVAL; }
  if (p->world > p->I || p->world > p->J) { set_error("bnmtf_create: world %d larger than a matrix dimension", p->world); return BNMTF_EINVAL; }
  HIPCHK(hipSetDevice(p->device));

  const auto t_create0 = std::chrono::steady_clock::now();
  bnmtf_model* h = new bnmtf_model();
  h->I = p->I; h->J = p->J; h->K = p->K; h->L = p->L;
  h->alpha = p->alpha; h->beta = p->beta; h->seed = p->seed;
  h->device = p->device; h->rank = p->rank; h->world = p->world;
  const int I = p->I, J = p->J;
  const float* R = p->R; const uint8_t* M = p->M;

  auto fail = [&](int rc) { bnmtf_destroy(h); return rc; };
  h->stream = device_pool().take_stream(p->device);
  if (!h->stream && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(BNMTF_EHIP); }

  // observed counts, training-mask constants (fp64) and the empty row/column check
  std::vector<uint32_t> rc(I, 0), cc(J, 0);
  double n_obs = 0, sR = 0, sR2 = 0;
  {
    constexpr int kRows = 64;                     // fixed chunks, combined in chunk order: the sums do not depend on the thread count
    const int nch = (I + kRows - 1) / kRows;
    std::vector<double> part((size_t)nch * 3, 0.0);
    std::vector<std::vector<uint32_t>> ccp(nch);
    parallel_chunks(I, kRows, [&](int a, int b) {
      const int ch = a / kRows;
      std::vector<uint32_t>& cl = ccp[ch];
      cl.assign(J, 0);
      double n = 0, s1 = 0, s2 = 0;
      for (int i = a; i < b; ++i) {
        uint32_t cnt = 0;
        for (int j = 0; j < J; ++j)
          if (M[(size_t)i * J + j]) {
            const double r = (double)R[(size_t)i * J + j];
            ++cnt; cl[j]++; s1 += r; s2 += r * r;
          }
        rc[i] = cnt; n += cnt;
      }
      part[(size_t)ch * 3] = n; part[(size_t)ch * 3 + 1] = s1; part[(size_t)ch * 3 + 2] = s2;
    });
    for (int ch = 0; ch < nch; ++ch) {
      n_obs += part[(size_t)ch * 3]; sR += part[(size_t)ch * 3 + 1]; sR2 += part[(size_t)ch * 3 + 2];
      for (int j = 0; j < J; ++j) cc[j] += ccp[ch][j];
    }
  }
  for (int i = 0; i < I; ++i) if (!rc[i]) { set_error("Fully unobserved row in R, row %d.", i); return fail(BNMTF_EINVAL); }
  for (int j = 0; j < J; ++j) if (!cc[j]) { set_error("Fully unobserved column in R, column %d.", j); return fail(BNMTF_EINVAL); }
  h->n_obs = n_obs; h->sumR = sR; h->sumR2 = sR2;

  CreateLaps laps;
  laps.t = t_create0;
  laps.lap("counts and sums over the mask");
  const int Wr = p->K, Wc = p->L > 0 ? p->L : p->K;
  int rcode;
  // the geometry every entry point may ask for, whichever path runs the model
  h->rows.nglob = I; h->rows.m = J; h->rows.W = Wr; h->rows.KP = Wr <= 32 ? 32 : 64;
  h->cols.nglob = J; h->cols.m = I; h->cols.W = Wc; h->cols.KP = Wc <= 32 ? 32 : 64;
  h->rows.obs_count = rc; h->cols.obs_count = cc;
  // Small models (kernel_small.hip: the whole run in one launch, one block per model) build only their own arena here -- ONE
  // allocation that also holds the full matrix, the mask and the scalars; the structures of the multi-launch path -- contraction
  // operands, slot layouts, Gram partials, ... -- are built the first time an entry point needs them (ensure_std).
  // BNMTF_SMALL=0: never.
  h->lam_rows.assign(p->lambda_rows, p->lambda_rows + (size_t)I * Wr);
  h->lam_cols.assign(p->lambda_cols, p->lambda_cols + (size_t)J * Wc);
  if (p->L > 0) h->lam_S.assign(p->lambda_S, p->lambda_S + (size_t)p->K * p->L);
  if (p->world == 1 && !getenv("BNMTF_FORCE_COMM")) {
    if ((rcode = small_build(h, R, M))) return fail(rcode);
  }
  if (!h->one_arena) {   // the full matrix and the training mask (predict / validation; the layout passes read them on the device); small scalars -- one allocation
    const size_t bR = ((size_t)I * J * sizeof(float) + 255) & ~(size_t)255, bM = ((size_t)I * J + 255) & ~(size_t)255;
    char* base = nullptr;
    if ((rcode = dalloc(&base, bR + bM + 256, false))) return fail(rcode);
    h->Rfull = reinterpret_cast<float*>(base); h->Mtrain = reinterpret_cast<uint8_t*>(base + bR);
    h->out6 = reinterpret_cast<double*>(base + bR + bM); h->tau_d = h->out6 + 8; h->acc = h->out6 + 12; h->tau_f = reinterpret_cast<float*>(h->out6 + 16);
    if (hipMemsetAsync(base + bR + bM, 0, 256, h->stream) != hipSuccess) { set_error("memset failed"); return fail(BNMTF_EHIP); }
  }
  if (hipMemcpyAsync(h->Rfull, R, (size_t)I * J * sizeof(float), hipMemcpyHostToDevice, h->stream) != hipSuccess) { set_error("copy R failed"); return fail(BNMTF_EHIP); }
  if (hipMemcpyAsync(h->Mtrain, M, (size_t)I * J, hipMemcpyHostToDevice, h->stream) != hipSuccess) { set_error("copy M failed"); return fail(BNMTF_EHIP); }
  if (h->small) {
    if (hipStreamSynchronize(h->stream) != hipSuccess) { set_error("bnmtf_create: uploads failed"); return fail(BNMTF_EHIP); }    // (R, M are the caller's)
    h->std_built = false;
    laps.lap("small-model arena");
  } else {
    if ((rcode = build_standard(h, p->lambda_S, p->comm_id))) return fail(rcode);
  }
  h->create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create0).count();
  describe_model(h);
  *out = h;
  return BNMTF_OK;
} BNMTF_ABI_GUARD
}  // extern "C"

namespace bnmtf {
// The multi-launch path's part of a handle: both directions' contraction operands and slot layouts, factor buffers, the BNMTF
// extras, the communicator, the q hand-over tables.  Called by bnmtf_create, or -- for a model that started on the small path --
// by the first entry point that needs it.
static int build_standard(bnmtf_model* h, const double* lambda_S, const uint8_t* comm_id) {
  const auto t_create0 = std::chrono::steady_clock::now();
  const int I = h->I, J = h->J;
  const int Wr = h->K, Wc = h->L > 0 ? h->L : h->K;
  struct { int K, L, rank, world; const uint8_t* comm_id; const double* lambda_S; uint64_t seed; } pv{h->K, h->L, h->rank, h->world, comm_id, lambda_S, h->seed};
  auto* p = &pv;
  auto fail = [&](int rc) { return rc; };
  CreateLaps laps;
  int rcode;
  const std::vector<uint32_t> rc = h->rows.obs_count, cc = h->cols.obs_count;
  rcode = build_dir(h->rows, I, J, Wr, p->rank, p->world, h->lam_rows.data(), h->Rfull, h->Mtrain, I, J, true, rc, h->stream);
  if (rcode) return fail(rcode);
  rcode = build_dir(h->cols, J, I, Wc, p->rank, p->world, h->lam_cols.data(), h->Rfull, h->Mtrain, I, J, false, cc, h->stream);
  if (rcode) return fail(rcode);
  laps.lap("both directions built (total)");
  h->rows.obs_count = rc; h->cols.obs_count = cc;
  if ((rcode = alloc_factor(h->rows, h->cols.inner_pad))) return fail(rcode);
  if ((rcode = alloc_factor(h->cols, h->rows.inner_pad))) return fail(rcode);

  laps.lap("factor buffers");
  if (!h->Ad && (rcode = dalloc(&h->Ad, (size_t)I * 64))) return fail(rcode);
  if (!h->Bd && (rcode = dalloc(&h->Bd, (size_t)J * 64))) return fail(rcode);
  if (p->L > 0 && (rcode = dalloc(&h->S, (size_t)p->K * p->L))) return fail(rcode);
  if (p->L > 0 && (rcode = bnmtf_alloc_extras(h, p->lambda_S))) return fail(rcode);

  if (p->world > 1) {
    if ((rcode = comm_create(&h->comm, p->comm_id, p->rank, p->world, h->stream))) return fail(rcode);
    // every rank must hold the same Philox key (same draws, same tau variates): compare through the communicator
    const double mine[4] = {(double)(uint32_t)p->seed, (double)(uint32_t)(p->seed >> 32), -(double)(uint32_t)p->seed, -(double)(uint32_t)(p->seed >> 32)};
    double got[4];
    if (hipMemcpy(h->acc, mine, sizeof(mine), hipMemcpyHostToDevice) != hipSuccess) { set_error("seed check: copy failed"); return fail(BNMTF_EHIP); }
    if ((rcode = comm_allreduce_max(h->comm, h->acc, 4, h->stream))) return fail(rcode);
    if (hipStreamSynchronize(h->stream) != hipSuccess || hipMemcpy(got, h->acc, sizeof(got), hipMemcpyDeviceToHost) != hipSuccess) { set_error("seed check: copy failed"); return fail(BNMTF_EHIP); }
    (void)hipMemset(h->acc, 0, 4 * sizeof(double));
    if (memcmp(mine, got, sizeof(mine)) != 0) {
      set_error("bnmtf_create: the ranks were given different seeds (this rank %llu): a sharded model needs one shared seed", (unsigned long long)p->seed);
      return fail(BNMTF_EINVAL);
    }
  } else if (getenv("BNMTF_FORCE_COMM")) {     // test hook: run the RCCL exchange path with a 1-rank communicator
    uint8_t id[128];
    if ((rcode = comm_unique_id(id))) return fail(rcode);
    if ((rcode = comm_create(&h->comm, id, 0, 1, h->stream))) return fail(rcode);
  }

  // q hand-over tables (BNMF on one GPU, both directions on the 16-wave or the plain 8-wave kernel).  By default only for
  // problems of >= 64 blocks per direction: below that the sweep is launch-bound and the pre-pass costs next to nothing.
  // BNMTF_HANDOVER=0: never; =1: whenever the tables can be built.
  {
    const char* e = getenv("BNMTF_HANDOVER");
    auto blocks = [](const Dir& d) { return d.ho_ppb > 0 ? d.f_npairs / d.ho_ppb : 0; };
    const bool want = p->L == 0 && !h->comm && (e ? atoi(e) != 0 : std::min(blocks(h->rows), blocks(h->cols)) >= 64);
    const auto t_ho0 = std::chrono::steady_clock::now();
    if (want && build_handover(h, h->rows, h->cols) && build_handover(h, h->cols, h->rows)) h->rows.ho_ready = h->cols.ho_ready = h->ho_enabled = true;
    if (const char* r = getenv("BNMTF_HANDOVER_REFRESH")) h->ho_refresh = (uint64_t)std::max(1, atoi(r));
    if (const char* u = getenv("BNMTF_UNIT")) h->uw_force = atoi(u) == 1;      // A/B switch: the unit-per-wave shape also where q could be handed over
    if (getenv("BNMTF_CREATE_TIMING")) fprintf(stderr, "hand-over tables: %.1f ms (create so far %.1f ms)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ho0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create0).count());
  }
  h->std_built = true;
  h->create_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create0).count();
  return BNMTF_OK;
}
static void describe_model(bnmtf_model* h) {
  const int I = h->I, J = h->J;
  struct { int K, L, rank, world; } pv{h->K, h->L, h->rank, h->world};
  auto* p = &pv;
  const double n_obs = h->n_obs;
  char buf[1024];
  snprintf(buf, sizeof(buf),
           "I=%d J=%d K=%d L=%d rank=%d/%d rows[n=%d n_pad=%d split=%d ipw=%d inner_pad=%d nmiss=%zu nslots=%zu sweep_nw=%d turns=%d twin=%d handover=%d emax=%d generic_units=%d] "
           "cols[n=%d n_pad=%d split=%d ipw=%d inner_pad=%d nmiss=%zu nslots=%zu sweep_nw=%d turns=%d emax=%d generic_units=%d] n_obs=%.0f create_ms=%.0f",
           I, J, p->K, p->L, p->rank, p->world, h->rows.n, h->rows.n_pad, h->rows.split, h->rows.ipw, h->rows.inner_pad,
           h->rows.nmiss, h->rows.nslots, h->rows.f_nw, (int)h->rows.use_turns, (int)h->rows.use_twin, (int)h->ho_enabled, h->rows.f_emax, h->rows.f_gen_count, h->cols.n, h->cols.n_pad, h->cols.split,
           h->cols.ipw, h->cols.inner_pad, h->cols.nmiss, h->cols.nslots, h->cols.f_nw, (int)h->cols.use_turns, h->cols.f_emax, h->cols.f_gen_count, n_obs, h->create_ms);
  h->description = buf;
  // which directions run the unit-per-wave sweep (kernel_sweep_unit.hip), its unit waves per block and most slots per lane
  snprintf(buf, sizeof(buf), " unit_sweep[rows=%d/%d/%d cols=%d/%d/%d]", (int)(h->rows.uw_ok && (!h->ho_enabled || h->uw_force)), h->rows.u_nw, h->rows.u_emax,
           (int)(h->cols.uw_ok && (!h->ho_enabled || h->uw_force)), h->cols.u_nw, h->cols.u_emax);
  h->description += buf;
  if (h->small) {
    snprintf(buf, sizeof(buf), " small[block=%d entry_threads=%d/%d slots=%d/%d lds=%zu std_built=%d]", h->small->nt, h->small->dev.rows.nthreads, h->small->dev.cols.nthreads,
             h->small->dev.rows.em, h->small->dev.cols.em, h->small->lds_bytes, (int)h->std_built);
    h->description += buf;
  }
}
}  // namespace bnmtf
extern "C" {

int bnmtf_destroy(bnmtf_handle h) try {
  if (!h) return BNMTF_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->comm) comm_destroy(h->comm);
  small_free(h);            // (first: a small model's arena also holds Rfull, the posterior sums, Ad / Bd -- their pointers are cleared)
  free_dir(h->rows); free_dir(h->cols); free_dir(h->reff); free_dir(h->ceff);
  dfree(h->slabsS); dfree(h->CfS); dfree(h->deltaS); dfree(h->s_partial); dfree(h->s_w); dfree(h->s_omp); dfree(h->lambdaS); dfree(h->s_numer); dfree(h->s_taup);
  dfree(h->exp_rows); dfree(h->exp_cols); dfree(h->exp_S); dfree(h->exp_tau);
  dfree(h->muS); dfree(h->tauS); dfree(h->varS); dfree(h->mv_rows); dfree(h->mv_cols); dfree(h->tri_order); dfree(h->tri_sums); dfree(h->tri_third); dfree(h->ss_Aperm);
  dfree(h->ss_Wc); dfree(h->ss_Gc); dfree(h->ss_cands); dfree(h->ss_slabs); dfree(h->ss_AB); dfree(h->ss_r); dfree(h->ss_bpart); dfree(h->ss_tinv); dfree(h->ss_rec);
  dfree(h->Rfull); h->Mtrain = nullptr; h->out6 = nullptr; h->tau_d = nullptr; h->tau_f = nullptr; h->acc = nullptr;   // (one allocation: bnmtf_create)
  dfree(h->Mscratch); dfree(h->Ad); dfree(h->Bd); dfree(h->AdW); dfree(h->BdW);
  dfree(h->A2d); dfree(h->B2d); dfree(h->vb_rec); dfree(h->vbred);
  dfree(h->rec); dfree(h->gunit); dfree(h->S);
  for (auto& pe : h->pending_events) { (void)hipEventDestroy(pe.second.first); (void)hipEventDestroy(pe.second.second); }
  for (auto e : h->event_pool) (void)hipEventDestroy(e);
  if (h->copy_stream) {
    (void)hipStreamSynchronize(h->copy_stream);
    for (int s2 = 0; s2 < 8; ++s2) { if (h->snap_ready[s2]) (void)hipEventDestroy(h->snap_ready[s2]); if (h->copy_done[s2]) (void)hipEventDestroy(h->copy_done[s2]); }
    (void)hipStreamDestroy(h->copy_stream);
  }
  dfree(h->snap_dev);
  if (h->snap_host) (void)hipHostFree(h->snap_host);
  if (h->xchg_stream) { (void)hipStreamSynchronize(h->xchg_stream); (void)hipStreamDestroy(h->xchg_stream); }
  if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
  if (h->ev_aux0) (void)hipEventDestroy(h->ev_aux0);
  if (h->ev_aux1) (void)hipEventDestroy(h->ev_aux1);
  if (h->stream && !(h->pool_stream && device_pool().give_stream(h->device, h->stream))) (void)hipStreamDestroy(h->stream);
  delete h;
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmtf_sync(bnmtf_handle h) try {
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmtf_omega_counts(bnmtf_handle h, uint64_t* total, uint32_t* row, uint32_t* col) try {
  if (total) *total = (uint64_t)h->n_obs;
  if (row) memcpy(row, h->rows.obs_count.data(), sizeof(uint32_t) * h->I);
  if (col) memcpy(col, h->cols.obs_count.data(), sizeof(uint32_t) * h->J);
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmtf_host_alloc(size_t bytes, void** out) try {
  *out = nullptr;
  const hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();        // the caller falls back to pageable memory: the next call must not find this error waiting
    *out = nullptr;
    set_error("hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    return BNMTF_EHIP;
  }
  return BNMTF_OK;
} BNMTF_ABI_GUARD
int bnmtf_host_free(void* p) try {
  if (p) HIPCHK(hipHostFree(p));
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmtf_set_expectation(bnmtf_handle h, int burn_in, int thinning) try {
  if (burn_in >= 0 && thinning < 1) { set_error("thinning must be >= 1"); return BNMTF_EINVAL; }
  h->exp_burn = burn_in; h->exp_thin = thinning < 1 ? 1 : thinning;
  return BNMTF_OK;
} BNMTF_ABI_GUARD
int bnmtf_get_expectation(bnmtf_handle h, double* A, double* S, double* B, double* tau, uint64_t* count) try {
  if (!h->exp_rows || h->exp_count == 0) { set_error("no samples accumulated (bnmtf_set_expectation before run, burn_in < iterations)"); return BNMTF_ESTATE; }
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const double inv = 1.0 / (double)h->exp_count;
  auto fetch = [&](const double* dev, int rows, int W, int KP, double* dst) -> int {
    std::vector<double> tmp((size_t)rows * KP);
    HIPCHK(hipMemcpy(tmp.data(), dev, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int r = 0; r < rows; ++r) for (int k = 0; k < W; ++k) dst[(size_t)r * W + k] = tmp[(size_t)r * KP + k] * inv;
    return BNMTF_OK;
  };
  if (A) CHK(fetch(h->exp_rows, h->I, h->rows.W, h->rows.KP, A));
  if (B) CHK(fetch(h->exp_cols, h->J, h->cols.W, h->cols.KP, B));
  if (S && h->L > 0) CHK(fetch(h->exp_S, h->K, h->L, h->L, S));
  if (tau) { double t; HIPCHK(hipMemcpy(&t, h->exp_tau, sizeof(double), hipMemcpyDeviceToHost)); *tau = t * inv; }
  if (count) *count = h->exp_count;
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmtf_set_iteration(bnmtf_handle h, uint64_t it) { h->iteration = it; return BNMTF_OK; }
int bnmtf_set_tau(bnmtf_handle h, double tau) try {        // the noise precision alone (the factors on the device stay as they are)
  if (!h->have_state) { set_error("bnmtf_set_tau before the state is set"); return BNMTF_ESTATE; }
  HIPCHK(hipSetDevice(h->device));
  return set_tau(h, tau);
} BNMTF_ABI_GUARD
int bnmtf_get_iteration(bnmtf_handle h, uint64_t* it) { *it = h->iteration; return BNMTF_OK; }

int bnmtf_set_minimum_tn(bnmtf_handle h, double minimum_TN) try {
  if (!(minimum_TN >= 0.0)) { set_error("minimum_TN must be >= 0"); return BNMTF_EINVAL; }
  h->min_tn = minimum_TN;
  return BNMTF_OK;
} BNMTF_ABI_GUARD
int bnmtf_set_profiling(bnmtf_handle h, int enable) try {
  // 0: off; 1: every kernel; 2 + k: kernel k only (so that a timed region carries two event records, not eight)
  h->profiling = enable == 0 ? 0u : (enable == 1 ? 0xFFFFFFFFu : 1u << (unsigned)((enable - 2) & 31));
  h->profile_stride = enable >= 2 ? (uint64_t)((enable - 2) >> 5) + 1 : 1;     // every n-th iteration only: an event record costs the queue a few microseconds
  for (int i = 0; i < BNMTF_KERNEL_COUNT; ++i) { h->kernel_ms[i] = 0; h->kernel_launches[i] = 0; }
  return BNMTF_OK;
} BNMTF_ABI_GUARD
int bnmtf_set_sweep_path(bnmtf_handle h, int fast) { h->use_fast = fast != 0; h->ho_regions_current = false; return BNMTF_OK; }
int bnmtf_comm_info(bnmtf_handle h, int* kind, int* ranks) { return comm_info(h->comm, kind, ranks); }
int bnmtf_has_experiments(void) try {
#ifdef BNMTF_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
} BNMTF_ABI_GUARD
int bnmtf_set_small_path(bnmtf_handle h, int mode) try {
  if (mode < 0 || mode > 2) { set_error("bnmtf_set_small_path: mode 0 (never), 1 (auto) or 2 (always)"); return BNMTF_EINVAL; }
  h->small_mode = mode;
  return BNMTF_OK;
} BNMTF_ABI_GUARD
int bnmtf_is_small(bnmtf_handle h, int* out) { *out = small_wanted(h) ? 1 : 0; return BNMTF_OK; }
int bnmtf_kernel_stats(bnmtf_handle h, int kernel, double* total_ms, uint64_t* launches) try {
  if (kernel < 0 || kernel >= BNMTF_KERNEL_COUNT) { set_error("bad kernel id"); return BNMTF_EINVAL; }
  HIPCHK(hipStreamSynchronize(h->stream));
  drain_events(h);
  *total_ms = h->kernel_ms[kernel]; *launches = h->kernel_launches[kernel];
  return BNMTF_OK;
} BNMTF_ABI_GUARD
int bnmtf_describe(bnmtf_handle h, char* buf, size_t buflen) try {
  // (+ which kernels the last variational half sweep ran on: vb_chip_ok)
  static const char* const kVbPath[] = {"", " vb_sweep=generic", " vb_sweep=pairs", " vb_sweep=masked"};
  static const char* const kTriPath[] = {"", " tri_vb_sweeps=generic", " tri_vb_sweeps=pairs+cov", ""};      // (bnmtf_vb_run: api_trivb.inc enqueue_tri_sweep)
  snprintf(buf, buflen, "%s%s%s", h->description.c_str(), kVbPath[h->last_vb_path & 3], kTriPath[h->last_tri_path & 3]);
  return BNMTF_OK;
} BNMTF_ABI_GUARD

// ------------------------------------------------------------------ BNMF Gibbs
int bnmf_set_state(bnmtf_handle h, const double* U, const double* V, double tau) try {
  if (h->L != 0) { set_error("bnmf_set_state on a BNMTF handle"); return BNMTF_ESTATE; }
  HIPCHK(hipSetDevice(h->device));
  h->std_cur = false; h->small_cur = false;
  if (h->std_built) {
    CHK(upload_factor(h, h->rows, U));
    CHK(upload_factor(h, h->cols, V));
    h->std_cur = true;
  }
  if (h->small) CHK(small_upload_state(h, U, V));
  CHK(set_tau(h, tau));
  h->have_state = true;
  h->ho_regions_current = false;          // (q hand-over: whatever the regions hold belongs to the state that was just replaced)
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmf_get_state(bnmtf_handle h, double* U, double* V, double* tau) try {
  if (!h->have_state) { set_error("no state set"); return BNMTF_ESTATE; }
  HIPCHK(hipSetDevice(h->device));
  if (h->small && h->small_cur) CHK(small_download_state(h, U, V));
  else {
    if (U) CHK(download_matrix(h, h->rows.X, h->I, h->rows.W, h->rows.KP, U));
    if (V) CHK(download_matrix(h, h->cols.X, h->J, h->cols.W, h->cols.KP, V));
  }
  if (tau) {
    HIPCHK(hipMemcpyAsync(tau, h->tau_d, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmf_cond_params(bnmtf_handle h, int which, int k, double* numer_out, double* tau_out) try {
  if (!h->have_state) { set_error("no state set"); return BNMTF_ESTATE; }
  if (h->world != 1) { set_error("cond_params is a single-GPU test hook"); return BNMTF_EINVAL; }
  Dir& d = which == 0 ? h->rows : h->cols;
  Dir& o = which == 0 ? h->cols : h->rows;
  if (k < 0 || k >= d.W) { set_error("column %d out of range", k); return BNMTF_EINVAL; }
  HIPCHK(hipSetDevice(h->device));
  CHK(ensure_std(h));
  enqueue_gemm(h, d, o, which == 0 ? BNMTF_KERNEL_GEMM_ROWS : BNMTF_KERNEL_GEMM_COLS);
  SweepArgs s = sweep_args(h, d, o, kSweepDraw, which == 0 ? kStreamRows : kStreamCols);
  s.cond_k = k;
  enqueue_sweep(h, d, o, s, false);
  HIPCHK(hipMemcpyAsync(numer_out, d.numer, sizeof(double) * d.n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(tau_out, d.taup, sizeof(double) * d.n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmf_gibbs_run(bnmtf_handle h, int n_iter, int update, float* U_out, float* V_out,
                   double* tau_out, double* perf_out, double* times_out) try {
  if (h->L != 0) { set_error("bnmf_gibbs_run on a BNMTF handle"); return BNMTF_ESTATE; }
  if (!h->have_state) { set_error("bnmf_gibbs_run before bnmf_set_state"); return BNMTF_ESTATE; }
  if (n_iter < 0) { set_error("negative iteration count"); return BNMTF_EINVAL; }
  if (n_iter == 0) return BNMTF_OK;
  HIPCHK(hipSetDevice(h->device));
  if (update < 0 || update > BNMTF_UPDATE_ICM) { set_error("unknown update rule"); return BNMTF_EINVAL; }
  if (small_wanted(h)) {                       // a small model: the whole call is one launch (kernel_small.hip)
    SmallOut o{U_out, V_out, tau_out, perf_out, times_out};
    return small_run_many(&h, 1, n_iter, update, &o);
  }
  CHK(ensure_std(h));
  h->std_cur = true; h->small_cur = false;
  CHK(ensure_rec(h, (size_t)n_iter));
  const int mode = update == BNMTF_UPDATE_DRAW ? kSweepDraw : kSweepMode;
  h->cur_min_x = update == BNMTF_UPDATE_ICM ? (float)h->min_tn : 0.f;
  Dir& r = h->rows; Dir& c = h->cols;
  if (mode == kSweepDraw) CHK(stage_gamma_variates(h, n_iter));
  // q hand-over between the half sweeps: from this call's first rows sweep (a pre-pass) on; off again when the call returns
  HandoverScope ho_scope{h};
  // the [3] accumulator is only written by the generic sweep kernel (and summed across ranks): zero once, reset when used
  const bool acc_used = h->comm != nullptr || !h->use_fast || !c.fast_ok || c.f_gen_count > 0 || !(c.nch == 2 || sweep_fast_supported(c.KP, c.pw));
  HIPCHK(hipMemsetAsync(h->acc, 0, 4 * sizeof(double), h->stream));
  EventList ev;
  CHK(ev.create(times_out ? n_iter + 1 : 0));
  SampleSink sink;
  sink.add(r.X, h->I, r.W, r.KP, U_out);
  sink.add(c.X, h->J, c.W, c.KP, V_out);
  CHK(sink.begin(h, n_iter));
  CHK(expectation_begin(h));
  if (times_out) HIPCHK(hipEventRecord(ev[0], h->stream));

  for (int it = 0; it < n_iter; ++it) {
    CHK(sink.open_slot(it));
    // ---- U columns: P = R~ . V, then the K sequential row-wise updates
    if (!r.gemm_ahead) enqueue_gemm(h, r, c, BNMTF_KERNEL_GEMM_ROWS);      // (several GPUs: launched inside the previous iteration's exchange of V)
    r.gemm_ahead = false;
    CHK(await_gram(h, c));                       // V^T V of the previous iteration's exchange
    {
      ScopedKernelTimer t(h, BNMTF_KERNEL_SWEEP_ROWS);
      SweepArgs s = sweep_args(h, r, c, mode, kStreamRows);
      enqueue_sweep(h, r, c, s, false);
    }
#ifdef BNMTF_EXPERIMENTS
    static const bool snap_compact = getenv("BNMTF_SNAP_COMPACT") != nullptr;      // A/B switch: the packing kernel of round 2
#else
    constexpr bool snap_compact = false;
#endif
    if (!snap_compact) r.snap_dst = sink.slot_for(it, r.X);         // the sample of U goes out with the relayout (no packing kernel of its own)
    CHK(exchange_factor(h, r, &c, BNMTF_KERNEL_GEMM_COLS));        // one GPU: relayout + Gram; several: see exchange_factor
    if (snap_compact) sink.snapshot(it, r.X);
    // ---- V columns: Pv = R~^T . U
    if (!c.gemm_ahead) enqueue_gemm(h, c, r, BNMTF_KERNEL_GEMM_COLS);
    c.gemm_ahead = false;
    if (acc_used && it > 0) HIPCHK(hipMemsetAsync(h->acc, 0, 4 * sizeof(double), h->stream));
    CHK(await_gram(h, r));
    {
      ScopedKernelTimer t(h, BNMTF_KERNEL_SWEEP_COLS);
      SweepArgs s = sweep_args(h, c, r, mode, kStreamCols);
      s.acc = h->acc;
      enqueue_sweep(h, c, r, s, true);
    }
    const bool fast_stats = h->last_sweep_fast;
    if (h->comm && fast_stats) launch_sum_stats(c.stats, c.stats_blocks, h->acc, h->stream);   // fold the slab before the exchange
    if (!snap_compact) c.snap_dst = sink.slot_for(it, c.X);
    CHK(exchange_factor(h, c, it + 1 < n_iter ? &r : nullptr, BNMTF_KERNEL_GEMM_ROWS));
    if (snap_compact) sink.snapshot(it, c.X);
    if (h->comm) {
      // the three sums of the SSE identity: behind the Gram on the exchange stream (its event covers the fold above)
      if (h->xchg_stream) {
        CHK(comm_allreduce_sum(h->comm, h->acc, 4, h->xchg_stream));
        HIPCHK(hipEventRecord(c.ev_gram_all, h->xchg_stream));
      } else CHK(comm_allreduce_sum(h->comm, h->acc, 4, h->stream));       // (BNMTF_EXCHANGE=serial)
    }
    CHK(sink.close_slot(it));
    CHK(await_gram(h, c)); CHK(await_gram(h, r));
    // ---- tau and the metrics of this sample
    FinishArgs f;
    f.copy_src = nullptr; f.copy_dst = nullptr; f.copy_n = 0;
    f.Cr64 = r.C64; f.Cc64 = c.C64; f.sr = r.colsum; f.sc = c.colsum; f.KP = r.KP;
    f.acc = h->acc; f.stats = (fast_stats && !h->comm) ? c.stats : nullptr; f.nstats = (fast_stats && !h->comm) ? c.stats_blocks : 0;
    f.n_obs = h->n_obs; f.sumR = h->sumR; f.sumR2 = h->sumR2;
    f.alpha = h->alpha; f.beta = h->beta; f.update = update;
    f.key0 = (uint32_t)h->seed; f.key1 = (uint32_t)(h->seed >> 32); f.it = (uint32_t)h->iteration;
    f.gunit = mode == kSweepDraw ? h->gunit + it : nullptr;
    f.tau_d = h->tau_d; f.tau_f = h->tau_f; f.rec = h->rec + (size_t)it * 5;
    launch_finish(f, h->stream);
    expectation_add(h, it);
    if (times_out) HIPCHK(hipEventRecord(ev[it + 1], h->stream));
    h->iteration++;
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  CHK(sink.finish());
  HIPCHK(hipGetLastError());
  ho_scope.commit();
  drain_events(h);
  std::vector<double> rec((size_t)n_iter * 5);
  HIPCHK(hipMemcpy(rec.data(), h->rec, rec.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int it = 0; it < n_iter; ++it) {
    if (tau_out) tau_out[it] = rec[(size_t)it * 5];
    if (perf_out) for (int m = 0; m < 3; ++m) perf_out[(size_t)it * 3 + m] = rec[(size_t)it * 5 + 1 + m];
    if (times_out) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, ev[0], ev[it + 1]);
      times_out[it] = (double)ms * 1e-3;
    }
  }
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmf_gibbs_run_many(const bnmtf_handle* hs, int n_models, int n_iter, int update, float* const* U_outs, float* const* V_outs,
                        double* const* tau_outs, double* const* perf_outs, double* const* times_outs,
                        double* const* U_final, double* const* V_final, double* const* tau_final) try {
  if (n_models < 0 || n_iter < 0) { set_error("negative count"); return BNMTF_EINVAL; }
  if (n_models == 0 || n_iter == 0) return BNMTF_OK;
  if (update < 0 || update > BNMTF_UPDATE_ICM) { set_error("unknown update rule"); return BNMTF_EINVAL; }
  for (int b = 0; b < n_models; ++b) if (hs[b]->L != 0) { set_error("bnmf_gibbs_run_many on a BNMTF handle (model %d)", b); return BNMTF_ESTATE; }
  // models of the one-launch path go down in one grid per device; the others run one after the other
  std::vector<bnmtf_model*> batch; std::vector<SmallOut> outs;
  auto out_of = [&](int b) { return SmallOut{U_outs ? U_outs[b] : nullptr, V_outs ? V_outs[b] : nullptr, tau_outs ? tau_outs[b] : nullptr,
                                             perf_outs ? perf_outs[b] : nullptr, times_outs ? times_outs[b] : nullptr,
                                             U_final ? U_final[b] : nullptr, V_final ? V_final[b] : nullptr, tau_final ? tau_final[b] : nullptr}; };
  std::vector<char> taken(n_models, 0);
  // models on model b's device that actually JOIN the launch, b included: the rule of a CU-filling or wide-rank model depends on
  // the batch size (small_wanted), so the count is taken to its fixed point -- a model that would run alone on the multi-launch
  // path does not make a "batch" for the others (round 5's advice)
  auto peers = [&](int b) {
    int n = 0;
    for (int c = 0; c < n_models; ++c) n += (hs[c]->small && hs[c]->small_mode != 0 && hs[c]->device == hs[b]->device) ? 1 : 0;
    for (int round = 0; round < n_models && n > 0; ++round) {
      int m = 0;
      for (int c = 0; c < n_models; ++c) m += (hs[c]->device == hs[b]->device && small_wanted(hs[c], n)) ? 1 : 0;
      if (m == n) break;
      n = m;
    }
    return n;
  };
  for (int b = 0; b < n_models; ++b) {
    if (taken[b]) continue;
    const int np = peers(b);
    if (!small_wanted(hs[b], np)) {
      const SmallOut o = out_of(b);
      CHK(bnmf_gibbs_run(hs[b], n_iter, update, o.U, o.V, o.tau, o.perf, o.times));
      if (o.U_final || o.V_final || o.tau_final) CHK(bnmf_get_state(hs[b], o.U_final, o.V_final, o.tau_final));
      taken[b] = 1;
      continue;
    }
    batch.clear(); outs.clear();
    for (int c = b; c < n_models; ++c)
      if (!taken[c] && small_wanted(hs[c], np) && hs[c]->device == hs[b]->device) {
        bool dup = false;
        for (bnmtf_model* x : batch) dup = dup || x == hs[c];
        if (dup) { set_error("bnmf_gibbs_run_many: the same handle twice"); return BNMTF_EINVAL; }
        batch.push_back(hs[c]); outs.push_back(out_of(c)); taken[c] = 1;
      }
    CHK(small_run_many(batch.data(), (int)batch.size(), n_iter, update, outs.data()));
  }
  return BNMTF_OK;
} BNMTF_ABI_GUARD

// ---------------------------------------------------------------------- ranks above 64: a factorisation as column blocks
// The reference takes any K (bnmf_gibbs_optimised.py:54-78).  The kernels hold a latent factor per wave lane (K <= 64), so a wider
// model runs as ceil(K / 64) COLUMN BLOCKS, one handle each (bnmtf_amd/_blocked.py): the conditionals of block b's columns given the
// other blocks are those of a rank-K_b model on the residual data R - sum_{b' != b} U_b' V_b'^T -- exactly the reference's
// sequential column order when the blocks' half sweeps run in turn, rows first (:134-137), then columns (:139-142).
int bnmf_set_column_block(bnmtf_handle h, int col0) try {
  if (col0 < 0) { set_error("negative column offset"); return BNMTF_EINVAL; }
  if (h->L != 0 || h->comm) { set_error("column blocks: BNMF handles on one GPU"); return BNMTF_ESTATE; }
  h->col0 = (uint32_t)col0;
  h->block_mode = true;
  h->small_mode = 0;                          // (the one-launch kernel runs whole iterations: not a block's half sweeps)
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmf_set_residual_data(bnmtf_handle h, const bnmtf_handle* others, int n_others) try {
  if (n_others < 0 || n_others > kMaxOtherBlocks) { set_error("at most %d other column blocks", kMaxOtherBlocks); return BNMTF_EINVAL; }
  if (h->comm) { set_error("residual data: one GPU"); return BNMTF_ESTATE; }       // (the target may be a BNMTF handle -- an S block of a wider tri-factorisation; the others are two-factor products)
  HIPCHK(hipSetDevice(h->device));
  CHK(ensure_std(h));
  ResidualSpec rs;
  memset(&rs, 0, sizeof(rs));
  rs.n = n_others;
  for (int b = 0; b < n_others; ++b) {
    bnmtf_model* o = others[b];
    if (!o || o == h || o->I != h->I || o->J != h->J || o->L != 0 || o->device != h->device || !o->have_state) { set_error("residual data: block %d does not fit (shape, device, state)", b); return BNMTF_EINVAL; }
    CHK(ensure_std(o));
    if (!o->std_cur && !o->vb_ready) { set_error("residual data: block %d's state is not on its multi-launch structures", b); return BNMTF_ESTATE; }
    HIPCHK(hipStreamSynchronize(o->stream));                 // (its last half sweep runs on its own stream)
    rs.A[b] = o->rows.X; rs.B[b] = o->cols.X; rs.KP[b] = o->rows.KP; rs.W[b] = o->rows.W;
  }
  for (int which = 0; which < 2; ++which) {
    Dir& d = which == 0 ? h->rows : h->cols;
    launch_residual_operand(h->Rfull, h->Mtrain, h->I, h->J, which == 0 ? 1 : 0, d.n0, d.n, d.m, d.big, d.n_pad, rs, h->stream);
  }
  HIPCHK(hipGetLastError());
  return BNMTF_OK;
} BNMTF_ABI_GUARD

// one half of an iteration of run(): the contraction, the K sequential column updates of one factor (which = 0: U, :134-137; 1: V,
// :139-142) with the handle's current tau and iteration counter, the relayout + Gram the other direction reads.  tau, the
// metrics, the samples and the iteration counter are the caller's (a column-blocked model: bnmtf_amd/_blocked.py).
int bnmf_half_sweep(bnmtf_handle h, int which, int update) try {
  if (which < 0 || which > 1 || update < 0 || update > BNMTF_UPDATE_ICM) { set_error("bnmf_half_sweep: which in {0, 1}, a known update rule"); return BNMTF_EINVAL; }
  if (h->L != 0 || h->comm) { set_error("bnmf_half_sweep: BNMF handles on one GPU"); return BNMTF_ESTATE; }
  if (!h->have_state) { set_error("no state set"); return BNMTF_ESTATE; }
  HIPCHK(hipSetDevice(h->device));
  CHK(ensure_std(h));
  if (!h->std_cur) { set_error("bnmf_half_sweep: set the state first"); return BNMTF_ESTATE; }
  h->small_cur = false;
  const int mode = update == BNMTF_UPDATE_DRAW ? kSweepDraw : kSweepMode;
  h->cur_min_x = update == BNMTF_UPDATE_ICM ? (float)h->min_tn : 0.f;
  Dir& d = which == 0 ? h->rows : h->cols;
  Dir& o = which == 0 ? h->cols : h->rows;
  h->ho_active = false;
  h->rows.ho_filled = h->cols.ho_filled = false;
  enqueue_gemm(h, d, o, which == 0 ? BNMTF_KERNEL_GEMM_ROWS : BNMTF_KERNEL_GEMM_COLS);
  SweepArgs s = sweep_args(h, d, o, mode, which == 0 ? kStreamRows : kStreamCols);
  enqueue_sweep(h, d, o, s, false);
  CHK(exchange_factor(h, d));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  return BNMTF_OK;
} BNMTF_ABI_GUARD

// ---------------------------------------------------------------------- metrics
static int metric_sums_impl(bnmtf_handle h, const uint8_t* Mp, const double* A, const double* S, const double* B, double sums_out[6], int Kc_given);
int bnmtf_metric_sums(bnmtf_handle h, const uint8_t* Mp, const double* A, const double* S, const double* B,
                      double sums_out[6]) { return metric_sums_impl(h, Mp, A, S, B, sums_out, 0); }
int bnmtf_metric_sums_wide(bnmtf_handle h, const uint8_t* Mp, const double* A, const double* B, int Kc, double sums_out[6]) try {
  if (!A || !B || Kc <= 0) { set_error("bnmtf_metric_sums_wide: A [I][Kc], B [J][Kc] and Kc > 0 required"); return BNMTF_EINVAL; }
  if (h->L != 0) { set_error("bnmtf_metric_sums_wide on a BNMTF handle"); return BNMTF_ESTATE; }
  return metric_sums_impl(h, Mp, A, nullptr, B, sums_out, Kc);
} BNMTF_ABI_GUARD
static int metric_sums_impl(bnmtf_handle h, const uint8_t* Mp, const double* A, const double* S, const double* B, double sums_out[6], int Kc_given) {
  HIPCHK(hipSetDevice(h->device));
  const int I = h->I, J = h->J;
  std::vector<double> a_own, b_own, as;
  int Kc;  // contraction width of the two-factor product handed to the kernel
  if (!A) {
    if (!h->have_state) { set_error("no state set"); return BNMTF_ESTATE; }
    a_own.resize((size_t)I * h->rows.W); b_own.resize((size_t)J * h->cols.W);
    const bool from_small = h->small && h->small_cur;
    if (h->L > 0) as.resize((size_t)h->K * h->L);
    if (from_small) CHK(small_download_state(h, a_own.data(), b_own.data(), h->L > 0 ? as.data() : nullptr));
    else {
      CHK(download_matrix(h, h->rows.X, I, h->rows.W, h->rows.KP, a_own.data()));
      CHK(download_matrix(h, h->cols.X, J, h->cols.W, h->cols.KP, b_own.data()));
    }
    A = a_own.data(); B = b_own.data();
    if (h->L > 0) {
      if (!from_small) {
        std::vector<float> sf((size_t)h->K * h->L);
        HIPCHK(hipMemcpy(sf.data(), h->S, sf.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t t = 0; t < sf.size(); ++t) as[t] = sf[t];
      }
      S = as.data();
    }
  }
  std::vector<double> AS;
  if (S) {   // A.S  (I x L), then a plain two-factor product with B (J x L)
    const int K = h->K, L = h->L;
    AS.assign((size_t)I * L, 0.0);
    for (int i = 0; i < I; ++i)
      for (int k = 0; k < K; ++k) {
        const double aik = A[(size_t)i * K + k];
        for (int l = 0; l < L; ++l) AS[(size_t)i * L + l] += aik * S[(size_t)k * L + l];
      }
    A = AS.data(); Kc = L;
  } else {
    Kc = h->L > 0 ? h->L : h->K;
    if (h->L > 0) { set_error("bnmtf_metric_sums: S required for a BNMTF handle"); return BNMTF_EINVAL; }
  }
  if (A && !S && h->L == 0 && Kc_given > 0) Kc = Kc_given;       // (a column-blocked factorisation hands over all its columns: bnmtf_metric_sums_wide)
  // operand copies: the handle's [I][64] / [J][64] buffers (part of a small model's arena: never freed here); factors wider than
  // 64 columns (a column-blocked model) get buffers of their own
  double* Ad = h->Ad; double* Bd = h->Bd;
  if (Kc > 64) {
    if (h->ABd_width < Kc) {
      dfree(h->AdW); dfree(h->BdW); h->AdW = nullptr; h->BdW = nullptr;
      CHK(dalloc(&h->AdW, (size_t)I * Kc, false)); CHK(dalloc(&h->BdW, (size_t)J * Kc, false));
      h->ABd_width = Kc;
    }
    Ad = h->AdW; Bd = h->BdW;
  } else if (!h->Ad) { CHK(dalloc(&h->Ad, (size_t)I * 64, false)); CHK(dalloc(&h->Bd, (size_t)J * 64, false)); Ad = h->Ad; Bd = h->Bd; }
  HIPCHK(hipMemcpyAsync(Ad, A, sizeof(double) * (size_t)I * Kc, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(Bd, B, sizeof(double) * (size_t)J * Kc, hipMemcpyHostToDevice, h->stream));
  const uint8_t* mask = h->Mtrain;
  if (Mp) {
    if (!h->Mscratch) CHK(dalloc(&h->Mscratch, (size_t)I * J, false));
    HIPCHK(hipMemcpyAsync(h->Mscratch, Mp, (size_t)I * J, hipMemcpyHostToDevice, h->stream));
    mask = h->Mscratch;
  }
  MetricArgs m;
  m.R = h->Rfull; m.Mp = mask; m.I = I; m.J = J; m.A = Ad; m.B = Bd; m.K = Kc; m.out6 = h->out6; m.A2 = nullptr; m.B2 = nullptr;
  launch_metric_sums(m, h->stream);
  HIPCHK(hipMemcpyAsync(sums_out, h->out6, 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  return BNMTF_OK;
}

int bnmtf_beta_s(bnmtf_handle h, double* out) try {
  double s[6];
  CHK(bnmtf_metric_sums(h, nullptr, nullptr, nullptr, nullptr, s));
  *out = h->beta + 0.5 * (s[2] - 2.0 * s[5] + s[4]);
  return BNMTF_OK;
} BNMTF_ABI_GUARD

// ---------------------------------------------------------------- distributions
int bnmtf_tn_sample(const double* mu, const double* tau, size_t n, uint64_t seed, uint64_t it, uint32_t col,
                    uint32_t elem0, int device, double* out) try {
  if (n == 0) return BNMTF_OK;
  HIPCHK(hipSetDevice(device));
  DevBuf<double> dm, dt, dout;
  CHK(dm.alloc(n)); CHK(dt.alloc(n)); CHK(dout.alloc(n));
  HIPCHK(hipMemcpy(dm.p, mu, n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dt.p, tau, n * sizeof(double), hipMemcpyHostToDevice));
  launch_tn_sample(dm.p, dt.p, n, seed, (uint32_t)it, col, elem0, dout.p, nullptr);
  HIPCHK(hipMemcpy(out, dout.p, n * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipGetLastError());
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmtf_tn_moments(const double* mu, const double* tau, size_t n, int device, double* exp_out, double* var_out) try {
  if (n == 0) return BNMTF_OK;
  HIPCHK(hipSetDevice(device));
  DevBuf<double> dm, dt, de, dv;
  CHK(dm.alloc(n)); CHK(dt.alloc(n)); CHK(de.alloc(n)); CHK(dv.alloc(n));
  HIPCHK(hipMemcpy(dm.p, mu, n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dt.p, tau, n * sizeof(double), hipMemcpyHostToDevice));
  launch_tn_moments(dm.p, dt.p, n, de.p, dv.p, nullptr);
  HIPCHK(hipMemcpy(exp_out, de.p, n * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(var_out, dv.p, n * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipGetLastError());
  return BNMTF_OK;
} BNMTF_ABI_GUARD

int bnmtf_gamma_sample(double alpha, double beta, uint64_t seed, uint64_t it, int device, double* out) try {
  HIPCHK(hipSetDevice(device));
  DevBuf<double> d;
  CHK(d.alloc(1, true));
  launch_gamma_sample(alpha, beta, seed, (uint32_t)it, d.p, nullptr);
  HIPCHK(hipMemcpy(out, d.p, sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipGetLastError());
  return BNMTF_OK;
} BNMTF_ABI_GUARD

}  // extern "C"

#include "api_models.inc"
#include "api_trivb.inc"
#include "api_many.inc"
