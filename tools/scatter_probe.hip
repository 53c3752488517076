// Probe: cost of carrying q across half sweeps by scattered 4-byte stores (row-slot order -> column-slot order).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/scatter_probe tools/scatter_probe.hip && /tmp/scatter_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>

__global__ void scatter(const uint32_t* __restrict__ idx, const float* __restrict__ src, float* __restrict__ dst, int rows) {
  // one half-wave-like unit per 64 threads: EM rows of 64 lanes, like the sweep's slot arrays
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int r = 0; r < rows; ++r) {
    const size_t p = ((size_t)(t >> 6) * rows + r) * 64 + (t & 63);
    const uint32_t d = idx[p];
    if (d != 0xFFFFFFFFu) dst[d] = src[p];
  }
}
__global__ void gather(const uint32_t* __restrict__ idx, const float* __restrict__ src, float* __restrict__ dst, int rows) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int r = 0; r < rows; ++r) {
    const size_t p = ((size_t)(t >> 6) * rows + r) * 64 + (t & 63);
    const uint32_t d = idx[p];
    dst[p] = d != 0xFFFFFFFFu ? src[d] : 0.f;
  }
}

int main() {
  const int n = 8192, rows = 40;          // 4096 waves x 40 rows x 64 lanes
  const size_t slots = (size_t)(n / 2) * rows * 64;
  // transposition-like map: entry (i, j) with 10% density; row layout position p(i, .) -> column layout position of (j, i)
  std::mt19937 rng(1);
  std::vector<std::vector<uint32_t>> rowsE(n), colsE(n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) if ((rng() % 10) == 0) { rowsE[i].push_back(j); colsE[j].push_back(i); }
  auto layout = [&](std::vector<std::vector<uint32_t>>& E, std::vector<int64_t>& pos_of) {
    // unit u -> pair u/2, half u%2; entry with inner index v goes to lane (v%32) + 32*half, next free row (capped)
    pos_of.assign((size_t)n * n, -1);
    for (int u = 0; u < n; ++u) {
      int cnt[32] = {0};
      for (uint32_t v : E[u]) {
        const int l = v % 32;
        if (cnt[l] >= rows) continue;
        pos_of[(size_t)u * n + v] = ((int64_t)(u / 2) * rows + cnt[l]) * 64 + l + 32 * (u % 2);
        ++cnt[l];
      }
    }
  };
  std::vector<int64_t> prow, pcol;
  layout(rowsE, prow); layout(colsE, pcol);
  std::vector<uint32_t> idx(slots, 0xFFFFFFFFu);
  size_t live = 0;
  for (int i = 0; i < n; ++i) for (uint32_t j : rowsE[i]) {
    const int64_t a = prow[(size_t)i * n + j], b = pcol[(size_t)j * n + i];
    if (a >= 0 && b >= 0) { idx[a] = (uint32_t)b; ++live; }
  }
  printf("slots %zu live %zu\n", slots, live);
  uint32_t* d_idx; float *d_src, *d_dst;
  hipMalloc(&d_idx, slots * 4); hipMalloc(&d_src, slots * 4); hipMalloc(&d_dst, slots * 4);
  hipMemcpy(d_idx, idx.data(), slots * 4, hipMemcpyHostToDevice);
  hipMemset(d_src, 0, slots * 4); hipMemset(d_dst, 0, slots * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 256, blocks = (n / 2) * 64 / threads;
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < 10; ++it) {
        if (which == 0) hipLaunchKernelGGL(scatter, dim3(blocks), dim3(threads), 0, 0, d_idx, d_src, d_dst, rows);
        else hipLaunchKernelGGL(gather, dim3(blocks), dim3(threads), 0, 0, d_idx, d_src, d_dst, rows);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %.1f us per pass\n", which == 0 ? "scatter" : "gather", ms * 100.f);
    }
  }
  return 0;
}
