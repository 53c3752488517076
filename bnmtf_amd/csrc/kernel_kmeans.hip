// Masked K-means for initialise(init_FG='kmeans') (code/models/kmeans/kmeans.py:70-204): the two O(points x coordinates x K)
// passes of an iteration on the device.
//   assignment (:87-119): MSE between a point and a centroid over the coordinates both know; no overlap = infinitely far;
//     ties go to the lowest cluster index.  One wave per point, lanes stride the coordinates; points, centroids and sums
//     in fp64 like the reference (an fp32 copy of the data can flip a near-tie between two centroids).
//   update (:126-163, find_known_coordinate_values :170-182): per cluster and coordinate the count and the sum of the
//     members' observed values.  The O(K x coordinates) division, the masks and the empty-cluster rule ('singleton':
//     the point furthest from its centroid moves) stay on the host (bnmtf_amd/kmeans.py).
#include <atomic>

#include "model.h"

namespace bnmtf {

__global__ __launch_bounds__(256) void kmeans_assign_kernel(const double* X, const uint8_t* M, int n, int d, int K, const double* C,
                                                             const uint8_t* Mc, int* assign, double* dist) {
  const int lane = threadIdx.x & 63, p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  const double* x = X + (size_t)p * d;
  const uint8_t* m = M + (size_t)p * d;
  int best = -1;
  double best_mse = 0.0;
  bool have = false;                                   // a finite MSE has been seen
  for (int c = 0; c < K; ++c) {
    double num = 0.0, ov = 0.0;
    for (int j = lane; j < d; j += 64) {
      if (m[j] && Mc[(size_t)c * d + j]) {
        const double df = x[j] - C[(size_t)c * d + j];
        num = fma(df, df, num); ov += 1.0;
      }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { num += __shfl_xor(num, s, 64); ov += __shfl_xor(ov, s, 64); }
    // kmeans.py:107-113: while no defined MSE has been seen every cluster replaces the choice; after that only a smaller defined MSE does
    const bool defined = ov > 0.0;
    const double mse = defined ? num / ov : 0.0;
    if (!have) { best = c; have = defined; best_mse = mse; }          // (so a point that overlaps no centroid ends in the last cluster)
    else if (defined && mse < best_mse) { best = c; best_mse = mse; }
  }
  if (lane == 0) { assign[p] = best; dist[p] = have ? best_mse : __longlong_as_double(0x7ff0000000000000LL); }
}

// cnt[c][j] = #members of c that observe coordinate j, tot[c][j] = sum of their values, for the clusters c0 .. c0 + KC - 1:
// block = 64 coordinates x 4 point groups, private LDS accumulators per group (no atomics), groups summed in order
constexpr int kKMeansChunk = 40;                        // clusters per launch (8 x 40 x 64 doubles of LDS)
__global__ __launch_bounds__(256) void kmeans_sums_kernel(const double* X, const uint8_t* M, int n, int d, int c0, int KC, const int* assign,
                                                           double* cnt, double* tot) {
  extern __shared__ double acc[];                       // [4][KC][64] counts, then [4][KC][64] totals
  const int jl = threadIdx.x & 63, g = threadIdx.x >> 6, j = blockIdx.x * 64 + jl;
  double* ac = acc + (size_t)g * KC * 64;
  double* at = acc + (size_t)(4 + g) * KC * 64;
  for (int c = 0; c < KC; ++c) { ac[c * 64 + jl] = 0.0; at[c * 64 + jl] = 0.0; }
  if (j < d)
    for (int p = g; p < n; p += 4) {
      const int c = assign[p] - c0;
      if (c >= 0 && c < KC && M[(size_t)p * d + j]) { ac[c * 64 + jl] += 1.0; at[c * 64 + jl] += X[(size_t)p * d + j]; }
    }
  __syncthreads();
  if (j < d)
    for (int c = g; c < KC; c += 4) {
      double sc = 0.0, st = 0.0;
      for (int q = 0; q < 4; ++q) { sc += acc[((size_t)q * KC + c) * 64 + jl]; st += acc[((size_t)(4 + q) * KC + c) * 64 + jl]; }
      cnt[(size_t)(c0 + c) * d + j] = sc; tot[(size_t)(c0 + c) * d + j] = st;
    }
}

struct KMeansModel {
  int n = 0, d = 0, K = 0, device = 0;
  double* X = nullptr; uint8_t* M = nullptr; double* C = nullptr; uint8_t* Mc = nullptr; int* assign = nullptr;
  double *dist = nullptr, *cnt = nullptr, *tot = nullptr;
};

}  // namespace bnmtf

using namespace bnmtf;

extern "C" {

int bnmtf_kmeans_destroy(void* hv) {
  KMeansModel* h = static_cast<KMeansModel*>(hv);
  if (!h) return BNMTF_OK;
  (void)hipSetDevice(h->device);
  (void)hipFree(h->X); (void)hipFree(h->M); (void)hipFree(h->C); (void)hipFree(h->Mc); (void)hipFree(h->assign);
  (void)hipFree(h->dist); (void)hipFree(h->cnt); (void)hipFree(h->tot);
  delete h;
  return BNMTF_OK;
}

int bnmtf_kmeans_create(const double* X, const uint8_t* M, int n_points, int n_coords, int K, int device, void** out) {
  *out = nullptr;
  if (!X || !M || n_points < 1 || n_coords < 1 || K < 1 || K > 1024) { set_error("bnmtf_kmeans_create: bad argument"); return BNMTF_EINVAL; }
  HIPCHK(hipSetDevice(device));
  KMeansModel* h = new KMeansModel();
  h->n = n_points; h->d = n_coords; h->K = K; h->device = device;
  const size_t nd = (size_t)n_points * n_coords, kd = (size_t)K * n_coords;
  auto fail = [&](hipError_t e) { set_error("bnmtf_kmeans_create: %s", hipGetErrorString(e)); bnmtf_kmeans_destroy(h); return BNMTF_EHIP; };
  hipError_t e;
  if ((e = hipMalloc((void**)&h->X, nd * sizeof(double))) != hipSuccess) return fail(e);
  if ((e = hipMalloc((void**)&h->M, nd)) != hipSuccess) return fail(e);
  if ((e = hipMalloc((void**)&h->C, kd * sizeof(double))) != hipSuccess) return fail(e);
  if ((e = hipMalloc((void**)&h->Mc, kd)) != hipSuccess) return fail(e);
  if ((e = hipMalloc((void**)&h->assign, (size_t)n_points * sizeof(int))) != hipSuccess) return fail(e);
  if ((e = hipMalloc((void**)&h->dist, (size_t)n_points * sizeof(double))) != hipSuccess) return fail(e);
  if ((e = hipMalloc((void**)&h->cnt, kd * sizeof(double))) != hipSuccess) return fail(e);
  if ((e = hipMalloc((void**)&h->tot, kd * sizeof(double))) != hipSuccess) return fail(e);
  if ((e = hipMemcpy(h->X, X, nd * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
  if ((e = hipMemcpy(h->M, M, nd, hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
  *out = h;
  return BNMTF_OK;
}

// assignment() (kmeans.py:87-119) for the given centroids [K][n_coords] and their masks; assign_out [n_points],
// dist_out [n_points] (MSE to the chosen centroid; +inf when it shares no coordinate with the point)
int bnmtf_kmeans_assign(void* hv, const double* centroids, const uint8_t* mask_centroids, int32_t* assign_out, double* dist_out) {
  KMeansModel* h = static_cast<KMeansModel*>(hv);
  HIPCHK(hipSetDevice(h->device));
  const size_t kd = (size_t)h->K * h->d;
  HIPCHK(hipMemcpy(h->C, centroids, kd * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->Mc, mask_centroids, kd, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(kmeans_assign_kernel, dim3((h->n + 3) / 4), dim3(256), 0, nullptr, h->X, h->M, h->n, h->d, h->K, h->C, h->Mc, h->assign, h->dist);
  HIPCHK(hipMemcpy(assign_out, h->assign, (size_t)h->n * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(dist_out, h->dist, (size_t)h->n * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipGetLastError());
  return BNMTF_OK;
}

// row `index` of X := values [n_coords] (the reference's refilled centroid is a view of its data point, kmeans.py:141: the
// means later written into that centroid change the point)
int bnmtf_kmeans_set_row(void* hv, int index, const double* values) {
  KMeansModel* h = static_cast<KMeansModel*>(hv);
  if (!h || !values || index < 0 || index >= h->n) { set_error("bnmtf_kmeans_set_row: bad argument"); return BNMTF_EINVAL; }
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpy(h->X + (size_t)index * h->d, values, (size_t)h->d * sizeof(double), hipMemcpyHostToDevice));
  return BNMTF_OK;
}

// per cluster and coordinate: the number of members observing it and the sum of their values (update(), kmeans.py:126-182)
int bnmtf_kmeans_sums(void* hv, const int32_t* assign, double* cnt_out, double* tot_out) {
  KMeansModel* h = static_cast<KMeansModel*>(hv);
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpy(h->assign, assign, (size_t)h->n * sizeof(int), hipMemcpyHostToDevice));
  static std::atomic<uint64_t> lds_ok{0};
  int dev = h->device & 63;
  if (!(lds_ok.load() & (1ull << dev))) {
    HIPCHK(hipFuncSetAttribute((const void*)kmeans_sums_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    lds_ok.fetch_or(1ull << dev);
  }
  for (int c0 = 0; c0 < h->K; c0 += kKMeansChunk) {      // any K: kKMeansChunk clusters per pass over the points
    const int kc = h->K - c0 < kKMeansChunk ? h->K - c0 : kKMeansChunk;
    hipLaunchKernelGGL(kmeans_sums_kernel, dim3((h->d + 63) / 64), dim3(256), (size_t)8 * kc * 64 * sizeof(double), nullptr,
                       h->X, h->M, h->n, h->d, c0, kc, h->assign, h->cnt, h->tot);
  }
  const size_t kd = (size_t)h->K * h->d;
  HIPCHK(hipMemcpy(cnt_out, h->cnt, kd * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(tot_out, h->tot, kd * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipGetLastError());
  return BNMTF_OK;
}

}  // extern "C"
