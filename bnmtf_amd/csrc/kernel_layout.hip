// bnmtf_create's passes over the I x J data, on the device (the host did them with 32 threads in ~200 ms at 8192^2; R and M are
// uploaded anyway -- predict() needs them):
//   * the contraction operand of a direction, big[r][ul] = M ? R : 0 for unit ul (a row of R for the rows direction, a column for
//     the cols direction) and inner index r -- i.e. the masked matrix, transposed for the rows direction;
//   * a unit's missing inner indices in ascending order, 64-wide slots padded with the sentinel m (Dir::slot_ptr / Dir::idx).
#include "kernels.h"

namespace bnmtf {

// out[r][ul], r < m, ul < n.  rows direction (by_rows = 1): unit = row unit0 + ul of R, r = column: a 32 x 32 tile is transposed
// through LDS, reads and writes both coalesced.  cols direction: unit = column unit0 + ul, r = row: a masked copy.
__global__ __launch_bounds__(256) void masked_operand_kernel(const float* R, const uint8_t* M, int I, int J, int by_rows, int unit0, int n, int m,
                                                             float* out, int ld) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  if (by_rows) {
    const int u0 = blockIdx.y * 32, r0 = blockIdx.x * 32;           // units (rows of R) x inner (columns of R)
    for (int k = ty; k < 32; k += 8) {
      const int ul = u0 + k, r = r0 + tx;
      float v = 0.f;
      if (ul < n && r < m) { const size_t e = (size_t)(unit0 + ul) * J + r; v = M[e] ? R[e] : 0.f; }
      tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      const int r = r0 + k, ul = u0 + tx;
      if (r < m && ul < n) out[(size_t)r * ld + ul] = tile[tx][k];
    }
  } else {
    const int ul = blockIdx.y * 32 + tx;
    for (int k = ty; k < 32; k += 8) {
      const int r = blockIdx.x * 32 + k;
      if (ul < n && r < m) { const size_t e = (size_t)r * J + unit0 + ul; out[(size_t)r * ld + ul] = M[e] ? R[e] : 0.f; }
    }
  }
  (void)I;
}

// The residual operand of a column block (ranks above 64: bnmf_set_residual_data): as masked_operand_kernel, with
// sum_b A_b[i] . B_b[j] taken off every observed entry first (i = row of R, j = column of R; fp32 dot products in column order).
// The factors' rows of a 32 x 32 tile sit in LDS; not a tuned kernel -- the path it serves runs several launches per half sweep.
__global__ __launch_bounds__(256) void residual_operand_kernel(const float* R, const uint8_t* M, int I, int J, int by_rows, int unit0, int n, int m,
                                                               float* out, int ld, ResidualSpec rs) {
  __shared__ float tile[32][33];
  __shared__ float At[32][65], Bt[32][65];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  // tile of R: rows i0 .. i0+31, columns j0 .. j0+31
  const int i0 = by_rows ? unit0 + blockIdx.y * 32 : blockIdx.x * 32;
  const int j0 = by_rows ? blockIdx.x * 32 : unit0 + blockIdx.y * 32;
  float pred[4] = {0.f, 0.f, 0.f, 0.f};                             // element (i0 + ty + 8 t, j0 + tx)
  for (int b = 0; b < rs.n; ++b) {
    __syncthreads();
    for (int t = threadIdx.x; t < 32 * 64; t += 256) {
      const int r = t >> 6, k = t & 63;
      At[r][k] = (k < rs.W[b] && i0 + r < I) ? rs.A[b][(size_t)(i0 + r) * rs.KP[b] + k] : 0.f;
      Bt[r][k] = (k < rs.W[b] && j0 + r < J) ? rs.B[b][(size_t)(j0 + r) * rs.KP[b] + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float acc = pred[t];
      for (int k = 0; k < rs.W[b]; ++k) acc = fmaf(At[ty + 8 * t][k], Bt[tx][k], acc);
      pred[t] = acc;
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = i0 + ty + 8 * t, j = j0 + tx;
    float v = 0.f;
    if (i < I && j < J) { const size_t e = (size_t)i * J + j; v = M[e] ? R[e] - pred[t] : 0.f; }
    tile[ty + 8 * t][tx] = v;                                       // tile[row of R][column of R]
  }
  __syncthreads();
  if (by_rows) {                                                    // out[r = column of R][ul = row of R - unit0]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = blockIdx.x * 32 + ty + 8 * t, ul = blockIdx.y * 32 + tx;
      if (r < m && ul < n) out[(size_t)r * ld + ul] = tile[tx][ty + 8 * t];
    }
  } else {                                                          // out[r = row of R][ul = column of R - unit0]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = blockIdx.x * 32 + ty + 8 * t, ul = blockIdx.y * 32 + tx;
      if (r < m && ul < n) out[(size_t)r * ld + ul] = tile[ty + 8 * t][tx];
    }
  }
}

// one wave per unit: the inner indices with M = 0, in order, behind ptr[ul]; the rest of the unit's 64-wide slots gets m
__global__ __launch_bounds__(256) void missing_lists_kernel(const uint8_t* M, int I, int J, int by_rows, int unit0, int n, int m,
                                                            const uint32_t* ptr, uint32_t* idx) {
  const int lane = threadIdx.x & 63;
  const int ul = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ul >= n) return;
  const int u = unit0 + ul;
  uint32_t pos = ptr[ul];
  const uint32_t end = ptr[ul + 1];
  for (int r0 = 0; r0 < m; r0 += 64) {
    const int r = r0 + lane;
    const bool miss = r < m && (by_rows ? M[(size_t)u * J + r] : M[(size_t)r * J + u]) == 0;
    const unsigned long long b = __ballot(miss);
    if (miss) idx[pos + __popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)r;
    pos += (uint32_t)__popcll(b);
  }
  for (uint32_t p = pos + lane; p < end; p += 64) idx[p] = (uint32_t)m;
  (void)I;
}

void launch_masked_operand(const float* R, const uint8_t* M, int I, int J, int by_rows, int unit0, int n, int m, float* out, int ld, hipStream_t st) {
  if (n <= 0 || m <= 0) return;
  hipLaunchKernelGGL(masked_operand_kernel, dim3((m + 31) / 32, (n + 31) / 32), dim3(256), 0, st, R, M, I, J, by_rows, unit0, n, m, out, ld);
}
void launch_residual_operand(const float* R, const uint8_t* M, int I, int J, int by_rows, int unit0, int n, int m, float* out, int ld,
                             const ResidualSpec& rs, hipStream_t st) {
  if (n <= 0 || m <= 0) return;
  hipLaunchKernelGGL(residual_operand_kernel, dim3((m + 31) / 32, (n + 31) / 32), dim3(256), 0, st, R, M, I, J, by_rows, unit0, n, m, out, ld, rs);
}
void launch_missing_lists(const uint8_t* M, int I, int J, int by_rows, int unit0, int n, int m, const uint32_t* ptr, uint32_t* idx, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(missing_lists_kernel, dim3((n + 3) / 4), dim3(256), 0, st, M, I, J, by_rows, unit0, n, m, ptr, idx);
}

}  // namespace bnmtf
