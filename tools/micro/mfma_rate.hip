// Issue rate of v_mfma_f32_32x32x2_f32 (and the bf16 32x32x16) on one SIMD: N dependent / independent MFMAs per wave,
// W waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int BF>
__global__ __launch_bounds__(256) void k(float* out, int n, float seed) {
  f32x16 acc[NACC];
  for (int x = 0; x < NACC; ++x) for (int t = 0; t < 16; ++t) acc[x][t] = 0.f;
  float a = seed + threadIdx.x;
  bf16x8 b8; for (int t = 0; t < 8; ++t) b8[t] = (__bf16)a;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int x = 0; x < NACC; ++x) {
      if (BF) acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8, b8, acc[x], 0, 0, 0);
      else    acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[x], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int x = 0; x < NACC; ++x) for (int t = 0; t < 16; ++t) s += acc[x][t];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the A operand made by vector instructions right before each MFMA (as in a kernel that scales its operand)
template <int NACC, int NV>
__global__ __launch_bounds__(256) void kv(float* out, int n, float seed) {
  f32x16 acc[NACC];
  for (int x = 0; x < NACC; ++x) for (int t = 0; t < 16; ++t) acc[x][t] = 0.f;
  float a = seed + threadIdx.x, w = seed * 0.5f;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int x = 0; x < NACC; ++x) {
      float op = a;
#pragma unroll
      for (int v = 0; v < NV; ++v) op = fmaf(op, w, (float)(i + x + v));
      acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(op, a, acc[x], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int x = 0; x < NACC; ++x) for (int t = 0; t < 16; ++t) s += acc[x][t];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int NV>
void runv(int blocks_per_cu, int n) {
  float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL((kv<NACC, NV>), dim3(blocks), dim3(256), 0, 0, out, n, 1.f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((kv<NACC, NV>), dim3(blocks), dim3(256), 0, 0, out, n, 1.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)blocks_per_cu * n * NACC;
  printf("f32 32x32x2 + %d VALU per MFMA, acc=%d waves/SIMD=%d: %.1f us, %.1f ns per MFMA per SIMD\n", NV, NACC, blocks_per_cu, ms * 1e3, ms * 1e6 / mfma_per_simd);
  (void)hipFree(out);
}
template <int NACC, int BF>
void run(int blocks_per_cu, int n) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL((k<NACC, BF>), dim3(blocks), dim3(256), 0, 0, out, n, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, BF>), dim3(blocks), dim3(256), 0, 0, out, n, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)blocks_per_cu * n * NACC;
  printf("%s acc=%d waves/SIMD=%d: %.1f us, %.1f ns per MFMA per SIMD (= %.0f cycles at 2.4 GHz)\n", BF ? "bf16 32x32x16" : "f32 32x32x2 ", NACC, blocks_per_cu, ms * 1e3,
         ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4);
  hipFree(out);
}
int main() {
  run<1, 0>(1, 4000); run<2, 0>(1, 2000); run<4, 0>(1, 1000); run<1, 0>(4, 1000); run<2, 0>(4, 500);
  run<1, 1>(1, 4000); run<4, 1>(1, 1000); run<2, 1>(4, 500);
  runv<4, 1>(1, 1000); runv<4, 2>(1, 1000); runv<4, 2>(2, 1000); runv<4, 4>(2, 1000); runv<4, 8>(2, 500); runv<4, 2>(4, 500);
  return 0;
}
