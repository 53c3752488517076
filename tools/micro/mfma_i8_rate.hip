// Issue rate of v_mfma_i32_32x32x32_i8 against v_mfma_f32_32x32x16_bf16 on one SIMD (one wave per SIMD, four independent tiles):
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_i8_rate.hip -o /tmp/mfma_i8_rate && /tmp/mfma_i8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int I8>
__global__ __launch_bounds__(256) void k(int* out, int n, int seed) {
  i32x16 ai[4]; f32x16 af[4];
  for (int x = 0; x < 4; ++x) for (int t = 0; t < 16; ++t) { ai[x][t] = 0; af[x][t] = 0.f; }
  i32x4 a4 = {seed + (int)threadIdx.x, seed, 3, 1};
  bf16x8 b8; for (int t = 0; t < 8; ++t) b8[t] = (__bf16)(float)(seed + threadIdx.x);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      if (I8) ai[x] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, a4, ai[x], 0, 0, 0);
      else    af[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8, b8, af[x], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  int s = 0;
  for (int x = 0; x < 4; ++x) for (int t = 0; t < 16; ++t) s += ai[x][t] + (int)af[x][t];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) printf("%s: %.1f shader cycles per MFMA (one wave per SIMD)\n", I8 ? "i32_32x32x32_i8 " : "f32_32x32x16_bf16", (double)(t1 - t0) / (4.0 * n));
}
int main() {
  int* out; (void)hipMalloc(&out, 256 * 256 * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, 2000, 1); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, 2000, 1); (void)hipDeviceSynchronize();
  }
  return 0;
}
