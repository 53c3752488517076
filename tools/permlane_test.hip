// Probe of v_permlane16_swap_b32 semantics on gfx950 (build: hipcc --offload-arch=gfx950 -o permlane_test permlane_test.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* a_out, unsigned* b_out) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a_out[threadIdx.x] = r.x; b_out[threadIdx.x] = r.y;
  u2 r2 = __builtin_amdgcn_permlane16_swap(a, a, false, false);   // same value in both operands
  a_out[64 + threadIdx.x] = r2.x; b_out[64 + threadIdx.x] = r2.y;
}
int main() {
  unsigned *a, *b, ha[128], hb[128];
  hipMalloc(&a, 512); hipMalloc(&b, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b);
  hipMemcpy(ha, a, 512, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i += 8) printf("lane %2d: x=%3u y=%3u | same-operand: x=%3u y=%3u\n", i, ha[i], hb[i], ha[64 + i], hb[64 + i]);
  return 0;
}
