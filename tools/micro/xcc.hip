#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID, 4 bits at offset 0
    unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);  // HW_ID low 16 bits
    out[blockIdx.x] = x | (hw << 8);
  }
}
int main() {
  unsigned* d; hipMalloc(&d, 4096 * 4);
  for (int nb : {1, 1, 1, 16, 40}) {
    hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, d); hipDeviceSynchronize();
    unsigned h[64]; hipMemcpy(h, d, nb * 4, hipMemcpyDeviceToHost);
    printf("%d blocks: xcc ", nb); for (int i = 0; i < nb; ++i) printf("%u ", h[i] & 15); printf(" | cu(hw_id[11:8]) "); for (int i = 0; i < nb && i < 16; ++i) printf("%u ", (h[i] >> 16) & 15); printf("\n");
  }
  return 0;
}
