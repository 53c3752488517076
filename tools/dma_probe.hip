// Probe: rate at which one CU can stage L2-resident panels into LDS (LDS-DMA vs load + ds_write), with every CU
// of the chip staging the same panels at once, as the sweep's pre-pass does.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_probe tools/dma_probe.hip && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void stage(const float* __restrict__ src, int panel_floats, int npanels, int steps, float* out,
                                                  unsigned long long* cyc) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunks = panel_floats / 256;
  float acc = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) {
    const float* p = src + (size_t)(s % npanels) * panel_floats;
    float* d = lds + (size_t)(s & 1) * panel_floats;
    if (MODE == 0) {
      for (int c = wave; c < chunks; c += NW)
        __builtin_amdgcn_global_load_lds(p + (size_t)c * 256 + lane * 4, (lds_ptr)(d + (size_t)c * 256), 16, 0, 0);
    } else if (MODE == 1) {
      for (int c = wave; c < chunks; c += NW) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + (size_t)c * 256 + lane * 4);
        *reinterpret_cast<f32x4*>(d + (size_t)c * 256 + lane * 4) = v;
      }
    } else {   // dword-wide LDS-DMA
      for (int c = wave; c < chunks * 4; c += NW)
        __builtin_amdgcn_global_load_lds(p + (size_t)c * 64 + lane, (lds_ptr)(d + (size_t)c * 64), 4, 0, 0);
    }
    __syncthreads();
    acc += d[(threadIdx.x * 17 + s) % panel_floats];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int NW>
void run(const char* name, const float* src, int panel_floats, int npanels, int blocks, float* out, unsigned long long* cyc) {
  const int steps = 256;
  const size_t lds = (size_t)2 * panel_floats * 4;
  hipFuncSetAttribute((const void*)stage<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stage<MODE, NW>), dim3(blocks), dim3(NW * 64), lds, 0, src, panel_floats, npanels, steps, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if (rep == 1) {
      const double bytes = (double)panel_floats * 4 * steps;
      printf("%-28s panel %6d B  blocks %4d waves %d: %8.1f us  %7.1f cycles/step(clk=%.0f)  %.1f B/clk/CU(by time @2.4GHz)  %.2f TB/s chip\n", name,
             panel_floats * 4, blocks, NW, ms * 1e3, ms * 1e-3 * 2.4e9 / steps, (double)c / steps, bytes / (ms * 1e-3 * 2.4e9),
             bytes * blocks / (ms * 1e-3) / 1e12);
    }
  }
}

int main() {
  const int npanels = 32;
  const int pf_max = 16896;
  float* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, (size_t)npanels * pf_max * 4); hipMemset(src, 0, (size_t)npanels * pf_max * 4);
  hipMalloc(&out, 64); hipMalloc(&cyc, 8);
  for (int blocks : {1, 32, 256, 512}) {
    run<0, 8>("lds-dma x4 (66 KiB panels)", src, 16896, npanels, blocks, out, cyc);
    run<0, 8>("lds-dma x4 (33 KiB panels)", src, 8448, npanels, blocks, out, cyc);
    run<1, 8>("load+ds_write (33 KiB)", src, 8448, npanels, blocks, out, cyc);
    run<2, 8>("lds-dma x1 (33 KiB)", src, 8448, npanels, blocks, out, cyc);
  }
  run<0, 4>("lds-dma x4 (33 KiB) 2 blk/CU", src, 8448, npanels, 512, out, cyc);
  run<0, 16>("lds-dma x4 (66 KiB) 16 waves", src, 16896, npanels, 256, out, cyc);
  return 0;
}
