// Does a second pass over a 256 MiB operand find part of it in the 256 MB Infinity Cache when it runs in REVERSE order?
// (the two contractions of an iteration stream R~ and R~^T: if one array served both, the second could start where the first ended)
// hipcc --offload-arch=gfx950 -O3 tools/micro/mall_probe.hip -o /tmp/mall_probe && /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(256) void stream(const f32x4* p, size_t chunk_vec, int nchunks, int reverse, float* out) {
  const int b = reverse ? nchunks - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const f32x4* q = p + (size_t)b * chunk_vec;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = threadIdx.x; i < chunk_vec; i += 256 * 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = i + u * 256 < chunk_vec ? (NT ? __builtin_nontemporal_load(q + i + u * 256) : q[i + u * 256]) : acc;
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
template <int NT>
void run(const f32x4* p, size_t bytes, float* out) {
  const int nchunks = 4096;
  const size_t chunk_vec = bytes / 16 / nchunks;
  hipEvent_t e[8]; for (auto& x : e) (void)hipEventCreate(&x);
  const int order[7] = {0, 0, 1, 0, 1, 1, 0};
  for (int i = 0; i < 7; ++i) {
    (void)hipEventRecord(e[i]);
    hipLaunchKernelGGL(stream<NT>, dim3(nchunks), dim3(256), 0, 0, p, chunk_vec, nchunks, order[i], out);
  }
  (void)hipEventRecord(e[7]); (void)hipEventSynchronize(e[7]);
  printf("%s loads, %zu MiB:", NT ? "nontemporal" : "plain", bytes >> 20);
  for (int i = 0; i < 7; ++i) { float ms; (void)hipEventElapsedTime(&ms, e[i], e[i + 1]); printf("  %s %.1f us (%.2f TB/s)", order[i] ? "rev" : "fwd", ms * 1e3, bytes / (ms * 1e-3) / 1e12); }
  printf("\n");
}
int main() {
  for (size_t mib : {128, 256, 512}) {
    const size_t bytes = mib << 20;
    f32x4* p; float* out; (void)hipMalloc(&p, bytes); (void)hipMalloc(&out, 4); (void)hipMemset(p, 0, bytes);
    run<1>(p, bytes, out); run<0>(p, bytes, out);
    (void)hipFree(p); (void)hipFree(out);
  }
  return 0;
}
