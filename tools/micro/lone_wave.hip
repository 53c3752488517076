// What does ONE wave on an otherwise idle CU pay per instruction?  (The S chain of the tri-factorisation is such a wave.)
// Shader-clock stamps (s_memtime) around .rept blocks of hand-written instructions; 1 block of 64 threads.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lone_wave.hip -o /tmp/lone_wave && /tmp/lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>

#define STAMP(t) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#define REP(N, body) asm volatile(".rept " #N "\n\t" body "\n\t.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1) : "v"(w) : "vcc", "memory")

__global__ __launch_bounds__(64) void k(unsigned long long* out, float seed) {
  __shared__ float lds[1024];
  float a = seed + threadIdx.x, b = seed * 2.f, c = seed * 3.f, d = seed * 4.f, w = 0.999f;
  unsigned s0 = 1, s1 = 2;
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (float)((i * 4 + 4) & 4095);   // a pointer chain: next byte address
  __syncthreads();
  unsigned long long t0, t1; int n = 0;
  // 0: empty
  STAMP(t0); STAMP(t1); out[n++] = t1 - t0;
  // 1: 256 dependent v_fma
  STAMP(t0); REP(256, "v_fma_f32 %0, %0, %6, %1"); STAMP(t1); out[n++] = t1 - t0;
  // 2: 256 v_fma, four independent chains
  STAMP(t0); REP(64, "v_fma_f32 %0, %0, %6, %6\n\tv_fma_f32 %1, %1, %6, %6\n\tv_fma_f32 %2, %2, %6, %6\n\tv_fma_f32 %3, %3, %6, %6"); STAMP(t1); out[n++] = t1 - t0;
  // 3: 256 dependent v_rcp
  STAMP(t0); REP(256, "v_rcp_f32 %0, %0"); STAMP(t1); out[n++] = t1 - t0;
  // 4: 256 v_rcp, four independent
  STAMP(t0); REP(64, "v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3"); STAMP(t1); out[n++] = t1 - t0;
  // 5: 128 x (v_readlane -> v_fma with the SGPR), dependent
  STAMP(t0); REP(128, "v_readlane_b32 %4, %0, 3\n\tv_fma_f32 %0, %0, %6, %4"); STAMP(t1); out[n++] = t1 - t0;
  // 6: 64 x (v_cmp -> s_and -> s_ff1 -> v_readlane (lane from SGPR) -> v_add), dependent
  STAMP(t0); REP(64, "v_cmp_lt_f32 vcc, %1, %0\n\ts_and_b32 %4, vcc_lo, 0xffff\n\ts_ff1_i32_b32 %5, %4\n\tv_readlane_b32 %4, %0, %5\n\tv_add_f32 %0, %0, %4"); STAMP(t1); out[n++] = t1 - t0;
  // 7: 256 dependent s_add
  STAMP(t0); REP(256, "s_add_u32 %4, %4, %5"); STAMP(t1); out[n++] = t1 - t0;
  // 8: 128 x (v_fma ; s_add) independent of each other
  STAMP(t0); REP(128, "v_fma_f32 %0, %0, %6, %1\n\ts_add_u32 %4, %4, %5"); STAMP(t1); out[n++] = t1 - t0;
  // 9: 64 dependent LDS reads (address = the value read)
  {
    unsigned addr = threadIdx.x * 4;
    STAMP(t0);
    asm volatile(".rept 64\n\tds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_cvt_u32_f32 %0, %0\n\t.endr" : "+v"(addr) :: "memory");
    STAMP(t1); out[n++] = t1 - t0; a += (float)addr;
  }
  // 10: 256 dependent v_mul with DPP row_shr (cross-lane inside the vector unit)
  STAMP(t0); REP(256, "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf"); STAMP(t1); out[n++] = t1 - t0;
  // 11: 128 x (v_cmp_class -> s_cbranch_vccz not taken) + v_fma
  STAMP(t0); REP(128, "v_cmp_gt_f32 vcc, %0, %1\n\tv_fma_f32 %0, %0, %6, %1"); STAMP(t1); out[n++] = t1 - t0;
  // 12: 256 dependent v_sqrt
  STAMP(t0); REP(256, "v_sqrt_f32 %0, %0"); STAMP(t1); out[n++] = t1 - t0;
  // 13: 256 v_mov (no dependency at all), 4 registers
  STAMP(t0); REP(64, "v_mov_b32 %0, %6\n\tv_mov_b32 %1, %6\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %6"); STAMP(t1); out[n++] = t1 - t0;
  // 14: 128 x (v_writelane ; v_fma)
  STAMP(t0); REP(128, "v_writelane_b32 %0, %4, 5\n\tv_fma_f32 %0, %0, %6, %1"); STAMP(t1); out[n++] = t1 - t0;
  // 15: 64 x (s_set_gpr_idx_on ; v_mov (relative source) ; s_set_gpr_idx_off ; v_fma)
  STAMP(t0); REP(64, "s_set_gpr_idx_on %5, gpr_idx(SRC0)\n\tv_mov_b32 %1, %2\n\ts_set_gpr_idx_off\n\tv_fma_f32 %0, %0, %6, %1"); STAMP(t1); out[n++] = t1 - t0;
  // 16: 64 x (v_cmp -> s_and -> s_cmp -> s_cbranch (not taken) ; v_fma)
  STAMP(t0); REP(64, "v_cmp_lt_f32 vcc, %1, %0\n\ts_and_b32 %4, vcc_lo, 0x10000\n\ts_cmp_lg_u32 %4, 0x12345\n\ts_cbranch_scc0 1f\n\tv_fma_f32 %0, %0, %6, %1\n1:"); STAMP(t1); out[n++] = t1 - t0;
  // 17: 64 x the row chain's correction step: sub mul fma sqrt add rcp mul cmp s_and s_ff1 readlane x2 fma (13 instructions)
  STAMP(t0); REP(64, "v_sub_f32 %1, %0, %6\n\tv_mul_f32 %2, %1, %6\n\tv_fma_f32 %3, %2, %2, 4.0\n\tv_sqrt_f32 %3, %3\n\tv_add_f32 %3, %3, %2\n\tv_rcp_f32 %3, %3\n\tv_mul_f32 %2, %3, %6\n\tv_cmp_le_f32 vcc, |%2|, %6\n\ts_and_b32 %4, vcc_lo, 0xffff\n\ts_ff1_i32_b32 %5, %4\n\tv_readlane_b32 %4, %2, %5\n\tv_readlane_b32 %5, %3, %5\n\tv_fma_f32 %0, %0, %4, %0"); STAMP(t1); out[n++] = t1 - t0;
  // 18: 128 x (v_readlane with a constant lane ; s_nop 0)  19: 128 x v_cndmask with vcc written just before
  STAMP(t0); REP(128, "v_cmp_eq_u32 vcc, %4, %1\n\tv_cndmask_b32 %0, %0, %2, vcc"); STAMP(t1); out[n++] = t1 - t0;
  out[n++] = (unsigned long long)(a + b + c + d) + s0 + s1;
}
int main() {
  unsigned long long* out; (void)hipMalloc(&out, 64 * 8);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, 1.5f); (void)hipDeviceSynchronize(); }
  unsigned long long h[64]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"empty", "256 dependent v_fma", "256 v_fma in 4 independent chains", "256 dependent v_rcp", "256 v_rcp 4 independent",
                         "128 x (readlane -> fma)", "64 x (cmp -> s_and -> s_ff1 -> readlane(sgpr lane) -> add)", "256 dependent s_add",
                         "128 x (v_fma ; s_add)", "64 dependent ds_read_b32 (+waitcnt, cvt)", "256 dependent v_add dpp row_shr", "128 x (v_cmp ; v_fma)",
                         "256 dependent v_sqrt", "256 independent v_mov", "128 x (v_writelane ; v_fma)",
                         "64 x (s_set_gpr_idx_on ; v_mov rel ; s_set_gpr_idx_off ; v_fma)", "64 x (v_cmp ; s_and ; s_cmp ; s_cbranch not taken ; v_fma)",
                         "64 x (sub mul fma sqrt add rcp mul cmp s_and s_ff1 readlane readlane fma)", "128 x (v_cmp_eq ; v_cndmask vcc)"};
  const int cnt[] = {1, 256, 256, 256, 256, 256, 320, 256, 256, 64, 256, 256, 256, 256, 256, 256, 320, 832, 256};
  for (int i = 0; i < 19; ++i) printf("%-64s %7llu cycles  = %.1f per instruction (of %d)\n", names[i], h[i], (double)(h[i] - h[0]) / cnt[i], cnt[i]);
  return 0;
}
