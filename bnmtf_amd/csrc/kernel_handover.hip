// Tables of the q hand-over between the half sweeps (model.h Dir::ho_*, sweep_chip.inc), built on the device from the slot
// layouts that are already there -- a few hundred microseconds instead of a quarter of a second of host passes per model.
//
// One writer / reader pair of directions.  An entry (i, j) of the mask sits in a slot of W's layout (unit i, inner index j)
// and in a slot of Rd's (unit j, inner index i).  W's block bw stages the final q of its entries sorted by Rd's block,
// [bw][br] one run padded to four entries; run (bw, br) is copied as it is into block br's region at Rd, which is the runs
// (0, br), (1, br), ... behind one another, then >= 32 zeros (what empty slots read), padded to 256 entries (one LDS-DMA
// piece).  t_out (W) and t_in (Rd) give every slot its 16-bit place in the staging area / the region; two per word, packed
// like the slot table (rows 2r and 2r + 1 of a lane share a word).
// The place of an entry INSIDE its run is handed out by an LDS atomic, i.e. in no fixed order: q travels verbatim, so the
// order is invisible in the results.
#include "kernels.h"

namespace bnmtf {

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* a, uint32_t n, uint32_t v) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Rd: slot of every (unit, missing inner index), in the order of the unit's sorted index list (Dir::idx)
__global__ __launch_bounds__(256) void ho_inverse_kernel(HandoverArgs a) {
  const int pi = blockIdx.x;
  const uint32_t E = a.r_pE[pi], base = a.r_pB[pi];
  for (uint32_t t = threadIdx.x; t < E * 64u; t += 256u) {
    const uint32_t l = t & 63u;
    const int ul = a.r_umap[2 * pi + (l >> 5)];
    const size_t sid = (size_t)base * 64 + t;
    const uint32_t v = a.r_off[sid];
    if (ul < 0 || v >= a.r_inner) continue;
    const uint32_t lo = a.r_ptr[ul], n = a.r_ptr[ul + 1] - lo;
    a.inv[lo + lower_bound_u32(a.r_idx + lo, n, v)] = (uint32_t)sid;
  }
}

// W's block bw: where does each of its entries live at Rd, and which place does it take in the run to that block?
__global__ __launch_bounds__(1024) void ho_destination_kernel(HandoverArgs a) {
  extern __shared__ uint32_t cnt[];
  const int bw = blockIdx.x;
  for (int t = threadIdx.x; t < a.r_nb; t += 1024) cnt[t] = 0u;
  __syncthreads();
  for (int pi = a.w_ppb * bw; pi < min(a.w_ppb * (bw + 1), a.w_npairs); ++pi) {
    const uint32_t E = a.w_pE[pi], base = a.w_pB[pi];
    for (uint32_t t = threadIdx.x; t < E * 64u; t += 1024u) {
      const uint32_t l = t & 63u;
      const int ul = a.w_umap[2 * pi + (l >> 5)];
      const size_t sid = (size_t)base * 64 + t;
      const uint32_t j = a.w_off[sid];
      uint32_t rs = 0xFFFFFFFFu, rank = 0u;
      if (ul >= 0 && j < a.w_inner) {
        const uint32_t lo = a.r_ptr[j], n = a.r_ptr[j + 1] - lo;
        const uint32_t p = lower_bound_u32(a.r_idx + lo, n, (uint32_t)ul);
        if (p < n && a.r_idx[lo + p] == (uint32_t)ul) {
          rs = a.inv[lo + p];
          rank = atomicAdd(&cnt[a.r_row_blk[rs >> 6]], 1u);
        } else atomicAdd(&a.limits[2], 1u);            // the two layouts disagree about the mask: no hand-over
      }
      a.dst_slot[sid] = rs; a.dst_rank[sid] = rank;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < a.r_nb; t += 1024) a.count[(size_t)bw * a.r_nb + t] = cnt[t];
}

__device__ __forceinline__ uint32_t pad4(uint32_t v) { return (v + 3u) & ~3u; }

// the starts of the runs in W's staging areas (block x < w_nb) and in Rd's regions (block x >= w_nb)
__global__ void ho_scan_kernel(HandoverArgs a) {
  if (threadIdx.x != 0) return;
  const int b = blockIdx.x;
  uint32_t run = 0;
  if (b < a.w_nb) {
    for (int br = 0; br < a.r_nb; ++br) { a.sbase[(size_t)b * a.r_nb + br] = run; run += pad4(a.count[(size_t)b * a.r_nb + br]); }
    a.stotal[b] = run;
    atomicMax(&a.limits[0], run);
  } else {
    const int br = b - a.w_nb;
    for (int bw = 0; bw < a.w_nb; ++bw) { a.rbase[(size_t)br * a.w_nb + bw] = run; run += pad4(a.count[(size_t)bw * a.r_nb + br]); }
    a.rdata[br] = run;
    const uint32_t size = (run + 32u + 255u) & ~255u;
    a.rsize[br] = size;
    atomicMax(&a.limits[1], size);
  }
}
__global__ void ho_region_starts_kernel(HandoverArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t run = 0;
  for (int br = 0; br < a.r_nb; ++br) { a.rofs[br] = run; run += a.rsize[br]; }
  a.rofs[a.r_nb] = run;
  a.limits[3] = run;
}
__global__ __launch_bounds__(256) void ho_packets_kernel(HandoverArgs a) {
  const int bw = blockIdx.x;
  for (int br = threadIdx.x; br < a.r_nb; br += 256) {
    uint32_t* p = a.pk + ((size_t)bw * a.r_nb + br) * 3;
    p[0] = a.sbase[(size_t)bw * a.r_nb + br]; p[1] = pad4(a.count[(size_t)bw * a.r_nb + br]); p[2] = a.rofs[br] + a.rbase[(size_t)br * a.w_nb + bw];
  }
}
// the 16-bit place of slot (row, lane): rows 2r and 2r + 1 share the word of lane l
__device__ __forceinline__ size_t half_index(size_t row, uint32_t l) { return (((row >> 1) * 64 + l) << 1) + (row & 1); }

__global__ __launch_bounds__(256) void ho_reader_default_kernel(HandoverArgs a) {       // every slot of Rd: the zeros behind its block's runs
  const int pi = blockIdx.x;
  const uint32_t E = a.r_pE[pi], base = a.r_pB[pi];
  const uint32_t zero0 = a.rdata[pi / a.r_ppb];
  for (uint32_t t = threadIdx.x; t < E * 64u; t += 256u) a.t_in[half_index((size_t)base + (t >> 6), t & 63u)] = (uint16_t)(zero0 + (t & 31u));
}
__global__ __launch_bounds__(256) void ho_tables_kernel(HandoverArgs a) {
  const int pi = blockIdx.x, bw = pi / a.w_ppb;
  const uint32_t E = a.w_pE[pi], base = a.w_pB[pi];
  for (uint32_t t = threadIdx.x; t < E * 64u; t += 256u) {
    const uint32_t l = t & 63u;
    const size_t sid = (size_t)base * 64 + t;
    const uint32_t rs = a.dst_slot[sid];
    uint32_t out = a.stotal[bw] + (l & 31u);          // an empty slot's q (zero) goes to the 32 dump words behind the runs
    if (rs != 0xFFFFFFFFu) {
      const uint32_t br = a.r_row_blk[rs >> 6], rank = a.dst_rank[sid];
      out = a.sbase[(size_t)bw * a.r_nb + br] + rank;
      a.t_in[half_index(rs >> 6, rs & 63u)] = (uint16_t)(a.rbase[(size_t)br * a.w_nb + bw] + rank);
    }
    a.t_out[half_index((size_t)base + (t >> 6), l)] = (uint16_t)out;
  }
}

void launch_handover_build(const HandoverArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(ho_inverse_kernel, dim3(a.r_npairs), dim3(256), 0, st, a);
  hipLaunchKernelGGL(ho_destination_kernel, dim3(a.w_nb), dim3(1024), sizeof(uint32_t) * (size_t)a.r_nb, st, a);
  hipLaunchKernelGGL(ho_scan_kernel, dim3(a.w_nb + a.r_nb), dim3(64), 0, st, a);
  hipLaunchKernelGGL(ho_region_starts_kernel, dim3(1), dim3(64), 0, st, a);
  hipLaunchKernelGGL(ho_packets_kernel, dim3(a.w_nb), dim3(256), 0, st, a);
  hipLaunchKernelGGL(ho_reader_default_kernel, dim3(a.r_npairs), dim3(256), 0, st, a);
  hipLaunchKernelGGL(ho_tables_kernel, dim3(a.w_npairs), dim3(256), 0, st, a);
}

}  // namespace bnmtf
