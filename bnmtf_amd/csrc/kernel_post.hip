// After a half sweep (or a state upload) the freshly written factor X [rows][KP] is
// re-laid out for its two consumers and its Gram matrix is formed:
//   XT  [KP][ldT]        column k contiguous      -> LDS panel of the other direction's sweep
//   XT2 [KP/2][ld2][2]   column pairs interleaved -> pre-pass panels (ds_read_b64)
//   C = X^T X (fp64), column sums                 -> sweep (fp32 copy) and the SSE identity
// One pass over X: kPostRows (32) rows per block staged in LDS; per-block Gram partials go to a
// slab and are summed by gram_reduce_kernel (deterministic, no atomics).
#include "kernels.h"
#include "many.h"

namespace bnmtf {

// tile p of the packed upper triangle (rows of tiles 0 .. NT-1, row y holding tiles (y, y) .. (y, NT-1)) -> (ty, tx)
__device__ __forceinline__ void tri_tile(int p, int NT, int* ty, int* tx) {
  int y = 0, rem = p;
  while (rem >= NT - y) { rem -= NT - y; ++y; }
  *ty = y; *tx = y + rem;
}

template <bool VB>
__device__ __forceinline__ void post_body(const PostArgs& a) {
  constexpr int RB = kPostRows;
  constexpr int LD = 68, LDD = 66;           // floats per row of a layout tile ; doubles per row of the Gram tile
  // a layout block holds X rows (VB: and S2 rows) as floats; a Gram block holds its X rows as DOUBLES (converted once, when the
  // tile is filled: the 4 x 4 products below would otherwise convert every operand of every row again, 17 x as many conversions)
  __shared__ double smem[(RB * LDD * sizeof(double) >= (VB ? 2 : 1) * RB * LD * sizeof(float) ? RB * LDD : ((VB ? 2 : 1) * RB * LD + 1) / 2) + 4 * 64];
  float* tile0 = reinterpret_cast<float*>(smem);
  float* tile1 = tile0 + RB * LD;            // (VB layout blocks only)
  double* dtile = smem;
  double* dred = smem + RB * LDD;            // [4][64] column-sum partials
  const int KP = a.KP, tid = threadIdx.x;
  const int r0 = (a.blk0 + blockIdx.x) * RB;
  const int nr = min(RB, a.rows - r0);
  // two blocks per row group (blockIdx.y): one writes the layouts, the other forms the Gram partial and the column sums.
  // Each is half as long, and twice as many blocks hide each other's load -> store / load -> FMA latencies.
  // (a launch that does only one of the two has one block per group)
  const bool do_layout = a.do_layout && (!a.do_gram || blockIdx.y == 0), do_gram = a.do_gram && (!a.do_layout || blockIdx.y == 1);
  if (do_gram) {
    for (int pass = 0; pass < (VB ? 2 : 1); ++pass) {
      const float* src = pass == 0 ? a.X : a.S2;
      if (pass == 1) __syncthreads();
      for (int t = tid; t < RB * KP; t += 256) {
        const int r = t / KP, k = t % KP;
        // (Gram-only launches of a multi-GPU run: rows of other ranks count as zero)
        dtile[r * LDD + k] = (r < nr && (a.do_layout || (r0 + r >= a.own0 && r0 + r < a.own1))) ? (double)src[(size_t)(r0 + r) * KP + k] : 0.0;
      }
      __syncthreads();
      if (pass == 0) {
        // Gram partial of this block's rows: C is symmetric, so only the 4x4 tiles on and above the diagonal are formed
        // (NT (NT + 1) / 2 of NT^2, NT = KP / 4), one per thread, and stored packed -- 16 contiguous doubles per tile
        const int NT = KP / 4, NU = NT * (NT + 1) / 2;
        if (tid < NU) {
          int ty, tx;
          tri_tile(tid, NT, &ty, &tx);
          double acc[4][4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
          const double* pa = dtile + 4 * ty;
          const double* pb = dtile + 4 * tx;
          // (rows beyond nr are zero: the loop runs over the whole tile, unrolled, operands of four rows in flight)
#pragma unroll 4
          for (int r = 0; r < RB; ++r) {
            const double2 a01 = *reinterpret_cast<const double2*>(pa + r * LDD), a23 = *reinterpret_cast<const double2*>(pa + r * LDD + 2);
            const double2 b01 = *reinterpret_cast<const double2*>(pb + r * LDD), b23 = *reinterpret_cast<const double2*>(pb + r * LDD + 2);
            const double ad[4] = {a01.x, a01.y, a23.x, a23.y}, bd[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[i][j] = fma(ad[i], bd[j], acc[i][j]);
          }
          double2* out = reinterpret_cast<double2*>(a.Cpart + ((size_t)blockIdx.x * NU + tid) * 16);
#pragma unroll
          for (int i = 0; i < 4; ++i) { out[2 * i] = double2{acc[i][0], acc[i][1]}; out[2 * i + 1] = double2{acc[i][2], acc[i][3]}; }
        }
      }
      // column sums: thread = column + 64 * group, a group adds every fourth row, the four partials meet in LDS
      {
        const int col = tid & 63, grp = tid >> 6;
        double sp = 0.0, mx = 0.0;
        if (col < KP)
#pragma unroll
          for (int r = grp; r < RB; r += 4) { const double v = dtile[r * LDD + col]; sp += v; mx = fmax(mx, v); }
        dred[grp * 64 + col] = sp;
        __syncthreads();
        if (tid < KP) (pass == 0 ? a.spart : a.s2part)[(size_t)blockIdx.x * KP + tid] = (dred[tid] + dred[64 + tid]) + (dred[128 + tid] + dred[192 + tid]);
        if (VB && a.mpart) {
          // VB, whole-factor launches: the block's largest S2 (columns [0, KP)) and largest E^2 ([KP, 2 KP)) per column -- the
          // fixed-point grid of kernel_maskgemm.hip's digit planes (E >= 0: the largest square is the square of the largest)
          __syncthreads();
          dred[grp * 64 + col] = mx;
          __syncthreads();
          if (tid < KP) {
            const float m = (float)fmax(fmax(dred[tid], dred[64 + tid]), fmax(dred[128 + tid], dred[192 + tid]));
            a.mpart[(size_t)blockIdx.x * 2 * KP + (pass == 0 ? KP : 0) + tid] = pass == 0 ? m * m : m;
          }
        }
      }
    }
    return;
  }
  if (!do_layout) return;
  const float* src = a.X;
  for (int pass = 0; pass < (VB ? 2 : 1); ++pass) {
    float* tile = pass == 0 ? tile0 : tile1;
    if (pass == 1) src = a.S2;
    for (int t = tid; t < RB * KP; t += 256) {
      const int r = t / KP, k = t % KP;
      tile[r * LD + k] = r < nr ? src[(size_t)(r0 + r) * KP + k] : 0.f;
    }
    __syncthreads();
    float* T1 = pass == 0 ? a.XT : a.S2T;
    if (T1)
      for (int t = tid; t < RB * KP; t += 256) {
        const int k = t / RB, r = t % RB;
        if (r < nr) T1[(size_t)k * a.ldT + r0 + r] = tile[r * LD + k];
      }
    if (pass == 0 && a.snap)
      for (int t = tid; t < RB * a.snapW; t += 256) {
        const int r = t / a.snapW, k = t % a.snapW;
        if (r < nr) a.snap[(size_t)(r0 + r) * a.snapW + k] = tile[r * LD + k];
      }
    if (pass == 0 && a.XT2)
      for (int t = tid; t < RB * KP; t += 256) {
        const int kp = t / (2 * RB), rem = t % (2 * RB), r = rem >> 1, c = rem & 1;
        if (r < nr) a.XT2[((size_t)kp * a.ld2 + r0 + r) * 2 + c] = tile[r * LD + 2 * kp + c];
      }
    if (pass == 1 && a.XS)        // VB: the (E, S2) pair panels of the fast VB sweep, [KP][ldT][2]
      for (int t = tid; t < RB * KP; t += 256) {
        const int k = t / RB, r = t % RB;
        if (r < nr) *reinterpret_cast<float2*>(a.XS + ((size_t)k * a.ldT + r0 + r) * 2) = float2{tile0[r * LD + k], tile1[r * LD + k]};
      }
  }
}
template <bool VB>
__global__ __launch_bounds__(256) void post_kernel(PostArgs a) { post_body<VB>(a); }
template <bool VB>       // list form (many.h): blockIdx.z = model
__global__ __launch_bounds__(256) void post_many(const PostArgs* list, int) { post_body<VB>(load_pack(list, blockIdx.z)); }

// 32 packed entries per block, 32 partial-slab strides per entry (thread = entry e + 32*g), summed through LDS in a fixed
// order: ceil(PS/32) blocks of 1024 threads (+ one for the column sums), every slab read is a coalesced 256 B segment; a
// tile above the diagonal is written twice (itself and its mirror).
__device__ __forceinline__ void gram_reduce_body(const PostArgs& a, int nblk) {
  __shared__ double red[1024];
  const int KP = a.KP;
  if (a.S2 && a.mpart && a.umax && (int)blockIdx.x == (int)gridDim.x - 2) {
    // one more extra block (launch_post adds it): the column maxima of [S2 | E^2] over the blocks' partials -- the bits of the
    // largest element (non-negative floats order like unsigned integers), read by vb_planes_kernel.  Thread = column + 128 * group,
    // eight loads in flight.  (Inside the column-sum block this doubled the kernel's 6 us.)
    float* fred = reinterpret_cast<float*>(red);
    const int c = threadIdx.x & 127, g8 = threadIdx.x >> 7;           // 8 groups of 128 columns
    float m = 0.f;
    if (c < 2 * KP)
      for (int b0 = g8; b0 < nblk; b0 += 8 * 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = b0 + 8 * u < nblk ? a.mpart[(size_t)(b0 + 8 * u) * 2 * KP + c] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(m, w[u]);
      }
    fred[threadIdx.x] = m;
    __syncthreads();
    if (g8 == 0 && c < 2 * KP) {
#pragma unroll
      for (int j = 1; j < 8; ++j) m = fmaxf(m, fred[c + 128 * j]);
      a.umax[c] = __builtin_bit_cast(unsigned, m);
    }
    return;
  }
  if ((int)blockIdx.x == (int)gridDim.x - 1) {
    // the extra block: column sums, in parallel with the Gram blocks.  Thread = column + 64 * group: a wave reads one
    // coalesced row of partials per load, eight loads in flight, 16 groups summed through LDS in a fixed order
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    double v = 0.0, v2 = 0.0;
    if (col < KP)
      for (int b0 = grp; b0 < nblk; b0 += 8 * 16) {
        double w[8], w2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int b = b0 + 16 * u;
          w[u] = b < nblk ? a.spart[(size_t)b * KP + col] : 0.0;
          w2[u] = (a.S2 && b < nblk) ? a.s2part[(size_t)b * KP + col] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { v += w[u]; v2 += w2[u]; }
      }
    red[threadIdx.x] = v;
    __syncthreads();
    if (grp == 0 && col < KP) {
      double tot = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) tot += red[col + 64 * j];
      a.colsum[col] = tot;
    }
    if (a.S2) {
      __syncthreads();
      red[threadIdx.x] = v2;
      __syncthreads();
      if (grp == 0 && col < KP) {
        double tot = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) tot += red[col + 64 * j];
        a.colsum2[col] = tot;
      }
    }
    return;
  }
  const int NT = KP / 4, NU = NT * (NT + 1) / 2, PS = NU * 16;       // packed slab: NU tiles of 16 doubles
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + e;                                   // packed entry
  double s = 0.0;
  if (t < PS)
    for (int b = g; b < nblk; b += 32) s += a.Cpart[(size_t)b * PS + t];
  red[threadIdx.x] = s;
  __syncthreads();
  if (g == 0 && t < PS) {                // one barrier, then 32 threads add the 32 strides in a fixed order
    double tot = 0.0;
#pragma unroll
    for (int j = 0; j < 32; ++j) tot += red[e + 32 * j];
    int ty, tx;
    tri_tile(t >> 4, NT, &ty, &tx);
    const int row = 4 * ty + ((t >> 2) & 3), col = 4 * tx + (t & 3);
    a.C64[row * KP + col] = tot; a.C32[row * KP + col] = (float)tot;
    if (ty != tx) { a.C64[col * KP + row] = tot; a.C32[col * KP + row] = (float)tot; }    // the mirror tile
  }
}
__global__ __launch_bounds__(1024) void gram_reduce_kernel(PostArgs a, int nblk) { gram_reduce_body(a, nblk); }
struct GramReducePack { PostArgs a; int nblk; int pad_; };
__global__ __launch_bounds__(1024) void gram_reduce_many(const GramReducePack* list, int) {
  const GramReducePack p = load_pack(list, blockIdx.z);
  gram_reduce_body(p.a, p.nblk);
}

void launch_post(const PostArgs& a0, hipStream_t st) {
  PostArgs a = a0;
  a.do_layout = 1; a.do_gram = 1; a.blk0 = 0; a.own0 = 0; a.own1 = a.rows;
  const int nblk = post_blocks(a.rows);
  const int nt = a.KP / 4, ps = nt * (nt + 1) / 2 * 16;
  const int extra = (a.S2 && a.mpart && a.umax) ? 2 : 1;      // the column-sum block, and (VB with the masked sums) the column-maximum block
  if (g_recorder) {
    record_launch(a.S2 ? (const void*)post_many<true> : (const void*)post_many<false>, dim3(nblk, 2), dim3(256), 0, a);
    GramReducePack p; memset(&p, 0, sizeof(p)); p.a = a; p.nblk = nblk;
    record_launch((const void*)gram_reduce_many, dim3((ps + 31) / 32 + extra), dim3(1024), 0, p);
    return;
  }
  if (a.S2) hipLaunchKernelGGL(post_kernel<true>, dim3(nblk, 2), dim3(256), 0, st, a); else hipLaunchKernelGGL(post_kernel<false>, dim3(nblk, 2), dim3(256), 0, st, a);
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((ps + 31) / 32 + extra), dim3(1024), 0, st, a, nblk);
}

void launch_post_layout(const PostArgs& a0, hipStream_t st) {
  PostArgs a = a0;
  a.do_layout = 1; a.do_gram = 0; a.blk0 = 0; a.own0 = 0; a.own1 = a.rows;
  if (a.S2) hipLaunchKernelGGL(post_kernel<true>, dim3(post_blocks(a.rows), 1), dim3(256), 0, st, a); else hipLaunchKernelGGL(post_kernel<false>, dim3(post_blocks(a.rows), 1), dim3(256), 0, st, a);
}

void launch_post_gram_rows(const PostArgs& a0, int own0, int own1, hipStream_t st) {
  PostArgs a = a0;
  a.do_layout = 0; a.do_gram = 1; a.own0 = own0; a.own1 = own1;
  a.blk0 = own0 / kPostRows;
  const int nblk = own1 > own0 ? (own1 + kPostRows - 1) / kPostRows - a.blk0 : 0;
  if (nblk > 0) { if (a.S2) hipLaunchKernelGGL(post_kernel<true>, dim3(nblk, 1), dim3(256), 0, st, a); else hipLaunchKernelGGL(post_kernel<false>, dim3(nblk, 1), dim3(256), 0, st, a); }
  const int nt = a.KP / 4, ps = nt * (nt + 1) / 2 * 16;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((ps + 31) / 32 + 1), dim3(1024), 0, st, a, nblk);     // no rows: zeros
}

__global__ void gram_cast_kernel(const double* C64, float* C32, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) C32[t] = (float)C64[t];
}
void launch_gram_cast(const double* C64, float* C32, int n, hipStream_t st) {
  hipLaunchKernelGGL(gram_cast_kernel, dim3((n + 255) / 256), dim3(256), 0, st, C64, C32, n);
}

}  // namespace bnmtf
