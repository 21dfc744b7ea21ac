// t2i_aux.hip — the bandwidth-bound kernels around the implicit GEMMs (gfx950 only): column reductions (bias
// gradients, BatchNorm moments), BatchNorm finalize/apply/backward, activations, residual joins, x_hat interpolation,
// text-embedding tile+concat, NCHW<->NHWC transposes, the gradient-penalty slope norm and TF-flavoured Adam.
// All are HBM-bound: 16-byte accesses where the shape allows, grid-stride loops capped at 2048 blocks (256 CUs x 8),
// wave64 shuffle reductions, fixed summation order (deterministic).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include <type_traits>

#include "t2i_internal.h"

namespace t2i {

static inline int ew_blocks(size_t n_items) {
  size_t b = (n_items + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vscale(float a, float s) { return a * s; }
__device__ __forceinline__ float4 vscale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

__device__ __forceinline__ float vsplat(float s, float) { return s; }
__device__ __forceinline__ float4 vsplat(float s, float4) { return make_float4(s, s, s, s); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// column reduction, stage 1: grid (col tiles of 64, row chunks); block (64 cols, 4 row lanes)
// ---------------------------------------------------------------------------------------------------------------
// Two refinements shared by the stage-1 kernels (both exist for batch norm, see bn_stats_* below):
//   shift   (second moment of a itself): every value is taken relative to the chunk's first row, s[c] = a[rbeg, c], i.e. the
//           partials are sum(a - s) and sum((a - s)^2) — the raw sums sum(a), sum(a^2) lose var = E[a^2] - E[a]^2 to
//           cancellation as soon as |mean| >> std (worst with few rows: a rank-2 batch norm over a batch of 2);
//   center  (second factor b): the partial is sum(a * (b - center[c])): the batch-norm backward needs sum(dy * (x - mean)),
//           and forming it as sum(dy*x) - mean*sum(dy) cancels the same way.
__global__ __launch_bounds__(256) void col_reduce_stage1(const float* __restrict__ a, const float* __restrict__ b,
                                                         int64_t rows, int C, int64_t rows_per_chunk,
                                                         float* __restrict__ part0, float* __restrict__ part1,
                                                         bool want1, int shift, const float* __restrict__ center) {
  __shared__ float s0[4][64], s1[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int64_t rbeg = (int64_t)blockIdx.y * rows_per_chunk;
  int64_t rend = rbeg + rows_per_chunk;
  if (rend > rows) rend = rows;
  float acc0 = 0.f, acc1 = 0.f;
  if (c < C) {
    const float sh = shift ? a[rbeg * C + c] : 0.f;
    const float ce = center ? center[c] : 0.f;
    for (int64_t r = rbeg + ty; r < rend; r += 4) {
      float va = a[r * C + c] - sh;
      acc0 += va;
      if (want1) acc1 += va * (b ? b[r * C + c] - ce : va);
    }
  }
  s0[ty][tx] = acc0;
  s1[ty][tx] = acc1;
  __syncthreads();
  if (ty == 0 && c < C) {
    part0[(size_t)blockIdx.y * C + c] = (s0[0][tx] + s0[1][tx]) + (s0[2][tx] + s0[3][tx]);
    if (want1) part1[(size_t)blockIdx.y * C + c] = (s1[0][tx] + s1[1][tx]) + (s1[2][tx] + s1[3][tx]);
  }
}

// stage 1 for C <= 4 (bias gradients of the 3-channel image layers): with lane = column only C of 64 lanes would work,
// so here a thread owns whole rows (consecutive threads, consecutive rows: contiguous 4*C-byte pieces) and the workgroup
// joins its 256 partial sums by wave shuffles + LDS in a fixed order.  46.7 -> ~5 us on B*64*64 x 3.
__global__ __launch_bounds__(256) void col_reduce_stage1_small(const float* __restrict__ a, const float* __restrict__ b,
                                                               int64_t rows, int C, int64_t rows_per_chunk,
                                                               float* __restrict__ part0, float* __restrict__ part1,
                                                               bool want1, int shift, const float* __restrict__ center) {
  __shared__ float s0[4][4], s1[4][4];
  const int64_t rbeg = (int64_t)blockIdx.y * rows_per_chunk;
  int64_t rend = rbeg + rows_per_chunk;
  if (rend > rows) rend = rows;
  float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
  float sh[4] = {0.f, 0.f, 0.f, 0.f}, ce[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (c < C) { if (shift) sh[c] = a[rbeg * C + c]; if (center) ce[c] = center[c]; }
  for (int64_t r = rbeg + threadIdx.x; r < rend; r += 256) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < C) {
        const float va = a[r * C + c] - sh[c];
        acc0[c] += va;
        if (want1) acc1[c] += va * (b ? b[r * C + c] - ce[c] : va);
      }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float v0 = wave_sum(acc0[c]), v1 = wave_sum(acc1[c]);
    if (lane == 0) { s0[wave][c] = v0; s1[wave][c] = v1; }
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    part0[(size_t)blockIdx.y * C + c] = (s0[0][c] + s0[1][c]) + (s0[2][c] + s0[3][c]);
    if (want1) part1[(size_t)blockIdx.y * C + c] = (s1[0][c] + s1[1][c]) + (s1[2][c] + s1[3][c]);
  }
}

// stage 2: one block per 64 columns; 4 row lanes walk the partials (fixed order), LDS joins them
__global__ __launch_bounds__(256) void col_reduce_stage2(const float* __restrict__ part0, const float* __restrict__ part1,
                                                         int nchunks, int C, float* __restrict__ out0,
                                                         float* __restrict__ out1, int accumulate) {
  __shared__ float s0[4][64], s1[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    for (int k = ty; k < nchunks; k += 4) {
      a0 += part0[(size_t)k * C + c];
      if (out1) a1 += part1[(size_t)k * C + c];
    }
  }
  s0[ty][tx] = a0;
  s1[ty][tx] = a1;
  __syncthreads();
  if (ty == 0 && c < C) {
    const float r0 = (s0[0][tx] + s0[1][tx]) + (s0[2][tx] + s0[3][tx]);
    out0[c] = accumulate ? out0[c] + r0 : r0;          // accumulate: sum straight into a gradient arena slot
    if (out1) {
      const float r1 = (s1[0][tx] + s1[1][tx]) + (s1[2][tx] + s1[3][tx]);
      out1[c] = accumulate ? out1[c] + r1 : r1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 STORAGE (BASELINE config 3, ABI v6): activation tensors may live in HBM as bf16 (t2i_dtype T2I_DT_BF16).  The vectorised
// kernels below are templated on H = "the activation inputs are bf16" and take every activation OUTPUT as a pair
// (fp32 pointer, bf16 pointer), either of which may be NULL: fp32 only (the fp32 path), both (fp32 tensor + bf16 twin, the
// round-2 "operand image" mode) or bf16 only (bf16 storage).  All arithmetic, statistics and partial sums stay fp32.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned aux_pk2(float lo, float hi);
template <bool H>
__device__ __forceinline__ float4 ld4(const void* p, size_t i4) {        // 4 consecutive elements starting at element 4 * i4
  if (H) {
    const uint2 u = reinterpret_cast<const uint2*>(p)[i4];
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xFFFF0000u));
  }
  return reinterpret_cast<const float4*>(p)[i4];
}
__device__ __forceinline__ void st4(float* y, void* yh, size_t i4, const float4 v);
template <bool H>
__device__ __forceinline__ float ld1(const void* p, size_t i) {
  if (H) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p)[i] << 16);
  return reinterpret_cast<const float*>(p)[i];
}

// 16-byte variants (C % 4 == 0, 16-byte aligned operands): a lane owns 4 adjacent columns, 16 lanes = 64 columns, 16 row
// lanes per workgroup; the row loop is unrolled so that several independent 16-byte loads are in flight per lane.
// (The scalar stage 1 above streamed 67 MB at 1.5 TB/s; this one is bound by HBM like the other elementwise kernels.)
template <bool WANT1, bool HAS_B, bool H = false>
__global__ __launch_bounds__(256) void col_reduce_stage1_v4(const void* __restrict__ a, const void* __restrict__ b,
                                                            int64_t rows, int C, int64_t rows_per_chunk,
                                                            float* __restrict__ part0, float* __restrict__ part1, int shift,
                                                            const float* __restrict__ center, int64_t rows_g = 0, int ncg = 0) {
  __shared__ float4 red[16][16];
  __shared__ float4 red1[WANT1 ? 16 : 1][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + tx * 4;
  // rows_g != 0: the rows are `rows / rows_g` groups of rows_g (the passes of a stacked batch), ncg chunks each; a chunk never
  // straddles groups and `center` is per group ([groups][C])
  int64_t rbeg = (int64_t)blockIdx.y * rows_per_chunk, rend = rbeg + rows_per_chunk;
  if (rows_g) {
    const int g = blockIdx.y / ncg, j = blockIdx.y - g * ncg;
    rbeg = (int64_t)g * rows_g + (int64_t)j * rows_per_chunk;
    rend = rbeg + rows_per_chunk;
    if (rend > (int64_t)(g + 1) * rows_g) rend = (int64_t)(g + 1) * rows_g;
    if (center) center += (size_t)g * C;
  }
  if (rend > rows) rend = rows;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    const float4 sh = shift ? ld4<H>(a, (size_t)(rbeg * C + c) >> 2) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 ce = (HAS_B && center) ? *reinterpret_cast<const float4*>(center + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int64_t r = rbeg + ty; r < rend; r += 16) {
      float4 v = ld4<H>(a, (size_t)(r * C + c) >> 2);
      v.x -= sh.x; v.y -= sh.y; v.z -= sh.z; v.w -= sh.w;
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      if (WANT1) {
        float4 w = v;
        if (HAS_B) { w = ld4<H>(b, (size_t)(r * C + c) >> 2); w.x -= ce.x; w.y -= ce.y; w.z -= ce.z; w.w -= ce.w; }
        acc1.x += v.x * w.x; acc1.y += v.y * w.y; acc1.z += v.z * w.z; acc1.w += v.w * w.w;
      }
    }
  }
  red[ty][tx] = acc;
  if (WANT1) red1[ty][tx] = acc1;
  __syncthreads();
  if (ty == 0 && c < C) {
    float4 s0 = red[0][tx];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 v = red[k][tx]; s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w; }
    *reinterpret_cast<float4*>(part0 + (size_t)blockIdx.y * C + c) = s0;
    if (WANT1) {
      float4 s1 = red1[0][tx];
#pragma unroll
      for (int k = 1; k < 16; ++k) { const float4 v = red1[k][tx]; s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w; }
      *reinterpret_cast<float4*>(part1 + (size_t)blockIdx.y * C + c) = s1;
    }
  }
}

__global__ __launch_bounds__(256) void col_reduce_stage2_v4(const float* __restrict__ part0, const float* __restrict__ part1,
                                                            int nchunks, int C, float* __restrict__ out0,
                                                            float* __restrict__ out1, int accumulate) {
  __shared__ float4 red[16][16], red1[16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + tx * 4;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    for (int k = ty; k < nchunks; k += 16) {
      const float4 v = *reinterpret_cast<const float4*>(part0 + (size_t)k * C + c);
      a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
      if (out1) {
        const float4 w = *reinterpret_cast<const float4*>(part1 + (size_t)k * C + c);
        a1.x += w.x; a1.y += w.y; a1.z += w.z; a1.w += w.w;
      }
    }
  }
  red[ty][tx] = a0;
  red1[ty][tx] = a1;
  __syncthreads();
  if (ty == 0 && c < C) {
    float4 s0 = red[0][tx], s1 = red1[0][tx];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 v = red[k][tx], w = red1[k][tx];
      s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
      s1.x += w.x; s1.y += w.y; s1.z += w.z; s1.w += w.w;
    }
    // outputs may be arbitrary arena slots (only 4-byte aligned): scalar stores
    const float r0[4] = {s0.x, s0.y, s0.z, s0.w}, r1[4] = {s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      out0[c + e] = accumulate ? out0[c + e] + r0[e] : r0[e];
      if (out1) out1[c + e] = accumulate ? out1[c + e] + r1[e] : r1[e];
    }
  }
}

static void col_reduce_plan(int64_t rows, int C, int* ctiles, int* nchunks, int64_t* rows_per_chunk) {
  *ctiles = (C + 63) / 64;
  int64_t want = tuning().colred_wgs / *ctiles;   // workgroups in flight (768 = 3 per CU in rounds 1-3)
  if (want < 1) want = 1;
  int64_t maxchunks = (rows + 63) / 64;       // at least 64 rows per chunk
  if (want > maxchunks) want = maxchunks;
  if (want > tuning().colred_cap) want = tuning().colred_cap;                 // keeps the second stage short
  if (want < 1) want = 1;
  *rows_per_chunk = (rows + want - 1) / want;
  *nchunks = (int)((rows + *rows_per_chunk - 1) / *rows_per_chunk);
  if (*nchunks < 1) *nchunks = 1;
}

hipError_t col_reduce_partials_launch(const float* part0, const float* part1, int chunks, int C, float* out0, float* out1,
                                      int accumulate, hipStream_t stream) {
  const int ct = (C + 63) / 64;
  const bool v4 = (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(part0) | reinterpret_cast<uintptr_t>(part1)) & 15) == 0;
  if (v4) hipLaunchKernelGGL(col_reduce_stage2_v4, dim3(ct), dim3(256), 0, stream, part0, part1, chunks, C, out0, out1, accumulate);
  else hipLaunchKernelGGL(col_reduce_stage2, dim3(ct), dim3(256), 0, stream, part0, part1, chunks, C, out0, out1, accumulate);
  return hipGetLastError();
}

size_t col_reduce_ws(int64_t rows, int C) {
  int ct, nc; int64_t rpc;
  col_reduce_plan(rows, C, &ct, &nc, &rpc);
  return (size_t)nc * C * 2 * sizeof(float);
}

// stage 1 of a column reduction into the workspace partials; shared by col_reduce_launch and bn_stats_launch
static void col_reduce_stage1_launch(const void* av, const void* bv, int64_t rows, int C, bool want1, int shift, const float* center,
                                     int ct, int nc, int64_t rpc, float* part0, float* part1, hipStream_t stream, bool in_bf16 = false) {
  const float* a = reinterpret_cast<const float*>(av);
  const float* b = reinterpret_cast<const float*>(bv);
  const bool v4 = (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(part0) |
                                     reinterpret_cast<uintptr_t>(center)) & 15) == 0;
  if (in_bf16) {                 // bf16 storage: the C API guarantees C % 4 == 0 and alignment
    if (!want1)
      hipLaunchKernelGGL((col_reduce_stage1_v4<false, false, true>), dim3(ct, nc), dim3(256), 0, stream, av, bv, rows, C, rpc, part0, part1, shift, center);
    else if (bv)
      hipLaunchKernelGGL((col_reduce_stage1_v4<true, true, true>), dim3(ct, nc), dim3(256), 0, stream, av, bv, rows, C, rpc, part0, part1, shift, center);
    else
      hipLaunchKernelGGL((col_reduce_stage1_v4<true, false, true>), dim3(ct, nc), dim3(256), 0, stream, av, bv, rows, C, rpc, part0, part1, shift, center);
  } else if (v4) {
    if (!want1)
      hipLaunchKernelGGL((col_reduce_stage1_v4<false, false, false>), dim3(ct, nc), dim3(256), 0, stream, av, bv, rows, C, rpc, part0, part1, shift, center);
    else if (b)
      hipLaunchKernelGGL((col_reduce_stage1_v4<true, true, false>), dim3(ct, nc), dim3(256), 0, stream, av, bv, rows, C, rpc, part0, part1, shift, center);
    else
      hipLaunchKernelGGL((col_reduce_stage1_v4<true, false, false>), dim3(ct, nc), dim3(256), 0, stream, av, bv, rows, C, rpc, part0, part1, shift, center);
  } else if (C <= 4) {
    hipLaunchKernelGGL(col_reduce_stage1_small, dim3(1, nc), dim3(256), 0, stream, a, b, rows, C, rpc, part0, part1, want1, shift, center);
  } else {
    hipLaunchKernelGGL(col_reduce_stage1, dim3(ct, nc), dim3(256), 0, stream, a, b, rows, C, rpc, part0, part1, want1, shift, center);
  }
}

hipError_t col_reduce_launch(const void* a, const void* b, const float* center, int64_t rows, int C, float* out0, float* out1,
                             int accumulate, void* ws, hipStream_t stream, bool in_bf16) {
  int ct, nc; int64_t rpc;
  col_reduce_plan(rows, C, &ct, &nc, &rpc);
  float* part0 = reinterpret_cast<float*>(ws);
  float* part1 = part0 + (size_t)nc * C;
  col_reduce_stage1_launch(a, b, rows, C, out1 != nullptr, 0, center, ct, nc, rpc, part0, part1, stream, in_bf16);
  const bool v4 = (C & 3) == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0;
  if (v4)
    hipLaunchKernelGGL(col_reduce_stage2_v4, dim3(ct), dim3(256), 0, stream, part0, (const float*)(out1 ? part1 : nullptr), nc, C,
                       out0, out1, accumulate);
  else
    hipLaunchKernelGGL(col_reduce_stage2, dim3(ct), dim3(256), 0, stream, part0, part1, nc, C, out0, out1, accumulate);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Batch-norm statistics, numerically stable: (sum, M2 = sum (x - mean)^2) per column.
//   stage 1 (col_reduce_stage1*, shift = 1): per row chunk k the sums S1 = sum(x - s_k), S2 = sum((x - s_k)^2) about the
//            chunk's first row s_k = x[rbeg_k, c];
//   stage 2: chunk k becomes the aggregate (n_k, mean_k = s_k + S1/n_k, M2_k = S2 - S1^2/n_k) — the subtraction is
//            harmless because s_k is one of the chunk's own samples — and aggregates are merged pairwise with Chan's
//            update (n, mean, M2) + (n', mean', M2') -> M2 + M2' + (mean' - mean)^2 n n'/(n + n'), in a fixed order.
//   TILES = true: the per-tile partials of a conv epilogue instead (t2i_conv2d_fwd_stats): part0 = the tile's column sums,
//            part1 = its M2 about the tile's own mean, n_k = rows of tile k.
// The two-moment formula var = sum(x^2)/n - mean^2 (round 1) is exact in real arithmetic but loses mean^2/var digits in
// fp32: a rank-2 batch norm over a batch of 2 whose two values differ by 1e-3 of their size comes out with a 40 % wrong
// variance, and the StackGAN Stage-II image (batch 2, ~40 batch-normed layers) sat 1e-3 from float64 for this reason alone.
// ---------------------------------------------------------------------------------------------------------------
struct Agg { float n, mean, m2; };

// Optional tail of the statistics' second stage: what bn_finalize_kernel does, done by the thread that already holds the
// column's (n, mean, M2) — one launch less per batch norm (20 per wgancls iteration).  gamma == nullptr: not requested.
struct BnFin {
  const float* gamma; const float* beta; float eps, decay;
  float* mean; float* rstd; float* scale; float* shift; float* mmean; float* mvar;
  int updates;       // how many times the moving averages take this batch's statistics (1; 2 when one evaluation stands for two identical runs)
  size_t upd_limit;  // stacked batches: only groups whose offset group * C lies below this move the moving averages (default: all)
};

__device__ __forceinline__ void bn_fin_col(const BnFin& f, int c, float n, float mu, float m2, size_t goff = 0) {
  // goff = group * C: the per-group outputs of a stacked batch; gamma / beta / moving averages are the layer's own (the moving averages
  // move once per group, in group order: the caller walks the groups sequentially in one thread)
  const float var = fmaxf(m2, 0.f) / n;                      // biased batch variance
  const float rs = rsqrtf(var + f.eps);
  f.mean[goff + c] = mu;
  f.rstd[goff + c] = rs;
  const float sc = f.gamma[c] * rs;
  f.scale[goff + c] = sc;
  f.shift[goff + c] = f.beta[c] - mu * sc;
  if (f.mmean && goff < f.upd_limit) {                       // TF fused BN: the moving variance takes the UNBIASED estimate
    const float unb = var * (n / fmaxf(n - 1.f, 1.f));
    float mm = f.mmean[c], mv = f.mvar[c];
    for (int u = 0; u < f.updates; ++u) {                    // sequential exponential-average steps, as that many runs of the update op would take
      mm = f.decay * mm + (1.f - f.decay) * mu;
      mv = f.decay * mv + (1.f - f.decay) * unb;
    }
    f.mmean[c] = mm;
    f.mvar[c] = mv;
  }
}

__device__ __forceinline__ Agg agg_merge(Agg a, Agg b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  Agg r;
  r.n = a.n + b.n;
  const float delta = b.mean - a.mean;
  const float f = b.n / r.n;
  r.mean = a.mean + delta * f;
  r.m2 = a.m2 + b.m2 + delta * delta * a.n * f;
  return r;
}

template <bool TILES>
__global__ __launch_bounds__(256) void bn_stats_stage2(const float* __restrict__ part0, const float* __restrict__ part1,
                                                       const float* __restrict__ x, int nchunks, int64_t rows, int64_t rows_per_chunk,
                                                       int C, float* __restrict__ sum, float* __restrict__ m2, BnFin fin) {
  __shared__ float sn[4][64], sm[4][64], sq[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  Agg a = {0.f, 0.f, 0.f};
  if (c < C) {
    for (int k = ty; k < nchunks; k += 4) {
      const int64_t rbeg = (int64_t)k * rows_per_chunk;
      int64_t rend = rbeg + rows_per_chunk;
      if (rend > rows) rend = rows;
      Agg b;
      b.n = (float)(rend - rbeg);
      const float p0 = part0[(size_t)k * C + c], p1 = part1[(size_t)k * C + c];
      if (TILES) {
        b.mean = p0 / b.n;
        b.m2 = p1;
      } else {
        const float d = p0 / b.n;
        b.mean = x[rbeg * C + c] + d;
        b.m2 = fmaxf(p1 - p0 * d, 0.f);
      }
      a = agg_merge(a, b);
    }
  }
  sn[ty][tx] = a.n; sm[ty][tx] = a.mean; sq[ty][tx] = a.m2;
  __syncthreads();
  if (ty == 0 && c < C) {
    Agg r = {sn[0][tx], sm[0][tx], sq[0][tx]};
#pragma unroll
    for (int k = 1; k < 4; ++k) r = agg_merge(r, Agg{sn[k][tx], sm[k][tx], sq[k][tx]});
    if (sum) { sum[c] = r.mean * r.n; m2[c] = r.m2; }
    if (fin.gamma) bn_fin_col(fin, c, r.n, r.mean, r.m2);
  }
}

// 16-byte variant (C % 4 == 0, aligned): a lane owns 4 adjacent columns, 16 lanes = 64 columns, 16 chunk lanes per workgroup
// (the 4-lane scalar version above walked up to 48 chunks per lane with three dependent loads each: 10-20 us per launch, 20
// launches per iteration)
struct Agg4 { float n; float4 mean, m2; };

__device__ __forceinline__ Agg4 agg4_merge(const Agg4& a, const Agg4& b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  Agg4 r;
  r.n = a.n + b.n;
  const float f = b.n / r.n, g = a.n * f;
  const float4 d = make_float4(b.mean.x - a.mean.x, b.mean.y - a.mean.y, b.mean.z - a.mean.z, b.mean.w - a.mean.w);
  r.mean = make_float4(a.mean.x + d.x * f, a.mean.y + d.y * f, a.mean.z + d.z * f, a.mean.w + d.w * f);
  r.m2 = make_float4(a.m2.x + b.m2.x + d.x * d.x * g, a.m2.y + b.m2.y + d.y * d.y * g, a.m2.z + b.m2.z + d.z * d.z * g,
                     a.m2.w + b.m2.w + d.w * d.w * g);
  return r;
}

// 16 columns (4 lanes x float4) per workgroup, 64 chunk lanes: a lane merges its nchunks/64 chunks (<= 3), then the 64 lane
// aggregates are merged 4 : 1 three times through LDS — a dependent chain of ~12 Chan merges (each a division) instead of the 28 of
// the first version (16 chunk lanes x 12 chunks, then 16 serial merges by one thread: 13 us per launch, 20 launches per iteration).
// The merge order is fixed, so results are repeatable; they differ from the first version's in the last bits.
template <bool TILES, bool H = false>
__global__ __launch_bounds__(256) void bn_stats_stage2_v4(const float* __restrict__ part0, const float* __restrict__ part1,
                                                          const void* __restrict__ x, int nchunks, int64_t rows,
                                                          int64_t rows_per_chunk, int C, float* __restrict__ sum, float* __restrict__ m2,
                                                          BnFin fin, int groups = 1) {
  // groups > 1: `rows` rows and `nchunks` chunks PER GROUP (chunk index g * nchunks + k, rows g * rows ...): the groups are walked one
  // after the other by the same threads, so the moving averages move once per group in group order; outputs are [groups][C]
  __shared__ float sn[64][4];
  __shared__ float4 sm[64][4], sq[64][4];
  const int tx = threadIdx.x & 3, ty = threadIdx.x >> 2;
  const int c = blockIdx.x * 16 + tx * 4;
  for (int g = 0; g < groups; ++g) {
    Agg4 a;
    a.n = 0.f; a.mean = make_float4(0.f, 0.f, 0.f, 0.f); a.m2 = a.mean;
    const int64_t row0 = (int64_t)g * rows;
    if (c < C) {
      for (int k = ty; k < nchunks; k += 64) {
        const int64_t rbeg = (int64_t)k * rows_per_chunk;
        int64_t rend = rbeg + rows_per_chunk;
        if (rend > rows) rend = rows;
        Agg4 b;
        b.n = (float)(rend - rbeg);
        const size_t kk = (size_t)g * nchunks + k;
        const float4 p0 = *reinterpret_cast<const float4*>(part0 + kk * C + c);
        const float4 p1 = *reinterpret_cast<const float4*>(part1 + kk * C + c);
        const float inv = 1.f / b.n;
        if (TILES) {
          b.mean = make_float4(p0.x * inv, p0.y * inv, p0.z * inv, p0.w * inv);
          b.m2 = p1;
        } else {
          const float4 s = ld4<H>(x, (size_t)((row0 + rbeg) * C + c) >> 2);
          const float4 d = make_float4(p0.x * inv, p0.y * inv, p0.z * inv, p0.w * inv);
          b.mean = make_float4(s.x + d.x, s.y + d.y, s.z + d.z, s.w + d.w);
          b.m2 = make_float4(fmaxf(p1.x - p0.x * d.x, 0.f), fmaxf(p1.y - p0.y * d.y, 0.f), fmaxf(p1.z - p0.z * d.z, 0.f),
                             fmaxf(p1.w - p0.w * d.w, 0.f));
        }
        a = agg4_merge(a, b);
      }
    }
    // 64 -> 16 -> 4 -> 1 lanes; lane ty of a level merges entries 4 ty .. 4 ty + 3 of the level below, in that order
#pragma unroll
    for (int width = 64; width > 1; width >>= 2) {
      __syncthreads();
      if (ty < width) { sn[ty][tx] = a.n; sm[ty][tx] = a.mean; sq[ty][tx] = a.m2; }
      __syncthreads();
      if (ty < (width >> 2)) {
        a.n = sn[4 * ty][tx]; a.mean = sm[4 * ty][tx]; a.m2 = sq[4 * ty][tx];
#pragma unroll
        for (int j = 1; j < 4; ++j) {
          Agg4 b;
          b.n = sn[4 * ty + j][tx]; b.mean = sm[4 * ty + j][tx]; b.m2 = sq[4 * ty + j][tx];
          a = agg4_merge(a, b);
        }
      }
    }
    if (ty == 0 && c < C) {
      const Agg4 r = a;
      const float mu[4] = {r.mean.x, r.mean.y, r.mean.z, r.mean.w};
      const float mo[4] = {r.m2.x, r.m2.y, r.m2.z, r.m2.w};
      const size_t goff = (size_t)g * C;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (sum) { sum[goff + c + e] = mu[e] * r.n; m2[goff + c + e] = mo[e]; }
        if (fin.gamma) bn_fin_col(fin, c + e, r.n, mu[e], mo[e], goff);
      }
    }
  }
}

template <bool TILES>
static void bn_stats_stage2_launch(const float* part0, const float* part1, const void* xv, int nc, int64_t rows, int64_t rpc, int C,
                                   float* sum, float* m2, const BnFin& fin, hipStream_t stream, bool x_bf16 = false) {
  const float* x = reinterpret_cast<const float*>(xv);
  const bool v4 = (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(part0) | reinterpret_cast<uintptr_t>(part1) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
  if (x_bf16)
    hipLaunchKernelGGL((bn_stats_stage2_v4<TILES, true>), dim3((C + 15) / 16), dim3(256), 0, stream, part0, part1, xv, nc, rows, rpc, C, sum, m2, fin);
  else if (v4)
    hipLaunchKernelGGL((bn_stats_stage2_v4<TILES, false>), dim3((C + 15) / 16), dim3(256), 0, stream, part0, part1, xv, nc, rows, rpc, C, sum, m2, fin);
  else
    hipLaunchKernelGGL(bn_stats_stage2<TILES>, dim3((C + 63) / 64), dim3(256), 0, stream, part0, part1, x, nc, rows, rpc, C, sum, m2, fin);
}

static BnFin make_fin(const float* gamma, const float* beta, float eps, float decay, float* mean, float* rstd, float* scale, float* shift,
                      float* mm, float* mv, int updates = 1) {
  BnFin f;
  f.gamma = gamma; f.beta = beta; f.eps = eps; f.decay = decay;
  f.mean = mean; f.rstd = rstd; f.scale = scale; f.shift = shift; f.mmean = mm; f.mvar = mv;
  f.updates = updates < 1 ? 1 : updates;
  f.upd_limit = ~(size_t)0;
  return f;
}

// gamma == nullptr: statistics only (sum, m2); else the batch norm's mean / rstd / scale / shift (+ moving averages) as well
hipError_t bn_stats_launch(const void* x, int64_t rows, int C, float* sum, float* m2, const float* gamma, const float* beta, float eps,
                           float decay, float* mean, float* rstd, float* scale, float* shift, float* mm, float* mv, void* ws,
                           hipStream_t stream, bool x_bf16) {
  int ct, nc; int64_t rpc;
  col_reduce_plan(rows, C, &ct, &nc, &rpc);
  float* part0 = reinterpret_cast<float*>(ws);
  float* part1 = part0 + (size_t)nc * C;
  col_reduce_stage1_launch(x, nullptr, rows, C, true, 1, nullptr, ct, nc, rpc, part0, part1, stream, x_bf16);
  bn_stats_stage2_launch<false>(part0, part1, x, nc, rows, rpc, C, sum, m2, make_fin(gamma, beta, eps, decay, mean, rstd, scale, shift, mm, mv),
                                stream, x_bf16);
  return hipGetLastError();
}

hipError_t bn_stats_tiles_launch(const float* part_sum, const float* part_m2, int chunks, int tile_rows, int64_t rows, int C, float* sum,
                                 float* m2, const float* gamma, const float* beta, float eps, float decay, float* mean, float* rstd,
                                 float* scale, float* shift, float* mm, float* mv, hipStream_t stream) {
  bn_stats_stage2_launch<true>(part_sum, part_m2, (const float*)nullptr, chunks, rows, (int64_t)tile_rows, C, sum, m2,
                               make_fin(gamma, beta, eps, decay, mean, rstd, scale, shift, mm, mv), stream);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// batch norm
// ---------------------------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ m2, float n, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float decay, float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mmean,
                                   float* __restrict__ mvar) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mu = sum[c] / n;
  const float var = fmaxf(m2[c], 0.f) / n;   // biased batch variance from the CENTERED second moment (bn_stats_*)
  const float rs = rsqrtf(var + eps);
  mean[c] = mu;
  rstd[c] = rs;
  const float sc = gamma[c] * rs;
  scale[c] = sc;
  shift[c] = beta[c] - mu * sc;
  if (mmean) {  // TF fused BN: the moving variance takes the UNBIASED estimate
    const float unb = var * (n / fmaxf(n - 1.f, 1.f));
    mmean[c] = decay * mmean[c] + (1.f - decay) * mu;
    mvar[c] = decay * mvar[c] + (1.f - decay) * unb;
  }
}

// two floats -> packed bf16 pair (round to nearest even): the bf16 twins of activation tensors (t2i_output_image)
__device__ __forceinline__ unsigned aux_pk2(float lo, float hi) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 h2 __attribute__((ext_vector_type(2)));
  f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
}
__device__ __forceinline__ void st4(float* y, void* yh, size_t i4, const float4 v) {     // fp32 and / or bf16 (RNE) copy of 4 elements
  if (y) reinterpret_cast<float4*>(y)[i4] = v;
  if (yh) reinterpret_cast<uint2*>(yh)[i4] = make_uint2(aux_pk2(v.x, v.y), aux_pk2(v.z, v.w));
}

template <bool VEC, bool H = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const void* __restrict__ xv, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, size_t n, int C, int act,
                                                       float alpha, float* __restrict__ y, uint2* __restrict__ yh, size_t per_group4 = 0) {
  const float* x = reinterpret_cast<const float*>(xv);
  if (VEC) {
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
      // per_group4 != 0: element quads per group of a stacked batch; scale / shift are [groups][C]
      const int c = (int)((i * 4) % (size_t)C) + (per_group4 ? (int)(i / per_group4) * C : 0);
      float4 v = ld4<H>(xv, i);
      const float4 sc = *reinterpret_cast<const float4*>(scale + c);
      const float4 sh = *reinterpret_cast<const float4*>(shift + c);
      v.x = apply_act(v.x * sc.x + sh.x, act, alpha);
      v.y = apply_act(v.y * sc.y + sh.y, act, alpha);
      v.z = apply_act(v.z * sc.z + sh.z, act, alpha);
      v.w = apply_act(v.w * sc.w + sh.w, act, alpha);
      st4(y, yh, i, v);
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      const int c = (int)(i % (size_t)C);
      y[i] = apply_act(x[i] * scale[c] + shift[c], act, alpha);
    }
  }
}

__global__ void bn_bwd_coef_kernel(const float* __restrict__ mean, const float* __restrict__ rstd,
                                   const float* __restrict__ gamma, const float* __restrict__ sum_dy,
                                   const float* __restrict__ sum_dy_x, float n, int C, float* __restrict__ dgamma,
                                   float* __restrict__ dbeta, float* __restrict__ k_dy, float* __restrict__ k_x,
                                   float* __restrict__ k_0, int accumulate) {
  // dx = g*rs*(dy - sdy/n - xhat*sdyxh/n), xhat = (x-mu)*rs  ==  k_dy*dy + k_x*x + k_0
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mu = mean[c], rs = rstd[c], g = gamma[c], sdy = sum_dy[c];
  const float sdyxh = rs * sum_dy_x[c];          // sum_dy_x = sum dy * (x - mean): centered by the reduction that produced it
  dgamma[c] = accumulate ? dgamma[c] + sdyxh : sdyxh;
  dbeta[c] = accumulate ? dbeta[c] + sdy : sdy;
  const float grs = g * rs;
  k_dy[c] = grs;
  k_x[c] = -grs * rs * sdyxh / n;
  k_0[c] = -grs * sdy / n + grs * rs * mu * sdyxh / n;
}

// second stage of the two backward reductions (sum dy, sum dy*(x - mean)) fused with bn_bwd_coef_kernel: the thread that
// finishes a column's sums turns them into dgamma, dbeta and the three coefficients of dx = k_dy*dy + k_x*x + k_0
__global__ __launch_bounds__(256) void bn_bwd_stage2_coef_v4(const float* __restrict__ part0, const float* __restrict__ part1, int nchunks,
                                                             int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, float n, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ k_dy, float* __restrict__ k_x,
                                                             float* __restrict__ k_0, int accumulate, int groups = 1) {
  // groups > 1 (a stacked batch): nchunks partials, n rows, mean / rstd [C] and one coefficient triple [3][C] PER GROUP (chunk g * nchunks + k,
  // mean + g C, k_* + g 3 C); dgamma / dbeta are the sums over the groups, formed in group order by the column's own thread
  __shared__ float4 red[16][16], red1[16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + tx * 4;
  float tg[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
  for (int g = 0; g < groups; ++g) {
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
      for (int k = ty; k < nchunks; k += 16) {
        const size_t kk = (size_t)g * nchunks + k;
        const float4 v = *reinterpret_cast<const float4*>(part0 + kk * C + c);
        const float4 w = *reinterpret_cast<const float4*>(part1 + kk * C + c);
        a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
        a1.x += w.x; a1.y += w.y; a1.z += w.z; a1.w += w.w;
      }
    }
    if (g) __syncthreads();
    red[ty][tx] = a0;
    red1[ty][tx] = a1;
    __syncthreads();
    if (ty == 0 && c < C) {
      float4 s0 = red[0][tx], s1 = red1[0][tx];
#pragma unroll
      for (int k = 1; k < 16; ++k) {
        const float4 v = red[k][tx], w = red1[k][tx];
        s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
        s1.x += w.x; s1.y += w.y; s1.z += w.z; s1.w += w.w;
      }
      const float sd[4] = {s0.x, s0.y, s0.z, s0.w}, sx[4] = {s1.x, s1.y, s1.z, s1.w};
      const size_t goff = (size_t)g * C, koff = (size_t)g * 3 * C;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int cc = c + e;
        const float mu = mean[goff + cc], rs = rstd[goff + cc], gm = gamma[cc], sdy = sd[e];
        const float sdyxh = rs * sx[e];
        tg[e] = g ? tg[e] + sdyxh : sdyxh;
        tb[e] = g ? tb[e] + sdy : sdy;
        const float grs = gm * rs;
        k_dy[koff + cc] = grs;
        k_x[koff + cc] = -grs * rs * sdyxh / n;
        k_0[koff + cc] = -grs * sdy / n + grs * rs * mu * sdyxh / n;
      }
    }
  }
  if (ty == 0 && c < C) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int cc = c + e;
      dgamma[cc] = accumulate ? dgamma[cc] + tg[e] : tg[e];
      dbeta[cc] = accumulate ? dbeta[cc] + tb[e] : tb[e];
    }
  }
}

// HD / HX: dy resp. x is bf16 (in bf16 storage the masked gradient g of bn_bwd_fused is a bf16 tensor like everything else)
template <bool VEC, bool HD = false, bool HX = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const void* __restrict__ dyv, const void* __restrict__ xv,
                                                           const float* __restrict__ k_dy, const float* __restrict__ k_x,
                                                           const float* __restrict__ k_0, size_t n, int C,
                                                           float* __restrict__ dx, uint2* __restrict__ dxh, size_t per_group4 = 0) {
  const float* dy = reinterpret_cast<const float*>(dyv);
  const float* x = reinterpret_cast<const float*>(xv);
  if (VEC) {
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
      // per_group4 != 0: element quads per group of a stacked batch; one coefficient triple [3][C] per group
      const int c = (int)((i * 4) % (size_t)C) + (per_group4 ? (int)(i / per_group4) * 3 * C : 0);
      const float4 d = ld4<HD>(dyv, i);
      const float4 v = ld4<HX>(xv, i);
      const float4 a = *reinterpret_cast<const float4*>(k_dy + c);
      const float4 b = *reinterpret_cast<const float4*>(k_x + c);
      const float4 e = *reinterpret_cast<const float4*>(k_0 + c);
      float4 o;
      o.x = a.x * d.x + b.x * v.x + e.x;
      o.y = a.y * d.y + b.y * v.y + e.y;
      o.z = a.z * d.z + b.z * v.z + e.z;
      o.w = a.w * d.w + b.w * v.w + e.w;
      st4(dx, dxh, i, o);
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      const int c = (int)(i % (size_t)C);
      dx[i] = k_dy[c] * dy[i] + k_x[c] * x[i] + k_0[c];
    }
  }
}

hipError_t bn_finalize_launch(const float* sum, const float* sumsq, int64_t n, int C, const float* gamma,
                              const float* beta, float eps, float decay, float* mean, float* rstd, float* scale,
                              float* shift, float* mm, float* mv, hipStream_t stream) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sum, sumsq, (float)n, C, gamma,
                     beta, eps, decay, mean, rstd, scale, shift, mm, mv);
  return hipGetLastError();
}

hipError_t bn_apply_launch(const void* x, const float* scale, const float* shift, int64_t rows, int C, int act,
                           float alpha, float* y, hipStream_t stream, void* y_h, bool x_bf16) {
  const bool al = C > 0;          // the C API passes -C when some pointer is not 16-byte aligned
  if (C < 0) C = -C;
  const size_t n = (size_t)rows * C;
  if (x_bf16)                     // bf16 storage (the C API has checked alignment and C % 4)
    hipLaunchKernelGGL((bn_apply_kernel<true, true>), dim3(ew_blocks(n >> 2)), dim3(256), 0, stream, x, scale, shift, n, C, act,
                       alpha, y, reinterpret_cast<uint2*>(y_h));
  else if (al && (C & 3) == 0)
    hipLaunchKernelGGL((bn_apply_kernel<true, false>), dim3(ew_blocks(n >> 2)), dim3(256), 0, stream, x, scale, shift, n, C, act,
                       alpha, y, reinterpret_cast<uint2*>(y_h));
  else
    hipLaunchKernelGGL((bn_apply_kernel<false, false>), dim3(ew_blocks(n)), dim3(256), 0, stream, x, scale, shift, n, C, act,
                       alpha, y, (uint2*)nullptr);
  return hipGetLastError();
}

// dgamma/dbeta double as scratch-free outputs; k_* coefficients live in the caller-visible dgamma-sized buffers
hipError_t bn_bwd_launch(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                         const float* sum_dy, const float* sum_dy_x, int64_t rows, int C, float* dx, float* dgamma,
                         float* dbeta, float* coef /* 3*C floats */, int accumulate, hipStream_t stream) {
  const bool al = C > 0;
  if (C < 0) C = -C;
  float* k_dy = coef; float* k_x = coef + C; float* k_0 = coef + 2 * (size_t)C;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, mean, rstd, gamma, sum_dy,
                     sum_dy_x, (float)rows, C, dgamma, dbeta, k_dy, k_x, k_0, accumulate);
  const size_t n = (size_t)rows * C;
  if (al && (C & 3) == 0)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false, false>), dim3(ew_blocks(n >> 2)), dim3(256), 0, stream, dy, x, k_dy, k_x, k_0,
                       n, C, dx, (uint2*)nullptr);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<false, false, false>), dim3(ew_blocks(n)), dim3(256), 0, stream, dy, x, k_dy, k_x, k_0, n,
                       C, dx, (uint2*)nullptr);
  return hipGetLastError();
}

// Whole batch-norm backward in three launches (C % 4 == 0, 16-byte aligned): [activation backward + the two reductions,
// stage 1] -> [stage 2 + coefficients] -> [dx].  y == nullptr: no activation in front (gy is used as it is).  gmask: the
// masked gradient dy*act'(y), written by stage 1 and read by the last launch (caller's buffer, may alias nothing else).
hipError_t bn_bwd_fused_launch(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma,
                               int64_t rows, int C, int act, float alpha, void* gmask, float* dx, float* dgamma, float* dbeta,
                               int accumulate, void* ws, hipStream_t stream, void* dx_h, bool in_bf16);

// ---------------------------------------------------------------------------------------------------------------
// elementwise (float4 body + scalar tail handled by the same kernel)
// ---------------------------------------------------------------------------------------------------------------
enum { EW_ACT_FWD = 0, EW_ACT_BWD = 1, EW_ADD_ACT = 2, EW_AXPBY = 3 };

__device__ __forceinline__ float act_grad_from_output(float y, int act, float alpha) {
  switch (act) {
    case T2I_ACT_LRELU: return y > 0.f ? 1.f : alpha;
    case T2I_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case T2I_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

template <int OP>
__device__ __forceinline__ float ew_op(float a, float b, int act, float alpha, float beta) {
  if (OP == EW_ACT_FWD) return apply_act(a, act, alpha);
  if (OP == EW_ACT_BWD) return a * act_grad_from_output(b, act, alpha);   // a = dy, b = y
  if (OP == EW_ADD_ACT) return apply_act(a + b, act, alpha);
  return alpha * a + beta * b;                                            // EW_AXPBY
}

template <int OP, bool HAS_B, bool H = false>
__global__ __launch_bounds__(256) void ew_kernel(const void* __restrict__ av, const void* __restrict__ bv, size_t n,
                                                 size_t n4, int act, float alpha, float beta, float* __restrict__ y,
                                                 uint2* __restrict__ yh) {
  const float* a = reinterpret_cast<const float*>(av);
  const float* b = reinterpret_cast<const float*>(bv);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = t; i < n4; i += stride) {
    const float4 va = ld4<H>(av, i);
    float4 vb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_B) vb = ld4<H>(bv, i);
    float4 o;
    o.x = ew_op<OP>(va.x, vb.x, act, alpha, beta);
    o.y = ew_op<OP>(va.y, vb.y, act, alpha, beta);
    o.z = ew_op<OP>(va.z, vb.z, act, alpha, beta);
    o.w = ew_op<OP>(va.w, vb.w, act, alpha, beta);
    st4(y, yh, i, o);
  }
  if (!H)      // scalar tail: fp32 tensors only (bf16 storage requires n % 4 == 0)
    for (size_t i = (n4 << 2) + t; i < n; i += stride) y[i] = ew_op<OP>(a[i], HAS_B ? b[i] : 0.f, act, alpha, beta);
}

// fused: dx = dy * act'(y)  AND  colsum[c] = sum_r dx[r,c]  — the activation backward and the bias gradient of a conv layer
// read the same tensor; one pass instead of two (float4 per lane, 16 lanes = 64 columns, 16 row lanes per workgroup).
template <bool SECOND, bool H = false>   // SECOND: also part1[c] = sum_r dx[r,c] * x2[r,c]  (batch-norm backward needs sum dy and sum dy*x)
__global__ __launch_bounds__(256) void act_bwd_colsum_stage1(const void* __restrict__ dy, const void* __restrict__ y,
                                                             const void* __restrict__ x2, const float* __restrict__ center,
                                                             int64_t rows, int C, int64_t rows_per_chunk, int act, float alpha,
                                                             float* __restrict__ dx, float* __restrict__ part,
                                                             float* __restrict__ part1, unsigned short* __restrict__ dxh,
                                                             int64_t rows_g = 0, int ncg = 0) {
  __shared__ float4 red[16][16];
  __shared__ float4 red1[SECOND ? 16 : 1][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + tx * 4;
  int64_t rbeg = (int64_t)blockIdx.y * rows_per_chunk, rend = rbeg + rows_per_chunk;
  if (rows_g) {                          // grouped rows (see col_reduce_stage1_v4): chunks per group, `center` per group
    const int g = blockIdx.y / ncg, j = blockIdx.y - g * ncg;
    rbeg = (int64_t)g * rows_g + (int64_t)j * rows_per_chunk;
    rend = rbeg + rows_per_chunk;
    if (rend > (int64_t)(g + 1) * rows_g) rend = (int64_t)(g + 1) * rows_g;
    if (center) center += (size_t)g * C;
  }
  if (rend > rows) rend = rows;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    const float4 ce = (SECOND && center) ? *reinterpret_cast<const float4*>(center + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
    for (int64_t r = rbeg + ty; r < rend; r += 16) {
      const size_t e4 = (size_t)(r * C + c) >> 2;
      const float4 g = ld4<H>(dy, e4);
      const float4 o = ld4<H>(y, e4);
      float4 d;
      d.x = g.x * act_grad_from_output(o.x, act, alpha); d.y = g.y * act_grad_from_output(o.y, act, alpha);
      d.z = g.z * act_grad_from_output(o.z, act, alpha); d.w = g.w * act_grad_from_output(o.w, act, alpha);
      st4(dx, dxh, e4, d);
      // bf16 storage: the bias / gamma / beta partial sums below take the UNROUNDED fp32 d, while dx is stored rounded to bf16 and the
      // convolutions behind it read the rounded values — the reductions are the more accurate of the two, and differ from
      // reduce(stored dx) by at most 2^-9 relative per element (random sign): sqrt(rows) * 2^-9 of an element's magnitude per column,
      // far inside config 3's tolerance.  tests/test_storage_gpu.py pins it (reductions bit-equal to the fp32-tensor path; dx one rounding).
      acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
      if (SECOND) {
        const float4 v = ld4<H>(x2, e4);
        acc1.x += d.x * (v.x - ce.x); acc1.y += d.y * (v.y - ce.y); acc1.z += d.z * (v.z - ce.z); acc1.w += d.w * (v.w - ce.w);
      }
    }
  }
  red[ty][tx] = acc;
  if (SECOND) red1[ty][tx] = acc1;
  __syncthreads();
  if (ty == 0 && c < C) {
    float4 s = red[0][tx];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 v = red[k][tx]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.y * C + c) = s;
    if (SECOND) {
      float4 s1 = red1[0][tx];
#pragma unroll
      for (int k = 1; k < 16; ++k) { const float4 v = red1[k][tx]; s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w; }
      *reinterpret_cast<float4*>(part1 + (size_t)blockIdx.y * C + c) = s1;
    }
  }
}

hipError_t act_bwd_colsum_launch(const void* dy, const void* y, const void* x2, const float* center, int64_t rows, int C, int act,
                                 float alpha, float* dx, float* sum0, float* sum1, int accumulate, void* ws, hipStream_t stream, void* dx_h,
                                 bool in_bf16) {
  unsigned short* dxh = reinterpret_cast<unsigned short*>(dx_h);
  int ct, nc; int64_t rpc;
  col_reduce_plan(rows, C, &ct, &nc, &rpc);
  float* part = reinterpret_cast<float*>(ws);
  float* part1 = part + (size_t)nc * C;
  const bool second = x2 != nullptr && sum1 != nullptr;
  if (second && in_bf16)
    hipLaunchKernelGGL((act_bwd_colsum_stage1<true, true>), dim3(ct, nc), dim3(256), 0, stream, dy, y, x2, center, rows, C, rpc, act, alpha,
                       dx, part, part1, dxh);
  else if (second)
    hipLaunchKernelGGL((act_bwd_colsum_stage1<true, false>), dim3(ct, nc), dim3(256), 0, stream, dy, y, x2, center, rows, C, rpc, act, alpha,
                       dx, part, part1, dxh);
  else if (in_bf16)
    hipLaunchKernelGGL((act_bwd_colsum_stage1<false, true>), dim3(ct, nc), dim3(256), 0, stream, dy, y, x2, center, rows, C, rpc, act, alpha,
                       dx, part, part1, dxh);
  else
    hipLaunchKernelGGL((act_bwd_colsum_stage1<false, false>), dim3(ct, nc), dim3(256), 0, stream, dy, y, x2, center, rows, C, rpc, act, alpha,
                       dx, part, part1, dxh);
  hipLaunchKernelGGL(col_reduce_stage2_v4, dim3(ct), dim3(256), 0, stream, part, (const float*)(second ? part1 : nullptr), nc, C,
                     sum0, second ? sum1 : (float*)nullptr, accumulate);
  return hipGetLastError();
}

hipError_t bn_bwd_fused_launch(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma,
                               int64_t rows, int C, int act, float alpha, void* gmask, float* dx, float* dgamma, float* dbeta,
                               int accumulate, void* ws, hipStream_t stream, void* dx_h, bool in_bf16) {
  // in_bf16 (bf16 storage): dy, y, x and the masked-gradient buffer gmask are bf16 tensors; dx is then written as bf16 only (dx NULL)
  int ct, nc; int64_t rpc;
  col_reduce_plan(rows, C, &ct, &nc, &rpc);
  float* part = reinterpret_cast<float*>(ws);
  float* part1 = part + (size_t)nc * C;
  float* coef = part1 + (size_t)nc * C;              // 3*C floats behind the partials
  const void* g = dy;
  if (y) {
    if (in_bf16)
      hipLaunchKernelGGL((act_bwd_colsum_stage1<true, true>), dim3(ct, nc), dim3(256), 0, stream, dy, y, x, mean, rows, C, rpc, act, alpha,
                         (float*)nullptr, part, part1, reinterpret_cast<unsigned short*>(gmask));
    else
      hipLaunchKernelGGL((act_bwd_colsum_stage1<true, false>), dim3(ct, nc), dim3(256), 0, stream, dy, y, x, mean, rows, C, rpc, act, alpha,
                         reinterpret_cast<float*>(gmask), part, part1, (unsigned short*)nullptr);
    g = gmask;
  } else if (in_bf16) {
    hipLaunchKernelGGL((col_reduce_stage1_v4<true, true, true>), dim3(ct, nc), dim3(256), 0, stream, dy, x, rows, C, rpc, part, part1, 0, mean);
  } else {
    hipLaunchKernelGGL((col_reduce_stage1_v4<true, true, false>), dim3(ct, nc), dim3(256), 0, stream, dy, x, rows, C, rpc, part, part1, 0, mean);
  }
  float* k_dy = coef; float* k_x = coef + C; float* k_0 = coef + 2 * (size_t)C;
  hipLaunchKernelGGL(bn_bwd_stage2_coef_v4, dim3(ct), dim3(256), 0, stream, part, part1, nc, C, mean, rstd, gamma, (float)rows, dgamma, dbeta,
                     k_dy, k_x, k_0, accumulate);
  const size_t n = (size_t)rows * C;
  if (in_bf16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true, true>), dim3(ew_blocks(n >> 2)), dim3(256), 0, stream, g, x, k_dy, k_x, k_0, n, C, dx,
                       reinterpret_cast<uint2*>(dx_h));
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false, false>), dim3(ew_blocks(n >> 2)), dim3(256), 0, stream, g, x, k_dy, k_x, k_0, n, C, dx,
                       reinterpret_cast<uint2*>(dx_h));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Batch norm of a STACKED batch: `groups` passes of the reference graph (the critic on fake / match / mismatch images,
// reference models/gancls/model.py:48-51) run as one batch of groups * B samples; convolutions are per-sample, batch norm is not —
// every pass keeps its own statistics.  The passes are contiguous row ranges, so the same three launches serve all of them:
// stage 1 with chunks that never straddle a group, stage 2 walking the groups (moving averages move once per group, in order),
// normalisation with per-group scale / shift; the backward likewise.  C % 4 == 0, 16-byte aligned tensors.
// ---------------------------------------------------------------------------------------------------------------
// ---- second stage folded into the consumer (round 5) ------------------------------------------------------------------------------------
// With few partial rows per column (<= BN_FUSE_MAX_CHUNKS per group: the small, launch-bound tensors) the second stage of the statistics is not a
// launch of its own: every workgroup of the NORMALISATION owns 64 columns x a run of rows of one group, and its prologue merges the partials of
// exactly those 64 columns (16 chunk lanes x 16 column quads, then 16 -> 4 -> 1 through LDS, Chan's update in a fixed order) into scale / shift in
// LDS.  The workgroup at (group 0, first run of rows) also walks ALL groups in order to write mean / rstd / scale / shift and to move the moving
// averages once per group — the one place where groups meet.  One launch less per batch norm, forward and backward.
constexpr int BN_FUSE_MAX_CHUNKS = 64;

template <bool TILES, bool H>
__device__ __forceinline__ Agg4 bn_merge_group(const float* __restrict__ part0, const float* __restrict__ part1, const void* __restrict__ x,
                                               int g, int nchunks, int64_t rows_g, int64_t rpc, int C, int c, int tx, int ty, bool col_ok,
                                               float (*sn)[16], float4 (*sm)[16], float4 (*sq)[16]) {
  Agg4 a;
  a.n = 0.f; a.mean = make_float4(0.f, 0.f, 0.f, 0.f); a.m2 = a.mean;
  if (col_ok) {
    for (int k = ty; k < nchunks; k += 16) {
      const int64_t rbeg = (int64_t)k * rpc;
      int64_t rend = rbeg + rpc;
      if (rend > rows_g) rend = rows_g;
      Agg4 b;
      b.n = (float)(rend - rbeg);
      const size_t kk = (size_t)g * nchunks + k;
      const float4 p0 = *reinterpret_cast<const float4*>(part0 + kk * C + c);
      const float4 p1 = *reinterpret_cast<const float4*>(part1 + kk * C + c);
      const float inv = 1.f / b.n;
      if (TILES) {
        b.mean = make_float4(p0.x * inv, p0.y * inv, p0.z * inv, p0.w * inv);
        b.m2 = p1;
      } else {
        const float4 s = ld4<H>(x, (size_t)(((int64_t)g * rows_g + rbeg) * C + c) >> 2);
        const float4 d = make_float4(p0.x * inv, p0.y * inv, p0.z * inv, p0.w * inv);
        b.mean = make_float4(s.x + d.x, s.y + d.y, s.z + d.z, s.w + d.w);
        b.m2 = make_float4(fmaxf(p1.x - p0.x * d.x, 0.f), fmaxf(p1.y - p0.y * d.y, 0.f), fmaxf(p1.z - p0.z * d.z, 0.f),
                           fmaxf(p1.w - p0.w * d.w, 0.f));
      }
      a = agg4_merge(a, b);
    }
  }
#pragma unroll
  for (int width = 16; width > 1; width >>= 2) {       // 16 -> 4 -> 1 chunk lanes
    __syncthreads();
    if (ty < width) { sn[ty][tx] = a.n; sm[ty][tx] = a.mean; sq[ty][tx] = a.m2; }
    __syncthreads();
    if (ty < (width >> 2)) {
      a.n = sn[4 * ty][tx]; a.mean = sm[4 * ty][tx]; a.m2 = sq[4 * ty][tx];
#pragma unroll
      for (int j = 1; j < 4; ++j) {
        Agg4 b;
        b.n = sn[4 * ty + j][tx]; b.mean = sm[4 * ty + j][tx]; b.m2 = sq[4 * ty + j][tx];
        a = agg4_merge(a, b);
      }
    }
  }
  return a;       // valid in the lanes ty == 0
}

// grid (ceil(C / 64), groups * nb_g): workgroup (bx, g * nb_g + j) normalises rows [g rows_g + j rpb, ...) of columns [64 bx, 64 bx + 64)
template <bool TILES, bool H>
__global__ __launch_bounds__(256) void bn_fwd_apply_fused(const float* __restrict__ part0, const float* __restrict__ part1,
                                                          const void* __restrict__ x, int nchunks, int64_t rows_g, int64_t rpc, int C, BnFin fin,
                                                          int groups, int64_t rpb, int nb_g, int act, float alpha, float* __restrict__ y,
                                                          uint2* __restrict__ yh) {
  __shared__ float sn[16][16];
  __shared__ float4 sm[16][16], sq[16][16];
  __shared__ float4 s_scale[16], s_shift[16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + tx * 4;
  const bool col_ok = c < C;
  const int g_own = blockIdx.y / nb_g, j = blockIdx.y - g_own * nb_g;
  const bool scribe = blockIdx.y == 0;                 // writes the layer's outputs and moves the moving averages, group by group
  for (int g = scribe ? 0 : g_own; g < (scribe ? groups : g_own + 1); ++g) {
    const Agg4 r = bn_merge_group<TILES, H>(part0, part1, x, g, nchunks, rows_g, rpc, C, c, tx, ty, col_ok, sn, sm, sq);
    if (ty == 0 && col_ok) {
      const float mu[4] = {r.mean.x, r.mean.y, r.mean.z, r.mean.w};
      const float mo[4] = {r.m2.x, r.m2.y, r.m2.z, r.m2.w};
      float sc[4], sh[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float var = fmaxf(mo[e], 0.f) / r.n;
        const float rs = rsqrtf(var + fin.eps);
        sc[e] = fin.gamma[c + e] * rs;
        sh[e] = fin.beta[c + e] - mu[e] * sc[e];
        if (scribe) bn_fin_col(fin, c + e, r.n, mu[e], mo[e], (size_t)g * C);     // (same expressions: the stored scale / shift are these bits)
      }
      if (g == g_own) { s_scale[tx] = make_float4(sc[0], sc[1], sc[2], sc[3]); s_shift[tx] = make_float4(sh[0], sh[1], sh[2], sh[3]); }
    }
  }
  __syncthreads();
  if (!col_ok) return;
  const float4 sc = s_scale[tx], sh = s_shift[tx];
  const int64_t rbeg = (int64_t)g_own * rows_g + (int64_t)j * rpb;
  int64_t rend = rbeg + rpb;
  if (rend > (int64_t)(g_own + 1) * rows_g) rend = (int64_t)(g_own + 1) * rows_g;
#pragma unroll 4
  for (int64_t r = rbeg + ty; r < rend; r += 16) {
    const size_t e4 = (size_t)(r * C + c) >> 2;
    float4 v = ld4<H>(x, e4);
    v.x = apply_act(v.x * sc.x + sh.x, act, alpha);
    v.y = apply_act(v.y * sc.y + sh.y, act, alpha);
    v.z = apply_act(v.z * sc.z + sh.z, act, alpha);
    v.w = apply_act(v.w * sc.w + sh.w, act, alpha);
    st4(y, yh, e4, v);
  }
}

// backward: prologue = sums of the group's partials (sum g, sum g (x - mean)) -> the three coefficients of dx for the workgroup's 64 columns;
// the scribe workgroup also forms dgamma / dbeta over all groups.  dx = k_dy g + k_x x + k_0.
template <bool HD, bool HX>
__global__ __launch_bounds__(256) void bn_bwd_apply_fused(const float* __restrict__ part0, const float* __restrict__ part1, int nchunks,
                                                          int64_t rows_g, int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          int accumulate, int groups, int64_t rpb, int nb_g, const void* __restrict__ gv,
                                                          const void* __restrict__ xv, float* __restrict__ dx, uint2* __restrict__ dxh) {
  __shared__ float4 red[16][16], red1[16][16];
  __shared__ float4 s_a[16], s_b[16], s_e[16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + tx * 4;
  const bool col_ok = c < C;
  const int g_own = blockIdx.y / nb_g, j = blockIdx.y - g_own * nb_g;
  const bool scribe = blockIdx.y == 0;
  const float n = (float)rows_g;
  float tg[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
  for (int g = scribe ? 0 : g_own; g < (scribe ? groups : g_own + 1); ++g) {
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col_ok) {
      for (int k = ty; k < nchunks; k += 16) {
        const size_t kk = (size_t)g * nchunks + k;
        const float4 v = *reinterpret_cast<const float4*>(part0 + kk * C + c);
        const float4 w = *reinterpret_cast<const float4*>(part1 + kk * C + c);
        a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
        a1.x += w.x; a1.y += w.y; a1.z += w.z; a1.w += w.w;
      }
    }
    __syncthreads();
    red[ty][tx] = a0;
    red1[ty][tx] = a1;
    __syncthreads();
    if (ty == 0 && col_ok) {
      float4 s0 = red[0][tx], s1 = red1[0][tx];
#pragma unroll
      for (int k = 1; k < 16; ++k) {
        const float4 v = red[k][tx], w = red1[k][tx];
        s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
        s1.x += w.x; s1.y += w.y; s1.z += w.z; s1.w += w.w;
      }
      const float sd[4] = {s0.x, s0.y, s0.z, s0.w}, sx[4] = {s1.x, s1.y, s1.z, s1.w};
      float ka[4], kb[4], ke[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int cc = c + e;
        const float mu = mean[(size_t)g * C + cc], rs = rstd[(size_t)g * C + cc], gm = gamma[cc], sdy = sd[e];
        const float sdyxh = rs * sx[e];
        tg[e] = g ? tg[e] + sdyxh : sdyxh;
        tb[e] = g ? tb[e] + sdy : sdy;
        const float grs = gm * rs;
        ka[e] = grs;
        kb[e] = -grs * rs * sdyxh / n;
        ke[e] = -grs * sdy / n + grs * rs * mu * sdyxh / n;
      }
      if (g == g_own) {
        s_a[tx] = make_float4(ka[0], ka[1], ka[2], ka[3]); s_b[tx] = make_float4(kb[0], kb[1], kb[2], kb[3]);
        s_e[tx] = make_float4(ke[0], ke[1], ke[2], ke[3]);
      }
    }
  }
  if (scribe && ty == 0 && col_ok) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dgamma[c + e] = accumulate ? dgamma[c + e] + tg[e] : tg[e];
      dbeta[c + e] = accumulate ? dbeta[c + e] + tb[e] : tb[e];
    }
  }
  __syncthreads();
  if (!col_ok) return;
  const float4 a = s_a[tx], b = s_b[tx], e = s_e[tx];
  const int64_t rbeg = (int64_t)g_own * rows_g + (int64_t)j * rpb;
  int64_t rend = rbeg + rpb;
  if (rend > (int64_t)(g_own + 1) * rows_g) rend = (int64_t)(g_own + 1) * rows_g;
#pragma unroll 4
  for (int64_t r = rbeg + ty; r < rend; r += 16) {
    const size_t e4 = (size_t)(r * C + c) >> 2;
    const float4 d = ld4<HD>(gv, e4);
    const float4 v = ld4<HX>(xv, e4);
    float4 o;
    o.x = a.x * d.x + b.x * v.x + e.x;
    o.y = a.y * d.y + b.y * v.y + e.y;
    o.z = a.z * d.z + b.z * v.z + e.z;
    o.w = a.w * d.w + b.w * v.w + e.w;
    st4(dx, dxh, e4, o);
  }
}

// rows per workgroup of the fused normalisation: about 1024 workgroups in flight, runs of at least 16 rows
static void bn_fuse_rows(int64_t rows_g, int ct, int groups, int64_t* rpb, int* nb_g) {
  int64_t want = 1024 / ((int64_t)ct * groups);
  if (want < 1) want = 1;
  const int64_t maxb = (rows_g + 15) / 16;
  if (want > maxb) want = maxb;
  *rpb = (rows_g + want - 1) / want;
  *nb_g = (int)((rows_g + *rpb - 1) / *rpb);
}

static void bn_group_plan(int64_t rows_g, int C, int groups, int* ct, int* ncg, int64_t* rpc) {
  col_reduce_plan(rows_g, C, ct, ncg, rpc);
  const int cap = tuning().colred_cap / groups > 0 ? tuning().colred_cap / groups : 1;     // the same total number of partial rows as one pass
  if (*ncg > cap) {
    *rpc = (rows_g + cap - 1) / cap;
    *ncg = (int)((rows_g + *rpc - 1) / *rpc);
  }
}

size_t bn_grouped_ws(int64_t rows_g, int C, int groups) {
  int ct, ncg; int64_t rpc;
  bn_group_plan(rows_g, C, groups, &ct, &ncg, &rpc);
  return ((size_t)groups * ncg * C * 2 + (size_t)groups * 3 * C) * sizeof(float);
}

hipError_t bn_fwd_grouped_launch(const void* x, int64_t rows_g, int C, int groups, const float* gamma, const float* beta, float eps, float decay,
                                 float* mean, float* rstd, float* scale, float* shift, float* mm, float* mv, int act, float alpha, float* y,
                                 void* y_h, void* ws, hipStream_t stream, bool x_bf16, const float* tile_sum, const float* tile_m2, int tile_chunks,
                                 int tile_rows, int moving_updates, int moving_groups) {
  // tile_sum != NULL: the producing conv's epilogue left per-tile partials (tile_chunks tiles of tile_rows rows per group): no first stage
  // moving_groups > 0: only the first moving_groups groups move the moving averages (a stacked pass whose later groups run outside UPDATE_OPS)
  int ct, ncg; int64_t rpc;
  bn_group_plan(rows_g, C, groups, &ct, &ncg, &rpc);
  const float* part0 = tile_sum;
  const float* part1 = tile_m2;
  const bool tiles = tile_sum != nullptr;
  if (tiles) { ncg = tile_chunks; rpc = tile_rows; }
  const int64_t rows = rows_g * groups;
  if (!tiles) {
    float* p0 = reinterpret_cast<float*>(ws);
    float* p1 = p0 + (size_t)groups * ncg * C;
    part0 = p0; part1 = p1;
    if (x_bf16)
      hipLaunchKernelGGL((col_reduce_stage1_v4<true, false, true>), dim3(ct, groups * ncg), dim3(256), 0, stream, x, (const void*)nullptr, rows, C, rpc,
                         p0, p1, 1, (const float*)nullptr, rows_g, ncg);
    else
      hipLaunchKernelGGL((col_reduce_stage1_v4<true, false, false>), dim3(ct, groups * ncg), dim3(256), 0, stream, x, (const void*)nullptr, rows, C, rpc,
                         p0, p1, 1, (const float*)nullptr, rows_g, ncg);
  }
  BnFin fin = make_fin(gamma, beta, eps, decay, mean, rstd, scale, shift, mm, mv, moving_updates);
  if (moving_groups > 0 && moving_groups < groups) fin.upd_limit = (size_t)moving_groups * C;
  uint2* yh = reinterpret_cast<uint2*>(y_h);
  if (tuning().bn_fuse && ncg <= BN_FUSE_MAX_CHUNKS) {          // second stage in the normalisation's prologue: one launch less
    int64_t rpb; int nb_g;
    bn_fuse_rows(rows_g, ct, groups, &rpb, &nb_g);
    const dim3 grid(ct, groups * nb_g);
#define T2I_BNF(TL, HH) hipLaunchKernelGGL((bn_fwd_apply_fused<TL, HH>), grid, dim3(256), 0, stream, part0, part1, x, ncg, rows_g, rpc, C, fin, groups, rpb, \
                                           nb_g, act, alpha, y, yh)
    if (tiles) { if (x_bf16) T2I_BNF(true, true); else T2I_BNF(true, false); }
    else { if (x_bf16) T2I_BNF(false, true); else T2I_BNF(false, false); }
#undef T2I_BNF
    return hipGetLastError();
  }
#define T2I_BNS(TL, HH) hipLaunchKernelGGL((bn_stats_stage2_v4<TL, HH>), dim3((C + 15) / 16), dim3(256), 0, stream, part0, part1, x, ncg, rows_g, rpc, C, \
                                           (float*)nullptr, (float*)nullptr, fin, groups)
  if (tiles) { if (x_bf16) T2I_BNS(true, true); else T2I_BNS(true, false); }
  else { if (x_bf16) T2I_BNS(false, true); else T2I_BNS(false, false); }
#undef T2I_BNS
  const size_t n = (size_t)rows * C, pg4 = ((size_t)rows_g * C) >> 2;
  if (x_bf16)
    hipLaunchKernelGGL((bn_apply_kernel<true, true>), dim3(ew_blocks(n >> 2)), dim3(256), 0, stream, x, scale, shift, n, C, act, alpha, y, yh, pg4);
  else
    hipLaunchKernelGGL((bn_apply_kernel<true, false>), dim3(ew_blocks(n >> 2)), dim3(256), 0, stream, x, scale, shift, n, C, act, alpha, y, yh, pg4);
  return hipGetLastError();
}

hipError_t bn_bwd_grouped_launch(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma,
                                 int64_t rows_g, int C, int groups, int act, float alpha, void* gmask, float* dx, float* dgamma, float* dbeta,
                                 int accumulate, void* ws, hipStream_t stream, void* dx_h, bool in_bf16) {
  int ct, ncg; int64_t rpc;
  bn_group_plan(rows_g, C, groups, &ct, &ncg, &rpc);
  const int64_t rows = rows_g * groups;
  float* part = reinterpret_cast<float*>(ws);
  float* part1 = part + (size_t)groups * ncg * C;
  float* coef = part1 + (size_t)groups * ncg * C;            // [groups][3][C] behind the partials
  const void* g = dy;
  const dim3 grid(ct, groups * ncg);
  if (y) {
    if (in_bf16)
      hipLaunchKernelGGL((act_bwd_colsum_stage1<true, true>), grid, dim3(256), 0, stream, dy, y, x, mean, rows, C, rpc, act, alpha,
                         (float*)nullptr, part, part1, reinterpret_cast<unsigned short*>(gmask), rows_g, ncg);
    else
      hipLaunchKernelGGL((act_bwd_colsum_stage1<true, false>), grid, dim3(256), 0, stream, dy, y, x, mean, rows, C, rpc, act, alpha,
                         reinterpret_cast<float*>(gmask), part, part1, (unsigned short*)nullptr, rows_g, ncg);
    g = gmask;
  } else if (in_bf16) {
    hipLaunchKernelGGL((col_reduce_stage1_v4<true, true, true>), grid, dim3(256), 0, stream, dy, x, rows, C, rpc, part, part1, 0, mean, rows_g, ncg);
  } else {
    hipLaunchKernelGGL((col_reduce_stage1_v4<true, true, false>), grid, dim3(256), 0, stream, dy, x, rows, C, rpc, part, part1, 0, mean, rows_g, ncg);
  }
  if (tuning().bn_fuse && ncg <= BN_FUSE_MAX_CHUNKS) {          // second stage + coefficients in the prologue of the dx kernel
    int64_t rpb; int nb_g;
    bn_fuse_rows(rows_g, ct, groups, &rpb, &nb_g);
    const dim3 grid2(ct, groups * nb_g);
    if (in_bf16)
      hipLaunchKernelGGL((bn_bwd_apply_fused<true, true>), grid2, dim3(256), 0, stream, part, part1, ncg, rows_g, C, mean, rstd, gamma, dgamma, dbeta,
                         accumulate, groups, rpb, nb_g, g, x, dx, reinterpret_cast<uint2*>(dx_h));
    else
      hipLaunchKernelGGL((bn_bwd_apply_fused<false, false>), grid2, dim3(256), 0, stream, part, part1, ncg, rows_g, C, mean, rstd, gamma, dgamma, dbeta,
                         accumulate, groups, rpb, nb_g, g, x, dx, reinterpret_cast<uint2*>(dx_h));
    return hipGetLastError();
  }
  float* k_dy = coef; float* k_x = coef + C; float* k_0 = coef + 2 * (size_t)C;
  hipLaunchKernelGGL(bn_bwd_stage2_coef_v4, dim3(ct), dim3(256), 0, stream, part, part1, ncg, C, mean, rstd, gamma, (float)rows_g, dgamma, dbeta,
                     k_dy, k_x, k_0, accumulate, groups);
  const size_t n = (size_t)rows * C, pg4 = ((size_t)rows_g * C) >> 2;
  if (in_bf16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true, true>), dim3(ew_blocks(n >> 2)), dim3(256), 0, stream, g, x, k_dy, k_x, k_0, n, C, dx,
                       reinterpret_cast<uint2*>(dx_h), pg4);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false, false>), dim3(ew_blocks(n >> 2)), dim3(256), 0, stream, g, x, k_dy, k_x, k_0, n, C, dx,
                       reinterpret_cast<uint2*>(dx_h), pg4);
  return hipGetLastError();
}

hipError_t ew_launch(int op, const void* a, const void* b, size_t n_flag, int act, float alpha, float beta, float* y,
                     hipStream_t stream, void* y_h, bool in_bf16) {
  uint2* yh = reinterpret_cast<uint2*>(y_h);          // bf16 copy of y (only with the float4 body: n % 4 == 0, aligned)
  // bit 63 of n_flag set => some pointer is not 16-byte aligned: no float4 body, everything through the scalar tail
  const bool al = (n_flag >> 63) == 0;
  const size_t n = n_flag & ~(1ull << 63);
  const size_t n4 = al ? (n >> 2) : 0;
  dim3 g(ew_blocks(al ? ((n + 3) >> 2) : n)), blk(256);
#define T2I_EW(OPC, HASB)                                                                                                        \
  do {                                                                                                                           \
    if (in_bf16) hipLaunchKernelGGL((ew_kernel<OPC, HASB, true>), g, blk, 0, stream, a, b, n, n4, act, alpha, beta, y, yh);      \
    else hipLaunchKernelGGL((ew_kernel<OPC, HASB, false>), g, blk, 0, stream, a, b, n, n4, act, alpha, beta, y, yh);             \
  } while (0)
  switch (op) {
    case EW_ACT_FWD: T2I_EW(EW_ACT_FWD, false); break;
    case EW_ACT_BWD: T2I_EW(EW_ACT_BWD, true); break;
    case EW_ADD_ACT: T2I_EW(EW_ADD_ACT, true); break;
    default:
      if (b) T2I_EW(EW_AXPBY, true);
      else T2I_EW(EW_AXPBY, false);
  }
#undef T2I_EW
  return hipGetLastError();
}

// x_hat[b,:] = eps[b]*g[b,:] + (1-eps[b])*x[b,:]
__global__ __launch_bounds__(256) void interp_kernel(const float* __restrict__ eps, const float* __restrict__ g,
                                                     const float* __restrict__ x, size_t n, size_t per,
                                                     float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float e = eps[i / per];
    out[i] = e * g[i] + (1.f - e) * x[i];
  }
}

// out[b,p,:] = [feat[b,p,:Cf] | emb[b,:Ce]].  T = float or unsigned short (bf16 storage: a pure copy, 2-byte elements)
template <typename T>
__global__ __launch_bounds__(256) void concat_tile_fwd_kernel(const T* __restrict__ feat, const T* __restrict__ emb,
                                                              size_t n, int P, int Cf, int Ce, T* __restrict__ out) {
  const int Ct = Cf + Ce;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / Ct;
    const int c = (int)(i - pix * Ct);
    out[i] = c < Cf ? feat[pix * Cf + c] : emb[(pix / P) * Ce + (c - Cf)];
  }
}

// dfeat = dout[..., :Cf]; demb[b,:] = sum_p dout[b,p,Cf:]  (H: bf16 tensors; the sum over the P positions is taken in fp32)
__device__ __forceinline__ void st1(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void st1(unsigned short* p, size_t i, float v) { p[i] = (unsigned short)(aux_pk2(v, 0.f) & 0xFFFFu); }
template <bool H>
__global__ __launch_bounds__(256) void concat_tile_bwd_kernel(const void* __restrict__ dout, size_t nfeat, size_t nemb,
                                                              int P, int Cf, int Ce, void* __restrict__ dfeat,
                                                              void* __restrict__ demb) {
  typedef typename std::conditional<H, unsigned short, float>::type T;
  const int Ct = Cf + Ce;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nfeat + nemb; i += (size_t)gridDim.x * blockDim.x) {
    if (i < nfeat) {
      const size_t pix = i / Cf;
      const int c = (int)(i - pix * Cf);
      reinterpret_cast<T*>(dfeat)[i] = reinterpret_cast<const T*>(dout)[pix * Ct + c];
    } else {
      const size_t j = i - nfeat;
      const size_t b = j / Ce;
      const int c = (int)(j - b * Ce);
      float s = 0.f;
      for (int p = 0; p < P; ++p) s += ld1<H>(dout, (b * P + p) * Ct + Cf + c);
      st1(reinterpret_cast<T*>(demb), j, s);
    }
  }
}

// [B,C,HW] <-> [B,HW,C] through a 32x32 LDS tile (+1 pad) so both sides are coalesced
template <typename T>            // float, or unsigned short for bf16 tensors (a pure copy)
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ x, int R, int Cc, T* __restrict__ y) {
  // per batch: x is [R, Cc] row-major -> y is [Cc, R]
  __shared__ T tile[32][33];
  const size_t boff = (size_t)blockIdx.z * R * Cc;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < Cc) ? x[boff + (size_t)r * Cc + c] : (T)0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, r = r0 + tx;
    if (r < R && c < Cc) y[boff + (size_t)c * R + r] = tile[tx][j];
  }
}

hipError_t interp_launch(const float* eps, const float* g, const float* x, int B, int64_t per, float* out,
                         hipStream_t stream) {
  const size_t n = (size_t)B * per;
  hipLaunchKernelGGL(interp_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, eps, g, x, n, (size_t)per, out);
  return hipGetLastError();
}

hipError_t concat_tile_fwd_launch(const void* feat, const void* emb, int B, int P, int Cf, int Ce, void* out,
                                  hipStream_t stream, bool bf16) {
  const size_t n = (size_t)B * P * (Cf + Ce);
  if (bf16)
    hipLaunchKernelGGL(concat_tile_fwd_kernel<unsigned short>, dim3(ew_blocks(n)), dim3(256), 0, stream, reinterpret_cast<const unsigned short*>(feat),
                       reinterpret_cast<const unsigned short*>(emb), n, P, Cf, Ce, reinterpret_cast<unsigned short*>(out));
  else
    hipLaunchKernelGGL(concat_tile_fwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, stream, reinterpret_cast<const float*>(feat),
                       reinterpret_cast<const float*>(emb), n, P, Cf, Ce, reinterpret_cast<float*>(out));
  return hipGetLastError();
}

hipError_t concat_tile_bwd_launch(const void* dout, int B, int P, int Cf, int Ce, void* dfeat, void* demb,
                                  hipStream_t stream, bool bf16) {
  const size_t nfeat = (size_t)B * P * Cf, nemb = (size_t)B * Ce;
  if (bf16)
    hipLaunchKernelGGL(concat_tile_bwd_kernel<true>, dim3(ew_blocks(nfeat + nemb)), dim3(256), 0, stream, dout, nfeat, nemb, P, Cf, Ce, dfeat, demb);
  else
    hipLaunchKernelGGL(concat_tile_bwd_kernel<false>, dim3(ew_blocks(nfeat + nemb)), dim3(256), 0, stream, dout, nfeat, nemb, P, Cf, Ce, dfeat, demb);
  return hipGetLastError();
}

hipError_t transpose_launch(const void* x, int B, int R, int Cc, void* y, hipStream_t stream, bool bf16) {
  dim3 grid((Cc + 31) / 32, (R + 31) / 32, B);
  if (bf16)
    hipLaunchKernelGGL(transpose_kernel<unsigned short>, grid, dim3(256), 0, stream, reinterpret_cast<const unsigned short*>(x), R, Cc,
                       reinterpret_cast<unsigned short*>(y));
  else
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, stream, reinterpret_cast<const float*>(x), R, Cc, reinterpret_cast<float*>(y));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// gradient penalty: per-sample L2 norm (one block per sample, wave64 shuffle + LDS across the 4 waves)
// ---------------------------------------------------------------------------------------------------------------
template <bool H>
__global__ __launch_bounds__(256) void gp_slopes_kernel(const void* __restrict__ gv, size_t per, float* __restrict__ slopes) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * per;
  float acc = 0.f;
  const size_t n4 = ((per & 3) == 0 && (base & 3) == 0) ? (per >> 2) : 0;
  for (size_t i = threadIdx.x; i < n4; i += 256) {
    const float4 v = ld4<H>(gv, (base >> 2) + i);
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (size_t i = (n4 << 2) + threadIdx.x; i < per; i += 256) { const float v = ld1<H>(gv, base + i); acc += v * v; }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) slopes[blockIdx.x] = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
}

template <bool H>
__global__ __launch_bounds__(256) void row_scale_kernel(const void* __restrict__ g, const float* __restrict__ coef,
                                                        size_t n, size_t per, void* __restrict__ out, const float* __restrict__ den) {
  typedef typename std::conditional<H, unsigned short, float>::type T;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / per;
    // den != NULL: the backward of the slope norm, coef_b = d / ||g_b|| with the guard of the tensor-library expression it replaces
    // (where(s > 0, d / clamp_min(s, 1e-30), 0): five launches per gradient-penalty term before)
    const float c = den ? (den[b] > 0.f ? coef[b] / fmaxf(den[b], 1e-30f) : 0.f) : coef[b];
    st1(reinterpret_cast<T*>(out), i, c * ld1<H>(g, i));
  }
}

hipError_t gp_slopes_launch(const void* g, int B, int64_t per, float* slopes, hipStream_t stream, bool bf16) {
  if (bf16) hipLaunchKernelGGL(gp_slopes_kernel<true>, dim3(B), dim3(256), 0, stream, g, (size_t)per, slopes);
  else hipLaunchKernelGGL(gp_slopes_kernel<false>, dim3(B), dim3(256), 0, stream, g, (size_t)per, slopes);
  return hipGetLastError();
}

hipError_t row_scale_launch(const void* g, const float* coef, int B, int64_t per, void* out, hipStream_t stream, bool bf16, const float* den) {
  const size_t n = (size_t)B * per;
  if (bf16) hipLaunchKernelGGL(row_scale_kernel<true>, dim3(ew_blocks(n)), dim3(256), 0, stream, g, coef, n, (size_t)per, out, den);
  else hipLaunchKernelGGL(row_scale_kernel<false>, dim3(ew_blocks(n)), dim3(256), 0, stream, g, coef, n, (size_t)per, out, den);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// tf.train.AdamOptimizer over a flat arena
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v, size_t n, float lr_val,
                                                      const float* __restrict__ lr_dev, float b1, float b2, float eps,
                                                      float gscale) {
  const float lr_t = lr_dev ? *lr_dev : lr_val;   // device scalar: a captured graph replays with a new step size
  const size_t n4 = n >> 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = t; i < n4; i += stride) {
    float4 W = reinterpret_cast<float4*>(w)[i];
    const float4 G0 = reinterpret_cast<const float4*>(g)[i];
    // beta1 == 0 (the wgancls optimizers: Adam(0, 0.9)): m_t = g_t whatever m_{t-1} was — the old moment is not read
    // (6 instead of 7 streams over the arena); m == NULL (round 4, only with beta1 == 0): it is not written either — m_t is
    // grad * grad_scale, which the caller can form whenever a checkpoint wants it (5 streams)
    float4 M = b1 != 0.f ? reinterpret_cast<float4*>(m)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 V = reinterpret_cast<float4*>(v)[i];
    float gx = G0.x * gscale, gy = G0.y * gscale, gz = G0.z * gscale, gw = G0.w * gscale;
    M.x = b1 * M.x + (1.f - b1) * gx; M.y = b1 * M.y + (1.f - b1) * gy;
    M.z = b1 * M.z + (1.f - b1) * gz; M.w = b1 * M.w + (1.f - b1) * gw;
    V.x = b2 * V.x + (1.f - b2) * gx * gx; V.y = b2 * V.y + (1.f - b2) * gy * gy;
    V.z = b2 * V.z + (1.f - b2) * gz * gz; V.w = b2 * V.w + (1.f - b2) * gw * gw;
    W.x -= lr_t * M.x / (sqrtf(V.x) + eps); W.y -= lr_t * M.y / (sqrtf(V.y) + eps);
    W.z -= lr_t * M.z / (sqrtf(V.z) + eps); W.w -= lr_t * M.w / (sqrtf(V.w) + eps);
    reinterpret_cast<float4*>(w)[i] = W;
    if (m) reinterpret_cast<float4*>(m)[i] = M;
    reinterpret_cast<float4*>(v)[i] = V;
  }
  for (size_t i = (n4 << 2) + t; i < n; i += stride) {
    const float gg = g[i] * gscale;
    const float mm = b1 * (b1 != 0.f ? m[i] : 0.f) + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    if (m) m[i] = mm;
    v[i] = vv;
    w[i] -= lr_t * mm / (sqrtf(vv) + eps);
  }
}

hipError_t adam_tf_launch(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, const float* lr_dev,
                          float b1, float b2, float eps, float gscale, hipStream_t stream) {
  const int adam_cap = tuning().adam_blocks;
  size_t nb = ((((size_t)n + 3) >> 2) + 255) / 256;
  if (nb > (size_t)adam_cap) nb = adam_cap;
  hipLaunchKernelGGL(adam_tf_kernel, dim3((int)nb), dim3(256), 0, stream, w, g, m, v, (size_t)n,
                     lr_t, lr_dev, b1, b2, eps, gscale);
  return hipGetLastError();
}

// kt <- kt - lr * d(balance_loss)/d(kt) with balance_loss = (kt*wdist2 - wdist)^2 (reference models/wgancls/model.py:85,100:
// GradientDescentOptimizer(0.001) on kt).  wdist / wdist2 arrive as SUMS over the data-parallel ranks of the per-rank batch
// means (scale = 1/ranks turns them into the global-batch means: the balance loss is quadratic in them, so averaging
// per-rank gradients would not be the gradient of the global loss).  One thread; no fma contraction, so that one rank
// (sum == value, scale == 1) and N identical ranks agree bit for bit.
__global__ void kt_sgd_kernel(float* __restrict__ kt, const float* __restrict__ sums, float scale, float lr) {
#pragma clang fp contract(off)
  const float wd = sums[0] * scale, wd2 = sums[1] * scale;
  const float k = *kt;
  const float grad = 2.f * (k * wd2 - wd) * wd2;
  *kt = k - lr * grad;
}

// ---------------------------------------------------------------------------------------------------------------
// Truncated normal (tf.truncated_normal, reference models/wgancls/model.py:119: the conditioning-augmentation noise the graph redraws on
// every run): out = mean + std * t with t ~ N(0,1) restricted to [lo, hi] — by inverting the normal CDF on a uniform draw in
// (Phi(lo), Phi(hi)), the same construction torch.nn.init.trunc_normal_ uses (8 tensor-library launches per draw: uniform, two scalings,
// erfinv, two more scalings, clamp).  Uniforms come from Philox4x32-10 keyed by (seed, offset + element / 4): counter-based, so the draw
// is a pure function of (seed, offset, index) — reproducible whatever the launch geometry.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void trunc_normal_kernel(float* __restrict__ out, size_t n, unsigned long long seed, unsigned long long offset,
                                                           float mean, float std, float lo, float hi, float cdf_lo, float cdf_span) {
  const size_t quads = (n + 3) >> 2;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long ctr = offset + q;
    unsigned r[4];
    philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t i = q * 4 + e;
      if (i >= n) break;
      const float u = ((float)(r[e] >> 8) + 0.5f) * (1.0f / 16777216.0f);       // (0, 1), 24 bits
      const float p = cdf_lo + u * cdf_span;                                       // in (Phi(lo), Phi(hi))
      float t = 1.41421356237f * erfinvf(2.f * p - 1.f);
      t = fminf(fmaxf(t, lo), hi);
      out[i] = mean + std * t;
    }
  }
}

hipError_t trunc_normal_launch(float* out, size_t n, unsigned long long seed, unsigned long long offset, float mean, float std, float lo, float hi,
                               hipStream_t stream) {
  const double cl = 0.5 * (1.0 + erf((double)lo / 1.4142135623730951)), ch = 0.5 * (1.0 + erf((double)hi / 1.4142135623730951));
  const size_t quads = (n + 3) >> 2;
  size_t blocks = (quads + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(trunc_normal_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, stream, out, n, seed, offset, mean, std, lo, hi, (float)cl,
                     (float)(ch - cl));
  return hipGetLastError();
}

// Zero a table of element ranges of one buffer in one launch (the small slots of a gradient arena — biases, batch-norm gamma / beta — between
// the large filter slots, whose first contribution of a step is a plain store: optim.Arena.zero_grad).  ranges: device int64 [n][2] = (start, length).
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* __restrict__ base, const long long* __restrict__ ranges, int n) {
  for (int r = blockIdx.x; r < n; r += gridDim.x) {
    const long long s = ranges[2 * r], len = ranges[2 * r + 1];
    for (long long i = threadIdx.x; i < len; i += 256) base[s + i] = 0.f;
  }
}

hipError_t zero_ranges_launch(float* base, const long long* ranges, int n, hipStream_t stream) {
  hipLaunchKernelGGL(zero_ranges_kernel, dim3(n < 1024 ? n : 1024), dim3(256), 0, stream, base, ranges, n);
  return hipGetLastError();
}

hipError_t kt_sgd_launch(float* kt, const float* sums, float scale, float lr, hipStream_t stream) {
  hipLaunchKernelGGL(kt_sgd_kernel, dim3(1), dim3(1), 0, stream, kt, sums, scale, lr);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Data pipeline (reference preprocess/dataset.py:83-96,98-120,150): the per-batch image work — gather by id from the
// resident uint8 store, scale to [-1,1], random crop, horizontal flip — and the mean of `k` chosen caption embeddings.
// Pure byte / gather work: HBM-bound, one pass, coalesced over the channel-innermost output.
// Bit-exact with the NumPy reference: v = fl32(fl32(u8 * fl32(2/255)) - 1) with NO fused multiply-add, and the mean is
// the sequential fp32 sum in choice order followed by an IEEE division by k (np.mean over axis 0 of a [k,D] array).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_flip_normalize_kernel(const uint8_t* __restrict__ src, int S,
                                                                  const int32_t* __restrict__ ids,
                                                                  const int32_t* __restrict__ row0,
                                                                  const int32_t* __restrict__ col0,
                                                                  const int32_t* __restrict__ flip, int out_size,
                                                                  float* __restrict__ out) {
  const int b = blockIdx.y;
  const size_t img = (size_t)ids[b] * S * S * 3;
  const int r0 = row0[b], c0 = col0[b], fl = flip[b];
  const int per = out_size * out_size * 3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
    const int ch = i % 3, px = i / 3;
    const int c = px % out_size, r = px / out_size;
    const int sc = fl ? (c0 + out_size - 1 - c) : (c0 + c);
    const float u = (float)src[img + ((size_t)(r0 + r) * S + sc) * 3 + ch];
    float v;
    {
#pragma clang fp contract(off)      // hipcc contracts a*b-c into one fma by default; NumPy rounds the product first
      const float prod = u * 0.00784313725490196f;
      v = prod - 1.0f;
    }
    out[(size_t)b * per + i] = v;
  }
}

hipError_t crop_flip_normalize_launch(const uint8_t* src, int S, const int32_t* ids, const int32_t* row0, const int32_t* col0,
                                      const int32_t* flip, int B, int out_size, float* out, hipStream_t stream) {
  const int per = out_size * out_size * 3;
  int bx = (per + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(crop_flip_normalize_kernel, dim3(bx, B), dim3(256), 0, stream, src, S, ids, row0, col0, flip, out_size, out);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void gather_mean_kernel(const float* __restrict__ emb, int En, int D,
                                                          const int32_t* __restrict__ ids,
                                                          const int32_t* __restrict__ choice, int k,
                                                          float* __restrict__ out) {
  const int b = blockIdx.y;
  const float* base = emb + (size_t)ids[b] * En * D;
  for (int d = blockIdx.x * blockDim.x + threadIdx.x; d < D; d += gridDim.x * blockDim.x) {
    float s = base[(size_t)choice[b * k] * D + d];
    for (int j = 1; j < k; ++j) s = __fadd_rn(s, base[(size_t)choice[b * k + j] * D + d]);
    out[(size_t)b * D + d] = k > 1 ? __fdiv_rn(s, (float)k) : s;
  }
}

hipError_t gather_mean_launch(const float* emb, int En, int D, const int32_t* ids, const int32_t* choice, int B, int k,
                              float* out, hipStream_t stream) {
  hipLaunchKernelGGL(gather_mean_kernel, dim3((D + 255) / 256, B), dim3(256), 0, stream, emb, En, D, ids, choice, k, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// PGGAN operators (reference utils/ops.py:74-81 layer_norm, :100-101 pool, :109-111 upscale)
//   resample2<POOL>   2x2 stride-2 window sum (POOL) or nearest x2 replication, times a scalar.  avg pool = sum * 1/4;
//                     the two are adjoint up to that factor, so each is the other's backward (and double backward).
//   row_moments       per-sample sums over everything but the batch axis: s1 = sum a, s2 = sum a*b (b = a if NULL)
//   row_fma2          out[b,i] = a[b,i]*alpha[b] + b_[b,i]*gamma[b] + delta[b]  (per-sample scalars)
// Layer norm = row_moments -> row_fma2 (normalise) -> the per-channel affine + activation of bn_apply; its backward =
// col_reduce (dgamma, dbeta) + row_moments + row_fma2.  All HBM-bound single passes.
// ---------------------------------------------------------------------------------------------------------------
template <bool POOL, typename T>     // T = float4 (C % 4 == 0, aligned) or float; C is then counted in units of T
__global__ __launch_bounds__(256) void resample2_kernel(const T* __restrict__ x, int Ho, int Wo, int C, float scale,
                                                        size_t n_out, T* __restrict__ y) {
  // POOL: x [B,2Ho,2Wo,C] -> y [B,Ho,Wo,C];  else: x [B,Ho/2,Wo/2,C] -> y [B,Ho,Wo,C]
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t p = i / C;
    const int w = (int)(p % Wo); p /= Wo;
    const int h = (int)(p % Ho);
    const size_t b = p / Ho;
    if (POOL) {
      const size_t Wi = (size_t)2 * Wo;
      const T* s = x + ((b * 2 * Ho + 2 * h) * Wi + 2 * w) * C + c;
      y[i] = vscale(vadd(vadd(s[0], s[C]), vadd(s[Wi * C], s[Wi * C + C])), scale);
    } else {
      y[i] = vscale(x[((b * (Ho >> 1) + (h >> 1)) * (Wo >> 1) + (w >> 1)) * C + c], scale);
    }
  }
}

hipError_t resample2_launch(bool pool, const float* x, int B, int Ho, int Wo, int C, float scale, float* y, hipStream_t stream) {
  const bool v4 = (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  if (v4) {
    const size_t n = (size_t)B * Ho * Wo * (C >> 2);
    const float4* xv = reinterpret_cast<const float4*>(x);
    float4* yv = reinterpret_cast<float4*>(y);
    if (pool) hipLaunchKernelGGL((resample2_kernel<true, float4>), dim3(ew_blocks(n)), dim3(256), 0, stream, xv, Ho, Wo, C >> 2, scale, n, yv);
    else hipLaunchKernelGGL((resample2_kernel<false, float4>), dim3(ew_blocks(n)), dim3(256), 0, stream, xv, Ho, Wo, C >> 2, scale, n, yv);
    return hipGetLastError();
  }
  const size_t n = (size_t)B * Ho * Wo * C;
  if (pool) hipLaunchKernelGGL((resample2_kernel<true, float>), dim3(ew_blocks(n)), dim3(256), 0, stream, x, Ho, Wo, C, scale, n, y);
  else hipLaunchKernelGGL((resample2_kernel<false, float>), dim3(ew_blocks(n)), dim3(256), 0, stream, x, Ho, Wo, C, scale, n, y);
  return hipGetLastError();
}

// per-sample moments in two deterministic levels: grid (chunks, B) partial sums with 16-byte loads, then one small block
// per sample adds its chunks in order.  (A first version used one workgroup per sample: 64 workgroups streaming 1 MB each
// reached 0.27 TB/s.)
constexpr int ROW_CHUNKS_MAX = 64;
static int row_chunks(int64_t per) {
  int64_t c = (per + 8191) / 8192;
  if (c > ROW_CHUNKS_MAX) c = ROW_CHUNKS_MAX;
  return c < 1 ? 1 : (int)c;
}
size_t row_moments_ws(int B) { return (size_t)B * ROW_CHUNKS_MAX * 2 * sizeof(float); }

__global__ __launch_bounds__(256) void row_moments_stage1(const float* __restrict__ a, const float* __restrict__ b, size_t per,
                                                          size_t per_chunk, int vec, float* __restrict__ part) {
  __shared__ float red[2][4];
  const size_t beg = (size_t)blockIdx.x * per_chunk;
  size_t end = beg + per_chunk;
  if (end > per) end = per;
  const float* ra = a + (size_t)blockIdx.y * per;
  const float* rb = b ? b + (size_t)blockIdx.y * per : ra;
  float acc1 = 0.f, acc2 = 0.f;
  if (vec) {       // per, per_chunk multiples of 4 and 16-byte aligned bases
#pragma unroll 4
    for (size_t i = (beg >> 2) + threadIdx.x; i < (end >> 2); i += 256) {
      const float4 v = reinterpret_cast<const float4*>(ra)[i];
      const float4 w = reinterpret_cast<const float4*>(rb)[i];
      acc1 += (v.x + v.y) + (v.z + v.w);
      acc2 += (v.x * w.x + v.y * w.y) + (v.z * w.z + v.w * w.w);
    }
  } else {
    for (size_t i = beg + threadIdx.x; i < end; i += 256) {
      const float v = ra[i];
      acc1 += v;
      acc2 += v * rb[i];
    }
  }
  acc1 = wave_sum(acc1); acc2 = wave_sum(acc2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = acc1; red[1][threadIdx.x >> 6] = acc2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
    o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ void row_moments_stage2(const float* __restrict__ part, int B, int chunks, float* __restrict__ s1, float* __restrict__ s2) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B) return;
  float a = 0.f, b = 0.f;
  for (int k = 0; k < chunks; ++k) { a += part[((size_t)r * chunks + k) * 2]; b += part[((size_t)r * chunks + k) * 2 + 1]; }
  s1[r] = a; s2[r] = b;
}

hipError_t row_moments_launch(const float* a, const float* b, int B, int64_t per, float* s1, float* s2, void* ws, hipStream_t stream) {
  int chunks = row_chunks(per);
  const bool vec = (per & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
  size_t per_chunk = ((size_t)per + chunks - 1) / chunks;
  if (vec) per_chunk = (per_chunk + 3) & ~(size_t)3;
  chunks = (int)(((size_t)per + per_chunk - 1) / per_chunk);
  float* part = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(row_moments_stage1, dim3(chunks, B), dim3(256), 0, stream, a, b, (size_t)per, per_chunk, vec ? 1 : 0, part);
  hipLaunchKernelGGL(row_moments_stage2, dim3((B + 63) / 64), dim3(64), 0, stream, part, B, chunks, s1, s2);
  return hipGetLastError();
}

template <typename T>       // T = float4: n and per are counted in float4 units
__global__ __launch_bounds__(256) void row_fma2_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                       const float* __restrict__ alpha, const float* __restrict__ gamma,
                                                       const float* __restrict__ delta, size_t n, size_t per,
                                                       T* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / per;
    T v = vscale(a[i], alpha[r]);
    if (b) v = vadd(v, vscale(b[i], gamma[r]));
    if (delta) v = vadd(v, vsplat(delta[r], v));
    out[i] = v;
  }
}

hipError_t row_fma2_launch(const float* a, const float* b, const float* alpha, const float* gamma, const float* delta, int B,
                           int64_t per, float* out, hipStream_t stream) {
  const size_t n = (size_t)B * per;
  const bool v4 = (per & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (v4)
    hipLaunchKernelGGL(row_fma2_kernel<float4>, dim3(ew_blocks(n >> 2)), dim3(256), 0, stream, reinterpret_cast<const float4*>(a),
                       reinterpret_cast<const float4*>(b), alpha, gamma, delta, n >> 2, (size_t)per >> 2, reinterpret_cast<float4*>(out));
  else
    hipLaunchKernelGGL(row_fma2_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, stream, a, b, alpha, gamma, delta, n, (size_t)per, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Loss heads: the scalar-per-sample tail of the critic / generator losses in ONE launch each instead of ~40 four-microsecond
// elementwise launches of a tensor library (means, hinges, their backward seeds).
//   wgan_d_head    reference models/wgancls/model.py:72-92: from the 3B logits (fake | real | mismatch) and the two slope
//                  vectors, every loss scalar the trainer logs AND dD_loss/d(logit), dD_loss/d(slope) — the seeds that
//                  start the backward pass.  kt is read from device memory (graph-capturable).
//   ca_kl_fwd/bwd  conditioning augmentation c = mean + exp(log_sigma) * eps with the KL term
//                  mean(-ls + .5(-1 + exp(2 ls) + mean^2)) (model.py:117-127) and the joint backward.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* red) {      // red: 4 floats of LDS; all threads get the sum
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void wgan_d_head_kernel(const float* __restrict__ logits, const float* __restrict__ s1,
                                                          const float* __restrict__ s2, const float* __restrict__ kt_dev,
                                                          int B, float gp_coeff, float use_kt, float* __restrict__ seed_l,
                                                          float* __restrict__ seed_s1, float* __restrict__ seed_s2,
                                                          float* __restrict__ scal) {
  __shared__ float red[4];
  const float kt = use_kt != 0.f ? *kt_dev : 1.0f;
  const float invB = 1.0f / (float)B;
  float af = 0.f, ar = 0.f, am = 0.f, am2 = 0.f, h1 = 0.f, h2 = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float f = logits[i], r = logits[B + i], m = logits[2 * B + i];
    af += f; ar += r; am += m; am2 += m * m;
    const float a = fmaxf(s1[i] - 1.f, 0.f), b = fmaxf(s2[i] - 1.f, 0.f);
    h1 += a * a; h2 += b * b;
    seed_l[i] = invB;                       // D_loss = -wdist - kt*wdist2 + ...,  wdist = real - fake
    seed_l[B + i] = -(1.f + kt) * invB;
    seed_l[2 * B + i] = kt * invB;
    seed_s1[i] = gp_coeff * 2.f * a * invB;
    seed_s2[i] = gp_coeff * 2.f * b * invB;
  }
  af = block_sum256(af, red); ar = block_sum256(ar, red); am = block_sum256(am, red);
  am2 = block_sum256(am2, red); h1 = block_sum256(h1, red); h2 = block_sum256(h2, red);
  if (threadIdx.x == 0) {
    const float fake = af * invB, real = ar * invB, mis = am * invB;
    const float wd = real - fake, wd2 = real - mis, gp1 = h1 * invB, gp2 = h2 * invB;
    const float bal = kt * wd2 - wd;
    scal[0] = -wd - kt * wd2 + gp_coeff * (gp1 + gp2);    // D_loss
    scal[1] = real; scal[2] = fake; scal[3] = mis; scal[4] = wd; scal[5] = wd2; scal[6] = gp1; scal[7] = gp2;
    scal[8] = am2 * invB;                                   // reg_loss
    scal[9] = bal * bal;                                    // balance_loss
    scal[10] = 2.f * bal * wd2;                             // d balance_loss / d kt
    scal[11] = kt;
  }
}

hipError_t wgan_d_head_launch(const float* logits, const float* s1, const float* s2, const float* kt_dev, int B, float gp_coeff,
                              float* seed_l, float* seed_s1, float* seed_s2, float* scal, hipStream_t stream) {
  hipLaunchKernelGGL(wgan_d_head_kernel, dim3(1), dim3(256), 0, stream, logits, s1, s2, kt_dev, B, gp_coeff, kt_dev ? 1.f : 0.f,
                     seed_l, seed_s1, seed_s2, scal);
  return hipGetLastError();
}

// Sigmoid cross-entropy heads (reference models/gancls/trainer.py:20-34, models/stackgan/stageI/trainer.py:53-77): for each of up to three
// logit vectors of B samples  loss_k = mean_i [max(l,0) - l y_k + log1p(exp(-|l|))]  (tf.nn.sigmoid_cross_entropy_with_logits with a constant
// label y_k), total = sum_k w_k loss_k, the backward seeds d total / d l = w_k (sigmoid(l) - y_k) / B and the probabilities sigmoid(l) the
// reference's discriminator returns beside its logits.  One workgroup, one launch — the tensor-library version was ~25 launches per head set.
struct CeHeads { const float* l[3]; float y[3], w[3]; float* seed[3]; float* prob[3]; };

__global__ __launch_bounds__(256) void sigmoid_ce_head_kernel(CeHeads h, int B, float* __restrict__ losses) {
  __shared__ float red[4];
  const float invB = 1.0f / (float)B;
  float total = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (!h.l[k]) { if (threadIdx.x == 0) losses[1 + k] = 0.f; continue; }      // (workgroup-uniform)
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) {
      const float l = h.l[k][i];
      const float e = expf(-fabsf(l));                      // in (0, 1]
      acc += fmaxf(l, 0.f) - l * h.y[k] + log1pf(e);
      const float p = l >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      if (h.seed[k]) h.seed[k][i] = h.w[k] * (p - h.y[k]) * invB;
      if (h.prob[k]) h.prob[k][i] = p;
    }
    const float lk = block_sum256(acc, red) * invB;
    if (threadIdx.x == 0) losses[1 + k] = lk;
    total += h.w[k] * lk;
  }
  if (threadIdx.x == 0) losses[0] = total;
}

hipError_t sigmoid_ce_head_launch(const float* const* l, const float* y, const float* w, float* const* seed, float* const* prob, int B,
                                  float* losses, hipStream_t stream) {
  CeHeads h;
  for (int k = 0; k < 3; ++k) { h.l[k] = l[k]; h.y[k] = y[k]; h.w[k] = w[k]; h.seed[k] = seed[k]; h.prob[k] = prob[k]; }
  hipLaunchKernelGGL(sigmoid_ce_head_kernel, dim3(1), dim3(256), 0, stream, h, B, losses);
  return hipGetLastError();
}

// One workgroup of 1024 threads (the KL term is one scalar over all B*128 elements): a thread's elements are fetched with all
// loads in flight before the first expf (the 256-thread version walked 32 dependent load -> expf rounds: 27 us for 8192
// elements), summed in index order per thread, then lanes by shuffle and the 16 waves in fixed order.
__global__ __launch_bounds__(1024) void ca_kl_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ ls,
                                                         const float* __restrict__ eps, int n, float* __restrict__ code,
                                                         float* __restrict__ kl) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int base = 0; base < n; base += 8 * 1024) {
    float m[8], l[8], z[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = base + j * 1024 + (int)threadIdx.x;
      const bool ok = i < n;
      m[j] = ok ? mean[i] : 0.f; l[j] = ok ? ls[i] : 0.f; z[j] = ok ? eps[i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = base + j * 1024 + (int)threadIdx.x;
      if (i < n) {
        const float e = expf(l[j]);
        code[i] = m[j] + e * z[j];
        acc += -l[j] + 0.5f * (-1.f + e * e + m[j] * m[j]);
      }
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    kl[0] = t / (float)n;
  }
}

__global__ __launch_bounds__(256) void ca_kl_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ ls,
                                                        const float* __restrict__ eps, const float* __restrict__ dcode,
                                                        const float* __restrict__ dkl, int n, float* __restrict__ dmean,
                                                        float* __restrict__ dls) {
  const float k = dkl ? dkl[0] / (float)n : 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float m = mean[i], e = expf(ls[i]);
    const float dc = dcode ? dcode[i] : 0.f;
    dmean[i] = dc + k * m;
    dls[i] = dc * eps[i] * e + k * (e * e - 1.f);
  }
}

hipError_t ca_kl_fwd_launch(const float* mean, const float* ls, const float* eps, int n, float* code, float* kl, hipStream_t stream) {
  hipLaunchKernelGGL(ca_kl_fwd_kernel, dim3(1), dim3(1024), 0, stream, mean, ls, eps, n, code, kl);
  return hipGetLastError();
}

hipError_t ca_kl_bwd_launch(const float* mean, const float* ls, const float* eps, const float* dcode, const float* dkl, int n,
                            float* dmean, float* dls, hipStream_t stream) {
  hipLaunchKernelGGL(ca_kl_bwd_kernel, dim3(ew_blocks((size_t)n)), dim3(256), 0, stream, mean, ls, eps, dcode, dkl, n, dmean, dls);
  return hipGetLastError();
}

// fade-in with the mixing weight in DEVICE memory (so that a captured graph can be replayed with the next iteration's
// alpha): mode 0: out = (1-t)*a + t*b;  mode 1: out = t*a;  mode 2: out = (1-t)*a   (1 and 2 are the backward of 0)
__global__ __launch_bounds__(256) void lerp_dev_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ t_dev, int mode, size_t n,
                                                       float* __restrict__ out) {
  const float t = *t_dev, u = 1.f - t;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = mode == 0 ? u * a[i] + t * b[i] : (mode == 1 ? t * a[i] : u * a[i]);
}

hipError_t lerp_dev_launch(const float* a, const float* b, const float* t_dev, int mode, size_t n, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(lerp_dev_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, a, b, t_dev, mode, n, out);
  return hipGetLastError();
}

}  // namespace t2i
