// t2i_igemm_h.hip — implicit-GEMM convolution on the bf16 matrix pipe with bf16 OPERANDS IN MEMORY (gfx950 only).
//
// BASELINE config 3 ("bf16 MFMA, fp32 accumulate / master").  The first bf16 mode (igemm_kernel<..., MATH=1>) fetched fp32
// operands and rounded them on their way into LDS: it was bound by the fp32 L2 -> LDS stream, twice the bytes the MFMA
// needs, plus a v_cvt per element pair (0.086 of the bf16 peak).  Here the operands are bf16 in memory:
//   * activations: a bf16 copy of the gathered tensor (x for the forward conv, dy for the input gradient / transposed
//     conv), same NHWC layout, written by cast_bf16_kernel into the caller's workspace right before the GEMM (one
//     HBM-bound pass: 6 bytes per element against the >= 2*Cout*taps FLOPs it feeds);
//   * filters: a bf16 copy laid out K-INNER for the GEMM that uses it — [tap][Cout][Cin] for the forward conv (the
//     per-tap transpose of HWIO), [tap][Cin][Cout] (= HWIO itself) for the input gradient — produced by wcast_kernel and
//     kept in the transformed-filter cache until the optimizer changes the filter.
// Both operands are then K-inner: a thread moves 16 bytes (8 consecutive k of one row) from global memory to LDS with
// no arithmetic at all, and every MFMA operand is one ds_read_b128.
//   tile    128x128 (4 waves 2x2, 2x2 accumulators of 32x32 each) or 64x64; BK = 64 (4 MFMA steps of k = 16)
//   LDS     [rows][64 bf16 + 16 B pad]: row stride 36 dwords, conflict-free for the 16-lane groups of ds_read_b128
//   loop    tile t in LDS, tile t+1 in registers, tile t+2 in flight; one barrier per K-tile (512 MFMA cycles per wave)
//   K-tiles never straddle a filter tap (gathered channels % 64 == 0): (kh, kw, c0) are decoded once per tile on the scalar
//   unit; padding taps / ragged rows / split-K tails read zeros through the buffer-load range check.
// Results are those of the first bf16 mode bit for bit in exact arithmetic terms: both round the same fp32 values to bf16
// with RNE and accumulate exact products in fp32; only the summation order inside a K-tile differs (k = 64 per barrier
// instead of 32).  Outputs, bias, activation, split-K slabs and the accumulate-into-arena epilogue are fp32 as before.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "t2i_internal.h"

namespace t2i {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int HBK = 64;                 // K-tile in elements
constexpr int HROW = 36;                // LDS row stride in dwords (64 bf16 = 32 dwords + 4 pad)
constexpr unsigned HOOB = 0xFFFFFFF0u;  // byte offset beyond every legal buffer: the load returns 0

__device__ __forceinline__ unsigned pk2(float lo, float hi) {      // v_cvt_pk_bf16_f32, round to nearest even
  f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// ------------------------------------------------------------------------------------------------------------------
// staging kernels
// ------------------------------------------------------------------------------------------------------------------
// y[i] = bf16(x[i]), n % 8 == 0, both 16-byte aligned: two 16-byte loads -> one 16-byte store per thread and step
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, size_t n8, uint4* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    y[i] = make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
  }
}

// (the filter copies — wcast_kernel, [tap][Cout][Cin] / [tap][Cin][Cout] bf16 — live with the filter cache in t2i_winograd.hip)

hipError_t cast_bf16_launch(const float* x, size_t n, void* y, hipStream_t stream);

// y[i] = float(x[i]) for bf16 x (exact): the staging copy for the few paths that have no bf16-tensor loader (bf16 storage)
__global__ __launch_bounds__(256) void cast_f32_kernel(const uint4* __restrict__ x, size_t n8, float4* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 u = x[i];
    y[2 * i] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
    y[2 * i + 1] = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xFFFF0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xFFFF0000u));
  }
}
__global__ __launch_bounds__(256) void cast_f32_scalar_kernel(const unsigned short* __restrict__ x, size_t n, float* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = __uint_as_float((unsigned)x[i] << 16);
}
__global__ __launch_bounds__(256) void cast_bf16_scalar_kernel(const float* __restrict__ x, size_t n, __bf16* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (__bf16)x[i];
}

hipError_t cast_f32_launch(const void* x, size_t n, float* y, hipStream_t stream) {
  const bool vec = (n & 7) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  size_t items = vec ? (n >> 3) : n;
  size_t blocks = (items + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (vec) hipLaunchKernelGGL(cast_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const uint4*>(x), n >> 3, reinterpret_cast<float4*>(y));
  else hipLaunchKernelGGL(cast_f32_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const unsigned short*>(x), n, y);
  return hipGetLastError();
}

// any n / alignment (the vectorised cast_bf16_launch below needs n % 8 == 0 and 16-byte alignment)
hipError_t cast_bf16_any_launch(const float* x, size_t n, void* y, hipStream_t stream) {
  if ((n & 7) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) return cast_bf16_launch(x, n, y, stream);
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cast_bf16_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, reinterpret_cast<__bf16*>(y));
  return hipGetLastError();
}

hipError_t cast_bf16_launch(const float* x, size_t n, void* y, hipStream_t stream) {
  const size_t n8 = n >> 3;
  size_t blocks = (n8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n8, reinterpret_cast<uint4*>(y));
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// the GEMM
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ PhaseInfo load_phase_h(int idx) {
  PhaseInfo r;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) char* KernArg;
  typedef const __attribute__((address_space(4))) int32_t* Words;
  const Words src = (Words)((KernArg)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(IgemmParams, phase) + (size_t)idx * sizeof(PhaseInfo));
  int32_t* dst = reinterpret_cast<int32_t*>(&r);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(PhaseInfo) / 4); ++i) dst[i] = src[i];
#else
  (void)idx;
  r = PhaseInfo();
#endif
  return r;
}


// ------------------------------------------------------------------------------------------------------------------
// Epilogue through LDS (round 4).  The MFMA leaves a lane with ONE output column and 16 rows of each 32x32 block, so the direct
// epilogue issued 16 WMT WNT stores of 2 or 4 bytes per thread (64 at 128x128) — and a launch-level ablation showed those stores
// costing 7 us of EVERY launch whatever its K (15.4 -> 8.4 us for a one-K-tile GEMM on 256 tiles, profiles/r04_bf16_gemm_fixed_cost.txt):
// the epilogue is store-ISSUE-bound (256 store instructions per CU through one address pipe), not bandwidth-bound.  Here every
// wave passes its sub-tile through a private LDS patch, 32 rows at a time (ds_write_b32 by column, ds_read_b128 by row — LDS
// operations of one wave execute in order, no barrier), and a lane then owns 8 consecutive columns of a row: one 16-byte store
// for a bf16 output (8 per thread at 128x128), two for an fp32 one; bias is added from two 16-byte loads, the accumulate form
// reads its old values the same way.  Element arithmetic (bias, activation, accumulate, RNE to bf16) is the direct epilogue's:
// results are bit-identical.  Needs N % 8 == 0 and 16-byte aligned bases (else: direct epilogue).
// lds: >= 4 * 32 * (32 WNT + 4) floats, no longer read or written by anybody (the caller synchronises).
// ------------------------------------------------------------------------------------------------------------------
template <int MODE, int WMT, int WNT>
__device__ __forceinline__ void store_tile_h(const IgemmParams& p, const PhaseInfo& pi, const f32x16 (&acc)[WMT][WNT], float* lds, int bm, int bn,
                                             int split) {
  constexpr int SROW = 32 * WNT + 4, CH = 4 * WNT;          // staging row stride (floats); 8-column chunks per row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
  float* st = lds + wave * (32 * SROW);
  float* out = p.c + (p.splitk > 1 ? (size_t)split * p.out_elems : 0);
  const bool fused = (p.splitk == 1);
#pragma unroll
  for (int i = 0; i < WMT; ++i) {
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) st[((e & 3) + 8 * (e >> 2) + 4 * lh) * SROW + j * 32 + l31] = acc[i][j][e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 2 * WNT; ++it) {
      const int idx = it * 64 + lane, row = idx / CH, c8 = idx % CH;
      const float4 v0 = *reinterpret_cast<const float4*>(&st[row * SROW + c8 * 8]);
      const float4 v1 = *reinterpret_cast<const float4*>(&st[row * SROW + c8 * 8 + 4]);
      const int m = bm + (wm * WMT + i) * 32 + row;
      const int n = bn + wn * 32 * WNT + c8 * 8;
      bool mok = m < p.M;
      int rowoff;
      if (MODE == MODE_BWD_DATA) {
        const int mm = mok ? m : 0;
        const int b = p.div_hqwq.div(mm);
        const int rem = mm - b * p.hqwq;
        const int ihq = p.div_wq.div(rem);
        const int iwq = rem - ihq * p.Wq;
        const int ih = ihq * p.d.SH + pi.ph, iw = iwq * p.d.SW + pi.pw;
        mok = mok && ih < p.d.H && iw < p.d.W;
        rowoff = ((b * p.d.H + ih) * p.d.W + iw) * p.N;
      } else {
        rowoff = m * p.N;
      }
#if defined(T2I_HEXP) && (T2I_HEXP & 1)
      if (mok && n < p.N && v0.x == 1.2345e38f) {
#else
      if (mok && n < p.N) {
#endif
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (fused) {
          if (p.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = apply_act(v[k], p.act, p.alpha);
          if (p.accumulate) {
            const float4 o0 = *reinterpret_cast<const float4*>(out + rowoff + n), o1 = *reinterpret_cast<const float4*>(out + rowoff + n + 4);
            v[0] += o0.x; v[1] += o0.y; v[2] += o0.z; v[3] += o0.w; v[4] += o1.x; v[5] += o1.y; v[6] += o1.z; v[7] += o1.w;
          }
          if (p.c_h) {
            const uint4 h = {pk2(v[0], v[1]), pk2(v[2], v[3]), pk2(v[4], v[5]), pk2(v[6], v[7])};
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p.c_h) + rowoff + n) = h;
          }
        }
        if (!fused || p.c) {       // bf16 storage: an unsplit launch writes the bf16 tensor only (p.c == NULL)
          *reinterpret_cast<float4*>(out + rowoff + n) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(out + rowoff + n + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Batch-norm statistics from the epilogue of a bf16-operand forward GEMM (round 4; igemm_kernel's scheme, t2i_igemm.hip): per column
// the tile's SUM and its second moment ABOUT THE TILE'S OWN MEAN of the values the batch norm will read — act(acc + bias), rounded
// to bf16 when the output tensor is bf16 — written as p.stats[2][tiles_m][N]; the batch norm merges the tiles with Chan's update
// (t2i_bn_train_fwd_stats) instead of reading the activation again.  Two passes over the accumulators (still in registers); lanes l and
// l ^ 32 hold the same column, the NWM waves stacked along M meet in LDS.  lds: NWM * BN floats, free (the caller synchronises).
template <int WMT, int WNT, int NWM>
__device__ __forceinline__ void tile_stats_h(const IgemmParams& p, const f32x16 (&acc)[WMT][WNT], float* red, int bm, int bn, int tile_m, int tiles_m) {
  constexpr int BM = 32 * WMT * NWM, BN = 64 * WNT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
  const bool round_h = p.c_h != nullptr && p.c == nullptr;       // the tensor the batch norm reads is the bf16 one
  auto value = [&](float a, float bv) __attribute__((always_inline)) {
    float v = apply_act(a + bv, p.act, p.alpha);
    if (round_h) v = __uint_as_float(pk2(v, 0.f) << 16);
    return v;
  };
  float bv[WNT], cs[WNT];
#pragma unroll
  for (int j = 0; j < WNT; ++j) {
    const int n = bn + wn * 32 * WNT + j * 32 + l31;
    bv[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = bm + wm * 32 * WMT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        s += (m < p.M) ? value(acc[i][j][e], bv[j]) : 0.f;
      }
    cs[j] = s + __shfl_xor(s, 32, 64);
    if (lh == 0) red[wm * BN + wn * 32 * WNT + j * 32 + l31] = cs[j];
  }
  __syncthreads();
  const float rows_tile = (float)min(BM, p.M - bm);
  float cq[WNT];
#pragma unroll
  for (int j = 0; j < WNT; ++j) {
    const int col = wn * 32 * WNT + j * 32 + l31;
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < NWM; ++g) tot += red[g * BN + col];
    const float mean = tot / rows_tile;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = bm + wm * 32 * WMT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const float dlt = value(acc[i][j][e], bv[j]) - mean;
        q += (m < p.M) ? dlt * dlt : 0.f;
      }
    cq[j] = q + __shfl_xor(q, 32, 64);
  }
  __syncthreads();                                   // the sums have been read by everyone
  float tsum = 0.f;
  if (tid < BN) {
#pragma unroll
    for (int g = 0; g < NWM; ++g) tsum += red[g * BN + tid];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < WNT; ++j)
    if (lh == 0) red[wm * BN + wn * 32 * WNT + j * 32 + l31] = cq[j];
  __syncthreads();
  if (tid < BN && bn + tid < p.N) {
    float m2 = 0.f;
#pragma unroll
    for (int g = 0; g < NWM; ++g) m2 += red[g * BN + tid];
    p.stats[(size_t)tile_m * p.N + bn + tid] = tsum;
    p.stats[((size_t)tiles_m + tile_m) * p.N + bn + tid] = m2;
  }
  __syncthreads();                                   // red is free again (the epilogue's staging patches overlap it)
}

__device__ __forceinline__ bool vec_epilogue_ok(const IgemmParams& p) {
  return p.vec_epi && (p.N & 7) == 0 &&
         ((reinterpret_cast<uintptr_t>(p.c) | reinterpret_cast<uintptr_t>(p.c_h) | reinterpret_cast<uintptr_t>(p.bias) | (uintptr_t)(p.out_elems & 3)) & 15) == 0;
}

template <int WMT, int WNT>
struct SmemH {
  static constexpr int BM = 64 * WMT, BN = 64 * WNT;
  static constexpr int A_DW = BM * HROW, B_DW = BN * HROW;      // dwords per buffer
  static constexpr int BYTES = 2 * (A_DW + B_DW) * 4;
};

// MODE_FWD:      y [M = B*Ho*Wo, N = Cout]  = im2col(x_h)[M, K = KH*KW*Cin] * w_h[tap][n = co][k = ci]
// MODE_BWD_DATA: dx[M = B*Hq*Wq, N = Cin]   = gather(dy_h)[M, K = taps*Cout] * w_h[tap][n = ci][k = co]   (grid.z = phase)
template <int MODE, int WMT, int WNT>
__global__ __launch_bounds__(256) void igemm_h_kernel(IgemmParams p) {
  using S = SmemH<WMT, WNT>;
  constexpr int BM = S::BM, BN = S::BN;
  extern __shared__ __attribute__((aligned(16))) unsigned smem_h[];
  unsigned* As = smem_h;
  unsigned* Bs = smem_h + 2 * S::A_DW;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // block -> tile: XCD-contiguous runs of tile ids, grouped rasterisation (as igemm_kernel)
  const int tiles_m = p.tiles_m;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tile_m, tile_n;
  {
    const int g = p.group_n;
    const int per_group = g * tiles_m;
    const int grp = bid / per_group;
    const int r = bid - grp * per_group;
    const int n0 = grp * g;
    const int width = min(g, p.tiles_n - n0);
    tile_m = r / width;
    tile_n = n0 + (r - tile_m * width);
  }
  const int bm = tile_m * BM, bn = tile_n * BN;
  const int split = blockIdx.y;
  const PhaseInfo pi = load_phase_h(MODE == MODE_BWD_DATA ? blockIdx.z : 0);
  const int Kdim = (MODE == MODE_BWD_DATA) ? pi.K : p.K;
  const int kbeg = split * p.k_per_split;
  const int kend = min(Kdim, kbeg + p.k_per_split);
  const int ntiles = (kend - kbeg + HBK - 1) / HBK;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), (short)0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), (short)0, (int)p.b_bytes, 0x00020000);

  // ---- loader state: thread -> (16-byte piece kp = tid & 7 of the 128-byte K-tile row, rows (tid >> 3) + 32 i) ------------
  constexpr int A_LD = BM / 32, B_LD = BN / 32;
  const int kp = tid & 7, r0 = tid >> 3;
  const int Csrc = (MODE == MODE_FWD) ? p.d.Cin : p.d.Cout;          // channels of the gathered tensor = k extent of one tap
  const int Wsrc = (MODE == MODE_FWD) ? p.d.W : p.d.Wo;
  const unsigned Hs = (MODE == MODE_FWD) ? p.d.H : p.d.Ho, Ws = (MODE == MODE_FWD) ? p.d.W : p.d.Wo;
  int a_rowoff[A_LD], a_h0[A_LD], a_w0[A_LD];
  bool a_ok[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int m = bm + r0 + 32 * i;
    bool ok = m < p.M;
    const int mm = ok ? m : 0;
    int base, h0, w0;
    if (MODE == MODE_FWD) {
      const int b = p.div_howo.div(mm);
      const int rem = mm - b * p.howo;
      const int oh = p.div_wo.div(rem);
      const int ow = rem - oh * p.d.Wo;
      base = b * p.d.H * p.d.W;
      h0 = oh * p.d.SH - p.d.pad_t;
      w0 = ow * p.d.SW - p.d.pad_l;
    } else {
      const int b = p.div_hqwq.div(mm);
      const int rem = mm - b * p.hqwq;
      const int ihq = p.div_wq.div(rem);
      const int iwq = rem - ihq * p.Wq;
      base = b * p.d.Ho * p.d.Wo;
      h0 = ihq + pi.oh_off;
      w0 = iwq + pi.ow_off;
      ok = ok && (ihq * p.d.SH + pi.ph < p.d.H) && (iwq * p.d.SW + pi.pw < p.d.W);
    }
    a_ok[i] = ok; a_h0[i] = h0; a_w0[i] = w0;
    a_rowoff[i] = (base + h0 * Wsrc + w0) * Csrc + kp * 8;
  }
  int b_rowoff[B_LD];
  bool b_ok[B_LD];
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int n = bn + r0 + 32 * i;
    b_ok[i] = n < p.N;
    b_rowoff[i] = n * Csrc + kp * 8;                   // both filter images are [tap][N][Csrc]
  }

  u32x4 areg[A_LD], breg[B_LD];

  auto load_tile = [&](int t) __attribute__((always_inline)) {
    const int k0 = kbeg + t * HBK;
    const int tap = p.div_c.div(k0);                   // wave-uniform
    const int c0 = k0 - tap * Csrc;
    const bool kok = (k0 + kp * 8) < kend;
    int dh, dw, wtap;
    if (MODE == MODE_FWD) {
      dh = p.div_kw.div(tap);
      dw = tap - dh * p.d.KW;
      wtap = tap;
    } else {
      const int jh = pi.div_ntw.div(tap);
      const int jw = tap - jh * pi.ntw;
      dh = -jh; dw = -jw;
      wtap = (pi.kh0 + jh * p.d.SH) * p.d.KW + (pi.kw0 + jw * p.d.SW);
    }
    const int sa = (dh * Wsrc + dw) * Csrc + c0;
    const int sb = wtap * p.N * Csrc + c0;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const bool ok = a_ok[i] & kok & ((unsigned)(a_h0[i] + dh) < Hs) & ((unsigned)(a_w0[i] + dw) < Ws);
      areg[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? (unsigned)(a_rowoff[i] + sa) * 2u : HOOB, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      breg[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, (b_ok[i] & kok) ? (unsigned)(b_rowoff[i] + sb) * 2u : HOOB, 0, 0);
  };

  auto store_tile = [&](int buf) __attribute__((always_inline)) {
    unsigned* as = As + buf * S::A_DW;
    unsigned* bs = Bs + buf * S::B_DW;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) *reinterpret_cast<u32x4*>(&as[(r0 + 32 * i) * HROW + kp * 4]) = areg[i];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) *reinterpret_cast<u32x4*>(&bs[(r0 + 32 * i) * HROW + kp * 4]) = breg[i];
  };

  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_tile(0);
  store_tile(0);
  load_tile(1);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const unsigned* as = As + (t & 1) * S::A_DW;
    const unsigned* bs = Bs + (t & 1) * S::B_DW;
    bf16x8 fa[WMT][4], fb[WNT][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < WMT; ++i)
        fa[i][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&as[(wm * 32 * WMT + i * 32 + l31) * HROW + s * 8 + lh * 4]));
#pragma unroll
      for (int i = 0; i < WNT; ++i)
        fb[i][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&bs[(wn * 32 * WNT + i * 32 + l31) * HROW + s * 8 + lh * 4]));
    }
    store_tile((t + 1) & 1);          // tile t+1 (registers) -> the other buffer; every wave finished reading it before the
    load_tile(t + 2);                 // barrier that ended iteration t-1
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int n = 0; n < WNT; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][s], fb[n][s], acc[i][n], 0, 0, 0);
    __syncthreads();
  }

  // ---- epilogue (same conventions as igemm_kernel) ------------------------------------------------------------------------
  if (vec_epilogue_ok(p)) {       // launch-uniform; the K loop's last barrier has freed the LDS
    store_tile_h<MODE, WMT, WNT>(p, pi, acc, reinterpret_cast<float*>(smem_h), bm, bn, split);
    return;
  }
  float* out = p.c + (p.splitk > 1 ? (size_t)split * p.out_elems : 0);     // (never dereferenced when p.c is NULL: see below)
  const bool fused = (p.splitk == 1);
#pragma unroll
  for (int i = 0; i < WMT; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = bm + wm * 32 * WMT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      bool mok = m < p.M;
      int rowoff;
      if (MODE == MODE_BWD_DATA) {
        const int mm = mok ? m : 0;
        const int b = p.div_hqwq.div(mm);
        const int rem = mm - b * p.hqwq;
        const int ihq = p.div_wq.div(rem);
        const int iwq = rem - ihq * p.Wq;
        const int ih = ihq * p.d.SH + pi.ph, iw = iwq * p.d.SW + pi.pw;
        mok = mok && ih < p.d.H && iw < p.d.W;
        rowoff = ((b * p.d.H + ih) * p.d.W + iw) * p.N;
      } else {
        rowoff = m * p.N;
      }
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int n = bn + wn * 32 * WNT + j * 32 + l31;
#if defined(T2I_HEXP) && (T2I_HEXP & 1)
        if (mok && n < p.N && acc[i][j][e] == 1.2345e38f) {
#else
        if (mok && n < p.N) {
#endif
          float v = acc[i][j][e];
          if (fused) {
            if (p.bias) v += p.bias[n];
            v = apply_act(v, p.act, p.alpha);
            if (p.accumulate) v += out[rowoff + n];
            if (p.c_h) reinterpret_cast<__bf16*>(p.c_h)[rowoff + n] = (__bf16)v;
          }
          if (!fused || p.c) out[rowoff + n] = v;       // bf16 storage: an unsplit launch writes the bf16 tensor only (p.c == NULL)
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same GEMM with the operand tiles moved by the LDS DMA (buffer_load_dwordx4 ... lds): global memory -> LDS without the
// VGPR round trip and without ds_write_b128 (13 LDS-path cycles per wave instruction: with two 128x128 workgroups per CU the
// staging stores alone took 830 of the 1024 MFMA cycles of a K-tile round; timing-only ablation, profiles/r03_bf16_gemm_ablation.txt:
// the staging path costs 35 % of the kernel at B = 512).
//   LDS image   [rows][64 bf16 = 128 B], NO padding (the DMA writes lane-linear: [wave-uniform base] + lane * 16), the eight
//               16-byte k-chunks of a row XOR-swizzled: chunk position q of row r holds the row's chunk q ^ ((r >> 1) & 7).
//               The loader applies the XOR to its SOURCE address, the fragment reads to the LDS address; each 16-lane group
//               of a ds_read_b128 then touches 16 distinct 16-byte bank groups (conflict-free, like the padded image).
//   pipeline    two LDS buffers, tile t in buf[t & 1].  Iteration t: all fragment reads of tile t, barrier (buf[t & 1] is free),
//               DMA of tile t+2 into it, 16 MFMAs, s_waitcnt vmcnt(<pieces of one tile>) = tile t+1 has landed (loads complete
//               in order), barrier.  Two barriers per K-tile instead of one, but two tiles in flight and no staging registers.
//   The DMA is issued from inline assembly: through the builtin hipcc treats the LDS write as a dependency of every later
//   ds_read and waits vmcnt(0) before the next fragment read.  Out-of-range lanes (padding taps, ragged rows, K tails) write zeros.
// ------------------------------------------------------------------------------------------------------------------
typedef int i32x4h __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_h;

__device__ __forceinline__ i32x4h rsrc_words_h(const void* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  i32x4h r = {(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
  return r;
}
__device__ __forceinline__ void dma16_h(i32x4h rsrc, unsigned voff, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_base), "v"(voff), "s"(rsrc) : "memory", "m0");
}

template <int WMT, int WNT, int NBUF = 2>
struct SmemD {
  static constexpr int BM = 64 * WMT, BN = 64 * WNT;
  static constexpr int ROW = 32;                                // dwords per row, unpadded
  static constexpr int A_DW = BM * ROW, B_DW = BN * ROW;
  static constexpr int BYTES = NBUF * (A_DW + B_DW) * 4;
};

// The body is a device function of (params, block coordinates) so that TWO independent GEMMs can share one launch
// (igemm_pair_kernel below); igemm_hd_kernel is the plain one-GEMM launch.  bx / nbx: tile block and their number, by: K split,
// bz: stride phase (input gradient).  `p` must sit at kernarg offset 0 (load_phase_h).
// NW: waves per workgroup, 4 (2 x 2: tile 64 WMT x 64 WNT) or 8 (4 x 2, round 4: tile 128 WMT x 64 WNT — 256 x 128 with 64 x 64 wave
// tiles: per multiply-add 3/4 of the staged bytes and DMA instructions of the 128 x 128 tile, and two waves per SIMD, so that one
// wave's DMA issue and fragment reads sit beside the other's MFMAs; for GEMMs with >= 256 such tiles: big maps, B >= 256).
template <int MODE, int WMT, int WNT, int PIPE, int NW = 4>      // PIPE: 0 = two-phase K loop; 2, 3, 4 = software-pipelined loop with that many LDS buffers
__device__ __forceinline__ void hd_body(const IgemmParams& p, const int bx, const int nbx, const int by, const int bz) {
  constexpr bool PINGPONG = (PIPE == 8);                // 8 waves only: the two waves of a SIMD alternate between a load phase and a multiply phase
  constexpr bool PAIRED = (PIPE == 6);                  // two K-tiles per barrier pair, four one-tile LDS buffers (two stages)
  constexpr int NBUF = PINGPONG ? 3 : (PAIRED ? 4 : (PIPE > 2 ? PIPE : 2));
  using S = SmemD<WMT * (NW / 4), WNT, NBUF>;
  constexpr int RPP = NW * 8;                           // rows staged per pass: 8 threads per 128-byte row
  constexpr int BM = S::BM, BN = S::BN, ROW = S::ROW;
  extern __shared__ __attribute__((aligned(16))) unsigned smem_h[];
  unsigned* As = smem_h;
  unsigned* Bs = smem_h + NBUF * S::A_DW;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  const int tiles_m = p.tiles_m;
  int bid = bx;
  {
    const int nblk = nbx, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tile_m, tile_n;
  {
    const int g = p.group_n;
    const int per_group = g * tiles_m;
    const int grp = bid / per_group;
    const int r = bid - grp * per_group;
    const int n0 = grp * g;
    const int width = min(g, p.tiles_n - n0);
    tile_m = r / width;
    tile_n = n0 + (r - tile_m * width);
  }
  const int bm = tile_m * BM, bn = tile_n * BN;
  const int split = by;
  const PhaseInfo pi = load_phase_h(MODE == MODE_BWD_DATA ? bz : 0);
  const int Kdim = (MODE == MODE_BWD_DATA) ? pi.K : p.K;
  const int kbeg = split * p.k_per_split;
  const int kend = min(Kdim, kbeg + p.k_per_split);
#if defined(T2I_HEXP) && (T2I_HEXP & 2)
  const int ntiles = 0;
#else
  const int ntiles = (kend - kbeg + HBK - 1) / HBK;
#endif

  const i32x4h wa = rsrc_words_h(p.a, p.a_bytes), wb = rsrc_words_h(p.b, p.b_bytes);

  // ---- loader state: thread -> (LDS chunk position kp of the 128-byte K-tile row, rows (tid >> 3) + RPP i); it FETCHES chunk kg ----
  constexpr int A_LD = BM / RPP, B_LD = BN / RPP;
  const int kp = tid & 7, r0 = tid >> 3;
  const int kg = kp ^ ((r0 >> 1) & 7);               // rows r0 and r0 + 32 i swizzle alike
  const int Csrc = (MODE == MODE_FWD) ? p.d.Cin : p.d.Cout;
  const int Wsrc = (MODE == MODE_FWD) ? p.d.W : p.d.Wo;
  const unsigned Hs = (MODE == MODE_FWD) ? p.d.H : p.d.Ho, Ws = (MODE == MODE_FWD) ? p.d.W : p.d.Wo;
  // Per-row loader state (round 4, second pass: the per-K-tile address arithmetic was 19 % of the loop — profiles/r04_bf16_loop_ablation.txt):
  //   a_off2   byte offset of the row's pixel at tap (0, 0), this thread's 16-byte chunk included (may be "negative": padding)
  //   a_hw     which taps are inside the image for this row: bit jh = row h0 +- jh is, bit 16 + jw = column w0 +- jw is; 0 = no such row
  //   b_off2   byte offset of filter row n (poisoned beyond the buffer for n >= N: adding a tap offset < 2^30 keeps it there)
  // A K-tile then costs one add, one and, one compare and one select per A row and one add per B row; tap decode is scalar.
  unsigned a_off2[A_LD], a_hw[A_LD], b_off2[B_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int m = bm + r0 + RPP * i;
    bool ok = m < p.M;
    const int mm = ok ? m : 0;
    int base, h0, w0;
    if (MODE == MODE_FWD) {
      const int b = p.div_howo.div(mm);
      const int rem = mm - b * p.howo;
      const int oh = p.div_wo.div(rem);
      const int ow = rem - oh * p.d.Wo;
      base = b * p.d.H * p.d.W;
      h0 = oh * p.d.SH - p.d.pad_t;
      w0 = ow * p.d.SW - p.d.pad_l;
    } else {
      const int b = p.div_hqwq.div(mm);
      const int rem = mm - b * p.hqwq;
      const int ihq = p.div_wq.div(rem);
      const int iwq = rem - ihq * p.Wq;
      base = b * p.d.Ho * p.d.Wo;
      h0 = ihq + pi.oh_off;
      w0 = iwq + pi.ow_off;
      ok = ok && (ihq * p.d.SH + pi.ph < p.d.H) && (iwq * p.d.SW + pi.pw < p.d.W);
    }
    // taps j in [lo, hi) are inside: forward h0 + j in [0, Hs); input gradient h0 - j in [0, Hs)
    const int nh = (MODE == MODE_FWD) ? p.d.KH : pi.nth, nw = (MODE == MODE_FWD) ? p.d.KW : pi.ntw;    // <= 15 each (host side)
    auto taps_in = [](int x0, int size, int n) __attribute__((always_inline)) {
      const int lo = (MODE == MODE_FWD) ? max(0, -x0) : max(0, x0 - size + 1);
      const int hi = (MODE == MODE_FWD) ? min(n, size - x0) : min(n, x0 + 1);
      return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    };
    a_hw[i] = ok ? (taps_in(h0, (int)Hs, nh) | (taps_in(w0, (int)Ws, nw) << 16)) : 0u;
    a_off2[i] = (unsigned)((base + h0 * Wsrc + w0) * Csrc + kg * 8) * 2u;
  }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int n = bn + r0 + RPP * i;
    b_off2[i] = n < p.N ? (unsigned)(n * Csrc + kg * 8) * 2u : 0xC0000000u;
  }
  const unsigned lds_a0 = (unsigned)(size_t)(lds_ptr_h)(As + wave_u * 8 * ROW);   // this wave's 8 rows of row group 0, buffer 0
  const unsigned lds_b0 = (unsigned)(size_t)(lds_ptr_h)(Bs + wave_u * 8 * ROW);

  struct TileAddr { unsigned sa2, sel, sb2; };
  auto tile_addr = [&](int t) __attribute__((always_inline)) {
    const int k0 = kbeg + t * HBK;
    const int tap = p.div_c.div(k0);                   // wave-uniform
    const int c0 = k0 - tap * Csrc;
    const bool kok = (kg * 8) < (kend - k0);           // K tails and the look-ahead tiles past the end read zeros
    int dh, dw, wtap, jh, jw;
    if (MODE == MODE_FWD) {
      jh = p.div_kw.div(tap);
      jw = tap - jh * p.d.KW;
      dh = jh; dw = jw;
      wtap = tap;
    } else {
      jh = pi.div_ntw.div(tap);
      jw = tap - jh * pi.ntw;
      dh = -jh; dw = -jw;
      wtap = (pi.kh0 + jh * p.d.SH) * p.d.KW + (pi.kw0 + jw * p.d.SW);
    }
    const unsigned sa2 = (unsigned)((dh * Wsrc + dw) * Csrc + c0) * 2u;
    const unsigned sel = kok ? ((1u << (jh & 15)) | (0x10000u << (jw & 15))) : 0xFFFFFFFFu;      // all ones never matches (bit 15 is never set)
    const unsigned sb2 = kok ? (unsigned)(wtap * p.N * Csrc + c0) * 2u : 0xC0000000u;
    TileAddr ta = {sa2, sel, sb2};
    return ta;
  };
  // piece q of a K-tile: q < A_LD an A row group, else a B row group
  auto dma_piece = [&](const TileAddr& ta, int q, int buf) __attribute__((always_inline)) {
    if (q < A_LD)
      dma16_h(wa, (a_hw[q] & ta.sel) == ta.sel ? a_off2[q] + ta.sa2 : HOOB, lds_a0 + (unsigned)(buf * S::A_DW + RPP * q * ROW) * 4u);
    else
      dma16_h(wb, b_off2[q - A_LD] + ta.sb2, lds_b0 + (unsigned)(buf * S::B_DW + RPP * (q - A_LD) * ROW) * 4u);
  };
  auto dma_tile = [&](int t, int buf) __attribute__((always_inline)) {
    const TileAddr ta = tile_addr(t);
#pragma unroll
    for (int q = 0; q < A_LD + B_LD; ++q) dma_piece(ta, q, buf);
  };

  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  constexpr int PIECES = A_LD + B_LD;
#pragma unroll
  for (int b = 0; b < NBUF; ++b) dma_tile(b, b);
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PIECES * (PAIRED ? 2 : NBUF - 1)) : "memory");
  __syncthreads();                                     // tile 0 is in LDS (paired loop: tiles 0 and 1)
  if constexpr (PAIRED) {
    // Two K-tiles per barrier pair (round 4, second pass; single-round launches: one workgroup per CU).  The per-K-tile chain
    // (barrier -> DMA issue -> MFMAs -> wait for the next tile -> barrier -> fragment-read latency, profiles/r04_bf16_loop_ablation.txt)
    // is paid once per TWO tiles: stage st = i & 1 holds tiles 2i and 2i+1 in buffers 2 st, 2 st + 1.  Iteration i: the fragments
    // of both tiles are requested back to back, tile 2i multiplies while tile 2i+1's fragments arrive, barrier (the stage is free),
    // then the 2 x PIECES DMA pieces of tiles 2i+4, 2i+5 are issued ONE IN FRONT OF EACH MFMA of tile 2i+1 (the matrix pipe never
    // waits for a burst of DMA issue), wait for this wave's pieces of the other stage (issued an iteration ago), barrier.
    // Same k order per output element as every other loop: bit-identical.
    auto read_tile = [&](const unsigned* as, const unsigned* bs, bf16x8 (&fa)[WMT][4], bf16x8 (&fb)[WNT][4]) __attribute__((always_inline)) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int i = 0; i < WMT; ++i) {
          const int row = wm * 32 * WMT + i * 32 + l31;
          fa[i][s4] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&as[row * ROW + (((2 * s4 + lh) ^ ((row >> 1) & 7)) << 2)]));
        }
#pragma unroll
        for (int i = 0; i < WNT; ++i) {
          const int row = wn * 32 * WNT + i * 32 + l31;
          fb[i][s4] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&bs[row * ROW + (((2 * s4 + lh) ^ ((row >> 1) & 7)) << 2)]));
        }
      }
    };
    const int npairs = (ntiles + 1) >> 1;
    for (int i = 0; i < npairs; ++i) {
      const int st = i & 1;
      bf16x8 fa0[WMT][4], fb0[WNT][4], fa1[WMT][4], fb1[WNT][4];
      read_tile(As + (2 * st) * S::A_DW, Bs + (2 * st) * S::B_DW, fa0, fb0);
      read_tile(As + (2 * st + 1) * S::A_DW, Bs + (2 * st + 1) * S::B_DW, fa1, fb1);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int m = 0; m < WMT; ++m)
#pragma unroll
          for (int n = 0; n < WNT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[m][s4], fb0[n][s4], acc[m][n], 0, 0, 0);
      __syncthreads();                                   // every wave holds the fragments of both tiles: stage st is free
      const TileAddr t0 = tile_addr(2 * i + 4), t1 = tile_addr(2 * i + 5);     // past the end: zeros, never read
      constexpr int NMMA = 4 * WMT * WNT;
      if ((2 * i + 1) < ntiles) {
#pragma unroll
        for (int q = 0; q < NMMA; ++q) {                 // the 2 PIECES pieces spread over the NMMA multiply instructions
#pragma unroll
          for (int d = q * 2 * PIECES / NMMA; d < (q + 1) * 2 * PIECES / NMMA; ++d) {
            if (d < PIECES) dma_piece(t0, d, 2 * st);
            else dma_piece(t1, d - PIECES, 2 * st + 1);
          }
          const int s4 = q / (WMT * WNT), m = (q / WNT) % WMT, n = q % WNT;
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[m][s4], fb1[n][s4], acc[m][n], 0, 0, 0);
        }
      } else {                                           // odd number of K-tiles: the last pair's second tile does not exist
#pragma unroll
        for (int d = 0; d < 2 * PIECES; ++d) {
          if (d < PIECES) dma_piece(t0, d, 2 * st);
          else dma_piece(t1, d - PIECES, 2 * st + 1);
        }
      }
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * PIECES) : "memory");     // this wave's pieces of tiles 2i+2, 2i+3 have landed
      __syncthreads();
    }
  } else
  if constexpr (PINGPONG) {
    // Ping-pong K loop (round 4, 8 waves).  Waves w and w + 4 share a SIMD.  Every K-tile is two phases separated by workgroup
    // barriers; in each phase one half of the waves (a "group") reads its fragments of a tile from LDS and issues its DMA pieces
    // of the tile two ahead — no MFMA — while the other half multiplies the tile it read a phase earlier:
    //   phase 2t     group A: read(t), DMA(t+2), wait for its pieces of t+1        group B: MFMA(t-1)
    //   phase 2t+1   group A: MFMA(t)                                              group B: read(t), DMA(t+2), wait for its pieces of t+1
    // so the matrix pipe of a SIMD always has one wave feeding it while its partner pays for the DMA issue and the LDS reads
    // (which is what bounds the lock-step loops above).  Three LDS buffers: tile t+2 is written while tile t is still being read by
    // the lagging group and tile t+1 is landing.  Hazards: a buffer is re-filled two phases after its last read at the earliest (a
    // barrier in between); a tile is read only after BOTH groups waited for their own pieces of it (vmcnt) in front of a barrier.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // prologue: tiles 0, 1 (and 2: zeros or data) issued by everybody, all landed
    __syncthreads();
    const bool lag = wave_u >= 4;                          // group B runs one phase behind
    if (lag) __syncthreads();
    int cur = 0;
    for (int t = 0; t < ntiles; ++t) {
      const unsigned* as = As + cur * S::A_DW;
      const unsigned* bs = Bs + cur * S::B_DW;
      bf16x8 fa[WMT][4], fb[WNT][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < WMT; ++i) {
          const int row = wm * 32 * WMT + i * 32 + l31;
          fa[i][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&as[row * ROW + (((2 * s + lh) ^ ((row >> 1) & 7)) << 2)]));
        }
#pragma unroll
        for (int i = 0; i < WNT; ++i) {
          const int row = wn * 32 * WNT + i * 32 + l31;
          fb[i][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&bs[row * ROW + (((2 * s + lh) ^ ((row >> 1) & 7)) << 2)]));
        }
      }
      if (t > 0) {                                         // (t = 0: tile 2 went out with the prologue)
        const int tgt = (cur == 0) ? 2 : cur - 1;          // (t + 2) % 3 == (t - 1) % 3: the buffer of tile t-1, read two phases ago at the latest
        dma_tile(t + 2, tgt);
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PIECES) : "memory");      // this wave's pieces of tile t+1 have landed
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                                     // end of this group's load phase
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
          for (int n = 0; n < WNT; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][s], fb[n][s], acc[i][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                                     // end of this group's multiply phase
      __builtin_amdgcn_sched_barrier(0);
      cur = (cur == 2) ? 0 : cur + 1;
    }
    if (!lag) __syncthreads();                             // the leading group waits out the lagging group's last phase
  } else
  if constexpr (PIPE) {
    // Software-pipelined K loop (round 4).  With ONE workgroup on a CU (<= 256 tiles: every large layer at B = 64) nothing hides a
    // phase in which all four waves only read fragments, so the fragments of 16-k step s+1 are read while step s multiplies, and
    // ONE barrier per K-tile sits between steps 2 and 3: there every wave holds the last fragments of tile t (buf[t & 1] is free
    // for the DMA of tile t+2) and has seen its own pieces of tile t+1 land (vmcnt(0): only they are outstanding), so behind the
    // barrier tile t+1 is complete and its step-0 fragments are fetched under the step-3 MFMAs of tile t.
    auto read_step = [&](const unsigned* as, const unsigned* bs, int s, bf16x8 (&fa)[WMT], bf16x8 (&fb)[WNT]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < WMT; ++i) {
        const int row = wm * 32 * WMT + i * 32 + l31;
        fa[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&as[row * ROW + (((2 * s + lh) ^ ((row >> 1) & 7)) << 2)]));
      }
#pragma unroll
      for (int i = 0; i < WNT; ++i) {
        const int row = wn * 32 * WNT + i * 32 + l31;
        fb[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&bs[row * ROW + (((2 * s + lh) ^ ((row >> 1) & 7)) << 2)]));
      }
    };
    auto mma_step = [&](const bf16x8 (&fa)[WMT], const bf16x8 (&fb)[WNT]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int n = 0; n < WNT; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[n], acc[i][n], 0, 0, 0);
    };
    bf16x8 ca[WMT], cb[WNT];
    read_step(As, Bs, 0, ca, cb);
    int cur = 0;                                         // t % NBUF
    for (int t = 0; t < ntiles; ++t) {
      const int nxt = (cur + 1 == NBUF) ? 0 : cur + 1;
      const unsigned* as = As + cur * S::A_DW;
      const unsigned* bs = Bs + cur * S::B_DW;
      const unsigned* nas = As + nxt * S::A_DW;
      const unsigned* nbs = Bs + nxt * S::B_DW;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        bf16x8 na[WMT], nb[WNT];
        read_step(as, bs, s + 1, na, nb);
        mma_step(ca, cb);
#pragma unroll
        for (int i = 0; i < WMT; ++i) ca[i] = na[i];
#pragma unroll
        for (int i = 0; i < WNT; ++i) cb[i] = nb[i];
      }
      // this wave's pieces of tile t+1 have landed (loads complete in order; tiles t+2 .. t+NBUF-1 may still be in flight)
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PIECES * (NBUF - 2)) : "memory");
      __syncthreads();                                   // ... everybody's have, and everybody holds the step-3 fragments of tile t
      dma_tile(t + NBUF, cur);                           // past the end: zeros (k >= kend)
      {
        bf16x8 na[WMT], nb[WNT];
        read_step(nas, nbs, 0, na, nb);                  // past the end: the zeros of the look-ahead DMA, never multiplied
        mma_step(ca, cb);
#pragma unroll
        for (int i = 0; i < WMT; ++i) ca[i] = na[i];
#pragma unroll
        for (int i = 0; i < WNT; ++i) cb[i] = nb[i];
      }
      cur = nxt;
    }
  } else {
#if defined(T2I_HEXP) && (T2I_HEXP & 8)
  bf16x8 fa[WMT][4], fb[WNT][4];
#endif
  for (int t = 0; t < ntiles; ++t) {
    const unsigned* as = As + (t & 1) * S::A_DW;
    const unsigned* bs = Bs + (t & 1) * S::B_DW;
#if defined(T2I_HEXP) && (T2I_HEXP & 8)      // timing only: the fragments are read once
    if (t == 0)
#else
    bf16x8 fa[WMT][4], fb[WNT][4];
#endif
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < WMT; ++i) {
        const int row = wm * 32 * WMT + i * 32 + l31;
        fa[i][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&as[row * ROW + (((2 * s + lh) ^ ((row >> 1) & 7)) << 2)]));
      }
#pragma unroll
      for (int i = 0; i < WNT; ++i) {
        const int row = wn * 32 * WNT + i * 32 + l31;
        fb[i][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&bs[row * ROW + (((2 * s + lh) ^ ((row >> 1) & 7)) << 2)]));
      }
    }
#if !(defined(T2I_HEXP) && (T2I_HEXP & 64))
    __syncthreads();                                   // every wave holds its fragments of tile t: buf[t & 1] is free
#endif
#if !(defined(T2I_HEXP) && (T2I_HEXP & 16))
    dma_tile(t + 2, t & 1);                            // past the end: zeros (k >= kend), never read
#endif
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int n = 0; n < WNT; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][s], fb[n][s], acc[i][n], 0, 0, 0);
    // tile t+1 (issued an iteration ago) has landed when at most this iteration's pieces are outstanding
    if constexpr (A_LD + B_LD == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (A_LD + B_LD == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the look-ahead DMAs of the last two iterations

  // ---- epilogue: igemm_h_kernel's ---------------------------------------------------------------------------------------------
  if (vec_epilogue_ok(p)) {
    __syncthreads();              // every wave's look-ahead DMAs have landed and nobody reads fragments any more: the LDS is free
    if (MODE == MODE_FWD && p.stats != nullptr && p.splitk == 1)       // (the host asks for statistics only on this path)
      tile_stats_h<WMT, WNT, NW / 2>(p, acc, reinterpret_cast<float*>(smem_h), bm, bn, tile_m, tiles_m);
    store_tile_h<MODE, WMT, WNT>(p, pi, acc, reinterpret_cast<float*>(smem_h), bm, bn, split);
    return;
  }
  float* out = p.c + (p.splitk > 1 ? (size_t)split * p.out_elems : 0);
  const bool fused = (p.splitk == 1);
#pragma unroll
  for (int i = 0; i < WMT; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = bm + wm * 32 * WMT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      bool mok = m < p.M;
      int rowoff;
      if (MODE == MODE_BWD_DATA) {
        const int mm = mok ? m : 0;
        const int b = p.div_hqwq.div(mm);
        const int rem = mm - b * p.hqwq;
        const int ihq = p.div_wq.div(rem);
        const int iwq = rem - ihq * p.Wq;
        const int ih = ihq * p.d.SH + pi.ph, iw = iwq * p.d.SW + pi.pw;
        mok = mok && ih < p.d.H && iw < p.d.W;
        rowoff = ((b * p.d.H + ih) * p.d.W + iw) * p.N;
      } else {
        rowoff = m * p.N;
      }
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int n = bn + wn * 32 * WNT + j * 32 + l31;
#if defined(T2I_HEXP) && (T2I_HEXP & 1)
        if (mok && n < p.N && acc[i][j][e] == 1.2345e38f) {
#else
        if (mok && n < p.N) {
#endif
          float v = acc[i][j][e];
          if (fused) {
            if (p.bias) v += p.bias[n];
            v = apply_act(v, p.act, p.alpha);
            if (p.accumulate) v += out[rowoff + n];
            if (p.c_h) reinterpret_cast<__bf16*>(p.c_h)[rowoff + n] = (__bf16)v;
          }
          if (!fused || p.c) out[rowoff + n] = v;
        }
      }
    }
  }
}

template <int MODE, int WMT, int WNT, int PIPE>
__global__ __launch_bounds__(256) void igemm_hd_kernel(IgemmParams p) {
  hd_body<MODE, WMT, WNT, PIPE>(p, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z);
}

// 8 waves (4 x 2).  WMT = 2: tile 256 x 128 (see hd_body).  WMT = 1: the 128 x 128 tile with 32 x 64 wave tiles — two waves per SIMD on
// the tile the planner already uses: a wave's K-tile is then 8 MFMAs, 12 fragment reads and 4 DMA pieces, and while one wave of
// a SIMD pays for its address arithmetic, LDS reads and DMA issue, the other's MFMAs run (the 4-wave loop pays them one after the
// other: profiles/r04_bf16_loop_ablation.txt).  Same k order per output element: bit-identical to the 4-wave kernel.
template <int MODE, int WMT, int PIPE>
__global__ __launch_bounds__(512) void igemm_hd8_kernel(IgemmParams p) {
  hd_body<MODE, WMT, 2, PIPE, 8>(p, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z);
}

// ------------------------------------------------------------------------------------------------------------------
// Filter gradient with bf16 operands in memory:
//   dw[M = KH*KW*Cin, N = Cout] = sum over k = (b, oh, ow) of  x_h[b, oh*SH-pad_t+kh, ow*SW-pad_l+kw, ci] * dy_h[b, oh, ow, co]
// Both operands are "N-inner" in memory (channels contiguous, the reduction index k is the pixel), while the MFMA wants 8
// consecutive k per lane: the loaders transpose in registers.  A thread fetches, for ONE group of 8 channels, KPT
// consecutive pixels of one output row (16 bytes each), regroups the 16-bit halves with v_perm_b32 into per-channel runs
// of KPT consecutive k, and writes each run to its channel's LDS row (ds_write_b64 / b32).  The M tile lies inside one
// filter tap (Cin % BM == 0), so (kh, kw, ci0) are tile constants and a K-tile is 64 consecutive output pixels.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned perm_lo(unsigned k0, unsigned k1) { return __builtin_amdgcn_perm(k1, k0, 0x05040100u); }   // (k0.lo, k1.lo)
__device__ __forceinline__ unsigned perm_hi(unsigned k0, unsigned k1) { return __builtin_amdgcn_perm(k1, k0, 0x07060302u); }   // (k0.hi, k1.hi)

template <int WMT, int WNT>
__global__ __launch_bounds__(256) void igemm_h_filter_kernel(IgemmParams p) {
  using S = SmemH<WMT, WNT>;
  constexpr int BM = S::BM, BN = S::BN;
  constexpr int A_G = BM / 8, B_G = BN / 8;            // groups of 8 channels per tile row block
  constexpr int A_KL = 256 / A_G, B_KL = 256 / B_G;    // threads along k
  constexpr int A_KPT = HBK / A_KL, B_KPT = HBK / B_KL;   // consecutive pixels per thread: 4 (128-wide) or 2 (64-wide)
  extern __shared__ __attribute__((aligned(16))) unsigned smem_h[];
  unsigned* As = smem_h;
  unsigned* Bs = smem_h + 2 * S::A_DW;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
  const int bm = tile_m * BM, bn = tile_n * BN;
  const int split = blockIdx.y;
  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int ntiles = (kend - kbeg + HBK - 1) / HBK;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), (short)0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), (short)0, (int)p.b_bytes, 0x00020000);

  // tile constants: the filter tap and first input channel of this M tile
  const int tap = p.div_c.div(bm);
  const int ci0 = bm - tap * p.d.Cin;
  const int kh = p.div_kw.div(tap), kw = tap - kh * p.d.KW;
  // thread -> (channel group g, k lane kl).  A wave holds ALL k lanes and 64/KL adjacent groups: its transposing LDS stores
  // (rows 8 apart = 32 banks apart at the 36-dword row stride) then spread over all banks through the k offset — with the
  // groups fastest over lanes instead (fully coalesced 256-byte global reads) the same stores were 8-way bank conflicted and
  // the kernel was bound by LDS writes; a wave's global reads are still whole 64-byte (128-wide tile) / 32-byte segments.
  constexpr int A_GL = 64 / A_KL, B_GL = 64 / B_KL;
  const int a_g = (lane % A_GL) + wave * A_GL, a_kl = lane / A_GL;
  const int b_g = (lane % B_GL) + wave * B_GL, b_kl = lane / B_GL;
  const bool b_nok = (bn + b_g * 8) < p.N;

  u32x4 areg[A_KPT], breg[B_KPT];

  auto load_tile = [&](int t) __attribute__((always_inline)) {
    {
      const int k = kbeg + t * HBK + a_kl * A_KPT;          // A_KPT | 4 | Wo: the thread's pixels share one output row
      const bool kok = k < kend;
      const int kk = kok ? k : 0;
      const int b = p.div_howo.div(kk);
      const int rem = kk - b * p.howo;
      const int oh = p.div_wo.div(rem);
      const int ow = rem - oh * p.d.Wo;
      const int ih = oh * p.d.SH - p.d.pad_t + kh;
      const bool rok = kok & ((unsigned)ih < (unsigned)p.d.H);
      const int rowbase = (b * p.d.H + ih) * p.d.W;
#pragma unroll
      for (int i = 0; i < A_KPT; ++i) {
        const int iw = (ow + i) * p.d.SW - p.d.pad_l + kw;
        const bool ok = rok & ((unsigned)iw < (unsigned)p.d.W);
        areg[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? (unsigned)((rowbase + iw) * p.d.Cin + ci0 + a_g * 8) * 2u : HOOB, 0, 0);
      }
    }
    {
      const int k = kbeg + t * HBK + b_kl * B_KPT;          // dy pixels are the reduction index itself: consecutive in memory
      const bool ok = b_nok & (k < kend);
#pragma unroll
      for (int i = 0; i < B_KPT; ++i)
        breg[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, ok ? (unsigned)((k + i) * p.N + bn + b_g * 8) * 2u : HOOB, 0, 0);
    }
  };

  // regs[i] = 8 channels of pixel i  ->  per channel j a run of KPT consecutive k, written to LDS row (g*8 + j)
  auto store_one = [&](unsigned* dst_base, const u32x4* regs, int g, int kl, auto kpt_tag) __attribute__((always_inline)) {
    constexpr int KPT = decltype(kpt_tag)::value;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {                      // channel pair (2jp, 2jp+1) = dword jp of every load
      unsigned* row_lo = dst_base + (g * 8 + 2 * jp) * HROW + (kl * KPT) / 2;
      unsigned* row_hi = row_lo + HROW;
      if constexpr (KPT == 4) {
        *reinterpret_cast<uint2*>(row_lo) = make_uint2(perm_lo(regs[0][jp], regs[1][jp]), perm_lo(regs[2][jp], regs[3][jp]));
        *reinterpret_cast<uint2*>(row_hi) = make_uint2(perm_hi(regs[0][jp], regs[1][jp]), perm_hi(regs[2][jp], regs[3][jp]));
      } else {
        *row_lo = perm_lo(regs[0][jp], regs[1][jp]);
        *row_hi = perm_hi(regs[0][jp], regs[1][jp]);
      }
    }
  };
  auto store_tile = [&](int buf) __attribute__((always_inline)) {
    store_one(As + buf * S::A_DW, areg, a_g, a_kl, std::integral_constant<int, A_KPT>());
    store_one(Bs + buf * S::B_DW, breg, b_g, b_kl, std::integral_constant<int, B_KPT>());
  };

  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_tile(0);
  store_tile(0);
  load_tile(1);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const unsigned* as = As + (t & 1) * S::A_DW;
    const unsigned* bs = Bs + (t & 1) * S::B_DW;
    bf16x8 fa[WMT][4], fb[WNT][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < WMT; ++i)
        fa[i][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&as[(wm * 32 * WMT + i * 32 + l31) * HROW + s * 8 + lh * 4]));
#pragma unroll
      for (int i = 0; i < WNT; ++i)
        fb[i][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&bs[(wn * 32 * WNT + i * 32 + l31) * HROW + s * 8 + lh * 4]));
    }
    store_tile((t + 1) & 1);
    load_tile(t + 2);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int n = 0; n < WNT; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][s], fb[n][s], acc[i][n], 0, 0, 0);
    __syncthreads();
  }

  float* out = p.c + (p.splitk > 1 ? (size_t)split * p.out_elems : 0);
  const bool fused = (p.splitk == 1);
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = bm + wm * 32 * WMT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int n = bn + wn * 32 * WNT + j * 32 + l31;
        if (m < p.M && n < p.N) {
          float v = acc[i][j][e];
          if (fused && p.accumulate) v += out[(size_t)m * p.N + n];
          out[(size_t)m * p.N + n] = v;
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The filter gradient with NO register transpose: operand tiles by LDS DMA in their memory order ([k = pixel][channel], channel
// contiguous), fragments by ds_read_b64_tr_b16 — gfx950's transposing LDS read.
// Why: a timing-only ablation of igemm_h_filter_kernel (profiles/r03_bf16_gemm_ablation.txt) showed the transposing staging pass
// (16 v_perm + 16 ds_write_b64 per thread and K-tile) costing 42-55 % of the kernel (4x4x1152->1024 at B = 192: 120 -> 55 us
// without it).
// ds_read_b64_tr_b16 (semantics measured with tools/probe/tr_probe.hip): inside each 16-lane group, lane j passes the address
// of a 4-element chunk C_j; rows R_r = C_4r ++ C_4r+1 ++ C_4r+2 ++ C_4r+3 (r = 0..3) form a 4 x 16 matrix, and lane i receives
// COLUMN i: (R_0[i], R_1[i], R_2[i], R_3[i]).  With chunk C_(4r+q) = tile[k0 + r][m0 + 4q .. 4q+3] lane i therefore gets
// tile[k0 .. k0+3][m0 + i]: four consecutive k of one channel — half an MFMA operand.  Lane groups 0/1 take m0 = 0 / 16 and
// k-half 0, groups 2/3 the same channels at k-half 1: exactly v_mfma_f32_32x32x16_bf16's A/B layout (row = lane & 31, k = 8 *
// (lane >> 5) + 0..7) after two reads (k0 and k0 + 4).
// LDS image: [64 k][128 channels] bf16 = 256-byte rows, unpadded (the DMA writes lane-linear); the sixteen 16-byte slots of a
// row are XOR-swizzled with (k & 3) << 2: the four rows a lane group reads at once would otherwise sit on the same banks
// (a 256-byte row is exactly one bank row); with the XOR the 32 lanes of a half-wave touch 32 distinct 8-byte bank pairs.
// Tile 128 x 128 only (the shape that carries the filter-gradient time); other shapes keep igemm_h_filter_kernel.
// ------------------------------------------------------------------------------------------------------------------
typedef short s16x4h __attribute__((ext_vector_type(4)));
typedef short s16x8h __attribute__((ext_vector_type(8)));

template <int PIPE>
__device__ __forceinline__ void hft_body(const IgemmParams& p, const int bx, const int by) {
  constexpr int BM = 128, BN = 128;
  constexpr int TILE_B = HBK * 256;                       // bytes of one operand tile: 64 k-rows of 256 bytes
  extern __shared__ __attribute__((aligned(16))) unsigned smem_h[];
  char* As = reinterpret_cast<char*>(smem_h);             // 2 buffers
  char* Bs = As + 2 * TILE_B;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int tile_m = bx % p.tiles_m, tile_n = bx / p.tiles_m;
  const int bm = tile_m * BM, bn = tile_n * BN;
  const int split = by;
  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int ntiles = (kend - kbeg + HBK - 1) / HBK;

  const i32x4h wa = rsrc_words_h(p.a, p.a_bytes), wb = rsrc_words_h(p.b, p.b_bytes);

  // tile constants: the filter tap and first input channel of this M tile (Cin % 128 == 0: the tile lies inside one tap)
  const int tap = p.div_c.div(bm);
  const int ci0 = bm - tap * p.d.Cin;
  const int kh = p.div_kw.div(tap), kw = tap - kh * p.d.KW;

  // ---- loader: DMA instruction i of wave w covers k-rows 16 i + 4 w + (lane >> 4); lane -> LDS slot position (lane & 15), it
  // FETCHES channel slot sg = position ^ ((row & 3) << 2) (row & 3 == lane >> 4) ---------------------------------------------
  const int rloc = wave * 4 + (lane >> 4);               // + 16 i
  const int sg = (lane & 15) ^ ((lane >> 4) << 2);
  const bool b_nok = (bn + sg * 8) < p.N;
  const unsigned lds_a0 = (unsigned)(size_t)(lds_ptr_h)(As + wave_u * 1024);      // + i * 4096 + buf * TILE_B
  const unsigned lds_b0 = (unsigned)(size_t)(lds_ptr_h)(Bs + wave_u * 1024);

  auto dma_tile = [&](int t, int buf) __attribute__((always_inline)) {
    const int kt = kbeg + t * HBK + rloc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kt + 16 * i;
      const bool kok = k < kend;
      const int kk = kok ? k : 0;
      const int b = p.div_howo.div(kk);
      const int rem = kk - b * p.howo;
      const int oh = p.div_wo.div(rem);
      const int ow = rem - oh * p.d.Wo;
      const int ih = oh * p.d.SH - p.d.pad_t + kh, iw = ow * p.d.SW - p.d.pad_l + kw;
      const bool ok = kok & ((unsigned)ih < (unsigned)p.d.H) & ((unsigned)iw < (unsigned)p.d.W);
      dma16_h(wa, ok ? (unsigned)(((b * p.d.H + ih) * p.d.W + iw) * p.d.Cin + ci0 + sg * 8) * 2u : HOOB,
              lds_a0 + (unsigned)(buf * TILE_B + i * 4096));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kt + 16 * i;
      dma16_h(wb, (b_nok & (k < kend)) ? (unsigned)(k * p.N + bn + sg * 8) * 2u : HOOB, lds_b0 + (unsigned)(buf * TILE_B + i * 4096));
    }
  };

  // ---- fragment addresses: lane = (group g = lane >> 4: channel half gb = g & 1, k half = g >> 1; r = (lane & 15) >> 2; q = lane & 3)
  // reads the chunk at k-row 16 s + 8 (g >> 1) + r (+ 4), channel chunk c = blk * 8 + gb * 4 + q, stored at chunk c ^ (r << 3)
  const int g = lane >> 4, gb = g & 1, r4 = (lane & 15) >> 2, q = lane & 3;
  int fa_off[2], fb_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    fa_off[i] = (8 * (g >> 1) + r4) * 256 + ((((wm * 2 + i) * 8 + gb * 4 + q) ^ (r4 << 3)) << 3);
    fb_off[i] = (8 * (g >> 1) + r4) * 256 + ((((wn * 2 + i) * 8 + gb * 4 + q) ^ (r4 << 3)) << 3);
  }
  auto frag = [&](const char* tile, int off, int s) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) s16x4h* lp;
    const s16x4h lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(tile + off + s * 4096));
    const s16x4h hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(tile + off + s * 4096 + 1024));
    const s16x8h v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  dma_tile(0, 0);
  dma_tile(1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __syncthreads();
  if constexpr (PIPE) {
    // the software-pipelined loop of igemm_hd_kernel<.., PIPE = 1>: fragments of step s+1 are read under the MFMAs of step s, one
    // barrier per K-tile between steps 2 and 3
    auto mma_step = [&](const bf16x8 (&fa)[2], const bf16x8 (&fb)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[n], acc[i][n], 0, 0, 0);
    };
    bf16x8 ca[2], cb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { ca[i] = frag(As, fa_off[i], 0); cb[i] = frag(Bs, fb_off[i], 0); }
    for (int t = 0; t < ntiles; ++t) {
      const char* as = As + (t & 1) * TILE_B;
      const char* bs = Bs + (t & 1) * TILE_B;
      const char* nas = As + ((t + 1) & 1) * TILE_B;
      const char* nbs = Bs + ((t + 1) & 1) * TILE_B;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        bf16x8 na[2], nb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { na[i] = frag(as, fa_off[i], s + 1); nb[i] = frag(bs, fb_off[i], s + 1); }
        mma_step(ca, cb);
#pragma unroll
        for (int i = 0; i < 2; ++i) { ca[i] = na[i]; cb[i] = nb[i]; }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t+1 have landed
      __syncthreads();
      dma_tile(t + 2, t & 1);
      {
        bf16x8 na[2], nb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { na[i] = frag(nas, fa_off[i], 0); nb[i] = frag(nbs, fb_off[i], 0); }
        mma_step(ca, cb);
#pragma unroll
        for (int i = 0; i < 2; ++i) { ca[i] = na[i]; cb[i] = nb[i]; }
      }
    }
  } else {
  for (int t = 0; t < ntiles; ++t) {
    const char* as = As + (t & 1) * TILE_B;
    const char* bs = Bs + (t & 1) * TILE_B;
    bf16x8 fa[2][4], fb[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i][s] = frag(as, fa_off[i], s);
#pragma unroll
      for (int i = 0; i < 2; ++i) fb[i][s] = frag(bs, fb_off[i], s);
    }
    __syncthreads();                                   // every wave holds its fragments of tile t: buf[t & 1] is free
    dma_tile(t + 2, t & 1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][s], fb[n][s], acc[i][n], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile t+1 has landed
    __syncthreads();
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (vec_epilogue_ok(p)) {       // dw[M][N] fp32, plain / accumulate / split-K slab: igemm_h_kernel's forward conventions without bias
    __syncthreads();
    PhaseInfo none;
    store_tile_h<MODE_FWD, 2, 2>(p, none, acc, reinterpret_cast<float*>(smem_h), bm, bn, split);
    return;
  }
  float* out = p.c + (p.splitk > 1 ? (size_t)split * p.out_elems : 0);
  const bool fused = (p.splitk == 1);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = bm + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = bn + wn * 64 + j * 32 + l31;
        if (m < p.M && n < p.N) {
          float v = acc[i][j][e];
          if (fused && p.accumulate) v += out[(size_t)m * p.N + n];
          out[(size_t)m * p.N + n] = v;
        }
      }
    }
}

template <int PIPE>
__global__ __launch_bounds__(256) void igemm_hft_kernel(IgemmParams p) {
  hft_body<PIPE>(p, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------------------------
// Two independent GEMMs of one layer's backward in ONE launch (round 4): the input gradient (or, behind a transposed conv, the
// forward-type conv of the incoming gradient) and the filter gradient read the same incoming gradient and write different tensors.
// At B = 64 each has <= 256 workgroups — one per CU, nothing to hide its barriers, DMA waits, prologue and epilogue — and two
// streams of a captured graph do NOT run them side by side on this stack (tools/probe/pair_overlap.py: pair on two streams =
// pair on one).  Here the first n1 workgroups run hd_body on `a`, the rest hft_body on `b`: the dispatcher puts one of each on
// a CU (64 KB of LDS each), and one launch ramp / drain is paid instead of two.  Same tiles, same arithmetic: bit-identical to
// the two launches.  `a` first: hd_body reads its stride-phase table at kernarg offset 0.
// ------------------------------------------------------------------------------------------------------------------
struct PairParams {
  IgemmParams a, b;
  int32_t n1, tiles1, splitk1;       // blocks of the first GEMM = tiles1 x splitk1 x phases; of the second = tiles_m tiles_n x splitk
  int32_t tiles2;
};

template <int MODE, int WMT, int WNT>
__global__ __launch_bounds__(256) void igemm_pair_kernel(PairParams pp) {
  const int b = blockIdx.x;
  if (b < pp.n1) {
    const int bx = b % pp.tiles1, r = b / pp.tiles1;
    hd_body<MODE, WMT, WNT, 0>(pp.a, bx, pp.tiles1, r % pp.splitk1, r / pp.splitk1);
  } else {
    const int b2 = b - pp.n1;
    hft_body<0>(pp.b, b2 % pp.tiles2, b2 / pp.tiles2);
  }
}

template <int WMT, int WNT>
static hipError_t launch_hf(const IgemmParams& p, hipStream_t stream) {
  using S = SmemH<WMT, WNT>;
  auto k = igemm_h_filter_kernel<WMT, WNT>;
  static bool attr_done = false;
  if (!attr_done && S::BYTES > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, S::BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(k, dim3(p.tiles_m * p.tiles_n, p.splitk), dim3(256), S::BYTES, stream, p);
  return hipGetLastError();
}

hipError_t igemm_h_filter_launch(const IgemmParams& p, int wmt, int wnt, hipStream_t stream) {
  if (wmt == 2 && wnt == 2 && tuning().bf16_dma && (p.d.Cin % 128) == 0) {
    constexpr int bytes = 4 * HBK * 256;               // 2 operands x 2 buffers of [64][128] bf16
    static bool attr_done = false;
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_hft_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_hft_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e != hipSuccess) return e;
      attr_done = true;
    }
    if (tuning().bf16_dma >= 2) hipLaunchKernelGGL(igemm_hft_kernel<1>, dim3(p.tiles_m * p.tiles_n, p.splitk), dim3(256), bytes, stream, p);
    else hipLaunchKernelGGL(igemm_hft_kernel<0>, dim3(p.tiles_m * p.tiles_n, p.splitk), dim3(256), bytes, stream, p);
    return hipGetLastError();
  }
  if (wmt == 2 && wnt == 2) return launch_hf<2, 2>(p, stream);
  if (wmt == 2 && wnt == 1) return launch_hf<2, 1>(p, stream);
  if (wmt == 1 && wnt == 2) return launch_hf<1, 2>(p, stream);
  if (wmt == 1 && wnt == 1) return launch_hf<1, 1>(p, stream);
  return hipErrorInvalidValue;
}

template <int MODE, int WMT, int WNT, int PIPE>
static hipError_t launch_hd1(const IgemmParams& p, dim3 grid, hipStream_t stream) {
  using S = SmemD<WMT, WNT, (PIPE == 6 ? 4 : (PIPE > 2 ? PIPE : 2))>;
  auto k = igemm_hd_kernel<MODE, WMT, WNT, PIPE>;
  static bool attr_done = false;   // benign race: idempotent
  if (!attr_done && S::BYTES > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, S::BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(k, grid, dim3(256), S::BYTES, stream, p);
  return hipGetLastError();
}
template <int MODE, int WMT, int WNT>
static hipError_t launch_hd(const IgemmParams& p, dim3 grid, hipStream_t stream) {
  const int v = tuning().bf16_dma;
  if constexpr (WMT == 2 && WNT == 2) {       // deeper LDS rings (experiment): 96 / 128 KB, one workgroup per CU
    if (v == 1 && tuning().bf16_pair_tiles && grid.x * grid.y * grid.z <= (unsigned)tuning().bf16_pair_tiles)
      return launch_hd1<MODE, WMT, WNT, 6>(p, grid, stream);       // single-round launch: two K-tiles per barrier pair (128 KB of LDS)
    if (v == 3) return launch_hd1<MODE, WMT, WNT, 3>(p, grid, stream);
    if (v == 4) return launch_hd1<MODE, WMT, WNT, 4>(p, grid, stream);
  }
  return v >= 2 ? launch_hd1<MODE, WMT, WNT, 2>(p, grid, stream) : launch_hd1<MODE, WMT, WNT, 0>(p, grid, stream);
}

template <int MODE, int WMT, int WNT>
static hipError_t launch_h(const IgemmParams& p, dim3 grid, hipStream_t stream) {
  if (tuning().bf16_dma) return launch_hd<MODE, WMT, WNT>(p, grid, stream);
  using S = SmemH<WMT, WNT>;
  auto k = igemm_h_kernel<MODE, WMT, WNT>;
  static bool attr_done = false;   // benign race: idempotent
  if (!attr_done && S::BYTES > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, S::BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(k, grid, dim3(256), S::BYTES, stream, p);
  return hipGetLastError();
}

bool igemm_pair_fusable(const IgemmParams& pb, int wmt_b, int wnt_b) {       // the filter gradient must be igemm_hft_kernel's case
  return tuning().bf16_dma == 1 && wmt_b == 2 && wnt_b == 2 && (pb.d.Cin % 128) == 0;
}

template <int MODE, int WMT, int WNT>
static hipError_t launch_pair(const PairParams& pp, int nblk, hipStream_t stream) {
  constexpr int bytes = 4 * HBK * 256;               // igemm_hft_kernel's 64 KB (>= every SmemD<WMT, WNT>)
  static_assert(SmemD<WMT, WNT>::BYTES <= bytes, "pair LDS");
  auto k = igemm_pair_kernel<MODE, WMT, WNT>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(k, dim3(nblk), dim3(256), bytes, stream, pp);
  return hipGetLastError();
}

// first GEMM: mode FWD / BWD_DATA on tile (wmt, wnt); second: the filter gradient on igemm_hft_kernel's 128x128 tile
hipError_t igemm_pair_launch(int mode, const IgemmParams& pa, int wmt, int wnt, const IgemmParams& pb, hipStream_t stream) {
  PairParams pp;
  pp.a = pa; pp.b = pb;
  pp.tiles1 = pa.tiles_m * pa.tiles_n; pp.splitk1 = pa.splitk;
  pp.n1 = pp.tiles1 * pa.splitk * (mode == MODE_BWD_DATA ? pa.nphase : 1);
  pp.tiles2 = pb.tiles_m * pb.tiles_n;
  const int nblk = pp.n1 + pp.tiles2 * pb.splitk;
#define T2I_P(M_, a, b) if (mode == M_ && wmt == a && wnt == b) return launch_pair<M_, a, b>(pp, nblk, stream);
  T2I_P(MODE_FWD, 2, 2) T2I_P(MODE_FWD, 2, 1) T2I_P(MODE_FWD, 1, 2) T2I_P(MODE_FWD, 1, 1)
  T2I_P(MODE_BWD_DATA, 2, 2) T2I_P(MODE_BWD_DATA, 2, 1) T2I_P(MODE_BWD_DATA, 1, 2) T2I_P(MODE_BWD_DATA, 1, 1)
#undef T2I_P
  return hipErrorInvalidValue;
}

template <int MODE, int PIPE, int WMT = 2>
static hipError_t launch_hd8(const IgemmParams& p, dim3 grid, hipStream_t stream) {
  using S = SmemD<2 * WMT, 2, (PIPE == 8 ? 3 : (PIPE == 6 ? 4 : (PIPE > 2 ? PIPE : 2)))>;
  constexpr int EPI = 8 * 32 * (32 * 2 + 4) * 4;       // store_tile_h: one 32 x 68 float patch per wave
  constexpr int bytes = S::BYTES > EPI ? S::BYTES : EPI;
  auto k = igemm_hd8_kernel<MODE, WMT, PIPE>;
  static bool attr_done = false;   // benign race: idempotent
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(k, grid, dim3(512), bytes, stream, p);
  return hipGetLastError();
}

// tiles: (wmt, wnt) in {(2,2), (2,1), (1,2), (1,1)}; (4,2) = the 8-wave 256 x 128 tile (LDS-DMA kernel only)
hipError_t igemm_h_launch(int mode, const IgemmParams& p, int wmt, int wnt, hipStream_t stream) {
  dim3 grid(p.tiles_m * p.tiles_n, p.splitk, mode == MODE_BWD_DATA ? p.nphase : 1);
  if (wmt == 4 && wnt == 2) {
    if (!tuning().bf16_dma) return hipErrorInvalidValue;
    if (tuning().bf16_dma == 8) {          // ping-pong loop (hd_body, PIPE = 8)
      if (mode == MODE_FWD) return launch_hd8<MODE_FWD, 8>(p, grid, stream);
      if (mode == MODE_BWD_DATA) return launch_hd8<MODE_BWD_DATA, 8>(p, grid, stream);
    }
    const bool pipe = tuning().bf16_dma >= 2;
    if (mode == MODE_FWD) return pipe ? launch_hd8<MODE_FWD, 2>(p, grid, stream) : launch_hd8<MODE_FWD, 0>(p, grid, stream);
    if (mode == MODE_BWD_DATA) return pipe ? launch_hd8<MODE_BWD_DATA, 2>(p, grid, stream) : launch_hd8<MODE_BWD_DATA, 0>(p, grid, stream);
    return hipErrorInvalidValue;
  }
  if (wmt == 2 && wnt == 2 && tuning().bf16_dma == 1 && tuning().bf16_waves == 8) {       // the 128 x 128 tile on 8 waves
    if (tuning().bf16_pair_tiles && grid.x * grid.y * grid.z <= (unsigned)tuning().bf16_pair_tiles) {     // single-round launch: paired K-tiles
      if (mode == MODE_FWD) return launch_hd8<MODE_FWD, 6, 1>(p, grid, stream);
      if (mode == MODE_BWD_DATA) return launch_hd8<MODE_BWD_DATA, 6, 1>(p, grid, stream);
    }
    if (mode == MODE_FWD) return launch_hd8<MODE_FWD, 0, 1>(p, grid, stream);
    if (mode == MODE_BWD_DATA) return launch_hd8<MODE_BWD_DATA, 0, 1>(p, grid, stream);
  }
#define T2I_H(M_, a, b) if (mode == M_ && wmt == a && wnt == b) return launch_h<M_, a, b>(p, grid, stream);
  T2I_H(MODE_FWD, 2, 2) T2I_H(MODE_FWD, 2, 1) T2I_H(MODE_FWD, 1, 2) T2I_H(MODE_FWD, 1, 1)
  T2I_H(MODE_BWD_DATA, 2, 2) T2I_H(MODE_BWD_DATA, 2, 1) T2I_H(MODE_BWD_DATA, 1, 2) T2I_H(MODE_BWD_DATA, 1, 1)
#undef T2I_H
  return hipErrorInvalidValue;
}

}  // namespace t2i
