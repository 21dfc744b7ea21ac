// t2i_thin.hip — direct (non-GEMM) kernels for the 3-channel ends of the networks (gfx950 only).
//
// The image-side layers have 3 channels on one side (critic conv 3->128, generator deconv 128->3 and conv 3->3).  As
// implicit GEMMs their N (or both M and N) is 3: a 64-wide MFMA tile would spend >95% of the matrix pipe on padding,
// while the real cost is streaming the 128-channel tensor once (33.5 MB at B=64).  They are HBM/LDS-bound, so they
// get VALU kernels (SURVEY.md §7 "thin layers"):
//   thin_deconv_k4s2   dx[B,H,W,Ci<=4] = act(conv^T(dy[B,H/2,W/2,Co], w[4,4,Ci,Co]) + bias)   (k4 s2 SAME)
//                      = generator out_deconv forward, and the critic's first-layer input gradient.
//                      One workgroup = a 16x16 tile of dx of one image; its 10x10xCo patch of dy is staged in LDS once
//                      (pixel stride Co+4 floats: conflict-free ds_read_b128 across pixels); wave w computes the 64
//                      pixels of stride phase w, so its 2x2 filter taps are wave-uniform and come through scalar loads.
//   tiny_conv          y = act(conv(x, w) + b) and its input gradient for Cin<=4, Cout<=4 (the 3->3 output conv):
//                      one thread per pixel, filter through scalar loads, input through L1/L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "t2i_internal.h"

namespace t2i {

// ------------------------------------------------------------------------------------------------------------------
template <int CI, bool DYH = false>      // DYH: dy is a bf16 tensor (bf16 storage); widened exactly while it is staged into LDS
__global__ __launch_bounds__(256) void thin_deconv_k4s2_kernel(const void* __restrict__ dyv, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ dx,
                                                               int H, int W, int Co, int CH, int act, float alpha) {
  // CH = channels staged per pass (Co or Co / 2): with half the channels in LDS at a time a workgroup needs half the LDS, twice
  // as many are resident per CU, and one workgroup's staging overlaps another's arithmetic (the kernel is bound by staging)
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [10][10][CH+4] dy patch, then [16 taps][CI][CH] filter
  const int Ho = H >> 1, Wo = W >> 1;
  const int PS = CH + 4;                       // pixel stride in LDS
  const int b = blockIdx.z;
  const int ih0 = blockIdx.y * 16, iw0 = blockIdx.x * 16;
  const int oh0 = (ih0 >> 1) - 1, ow0 = (iw0 >> 1) - 1;
  const int c4n = CH >> 2;                     // float4 per pixel and pass
  float* wlds = tile + 100 * PS;
  // ---- wave = stride phase; lane = one of its 8x8 pixels --------------------------------------------------------------
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ph = wv >> 1, pw = wv & 1;
  const int lane = threadIdx.x & 63;
  const int py = lane >> 3, px = lane & 7;
  const int kh0 = (ph + 1) & 1, kw0 = (pw + 1) & 1;   // first tap of this phase (pad 1): kh = kh0 + 2*jh
  float acc[CI];
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) acc[ci] = 0.f;
  for (int cbase = 0; cbase < Co; cbase += CH) {
    if (cbase) __syncthreads();                // everyone is done reading the previous pass
    // ---- stage the dy patch (zero outside the image) and the filter slice --------------------------------------------
    if (CH == 64) {
      // round 4: every global load of the pass is issued BEFORE the first LDS store (7 + 3 per thread at 64 channels).  The generic loop
      // below stores each value as it arrives: ~7 dependent HBM round trips per pass, and staging is what this kernel waits for
      // (a blocked inner loop with 3x fewer LDS reads did not move it, see the end of this file).
      constexpr int C4N = 16, XN = (100 * C4N + 255) / 256, WN = (16 * CI * C4N + 255) / 256;
      float4 xv[XN], wv[WN];
#pragma unroll
      for (int u = 0; u < XN; ++u) {
        const int i = threadIdx.x + 256 * u;
        const int pix = i >> 4, c4 = i & 15;
        const int r = pix / 10, c = pix - r * 10;
        const int oh = oh0 + r, ow = ow0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < 100 * C4N && (unsigned)oh < (unsigned)Ho && (unsigned)ow < (unsigned)Wo) {
          const size_t e = ((size_t)(b * Ho + oh) * Wo + ow) * Co + cbase + c4 * 4;
          if (DYH) {
            const uint2 q = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(dyv) + e);
            v = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xFFFF0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xFFFF0000u));
          } else {
            v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dyv) + e);
          }
        }
        xv[u] = v;
      }
#pragma unroll
      for (int u = 0; u < WN; ++u) {
        const int i = threadIdx.x + 256 * u;
        const int row = i >> 4, c4 = i & 15;
        wv[u] = i < 16 * CI * C4N ? *reinterpret_cast<const float4*>(w + (size_t)row * Co + cbase + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < XN; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i < 100 * C4N) *reinterpret_cast<float4*>(&tile[(i >> 4) * PS + (i & 15) * 4]) = xv[u];
      }
#pragma unroll
      for (int u = 0; u < WN; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i < 16 * CI * C4N) reinterpret_cast<float4*>(wlds)[i] = wv[u];
      }
    } else {
    for (int i = threadIdx.x; i < 100 * c4n; i += 256) {
      const int pix = i / c4n, c4 = i - pix * c4n;
      const int r = pix / 10, c = pix - r * 10;
      const int oh = oh0 + r, ow = ow0 + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)oh < (unsigned)Ho && (unsigned)ow < (unsigned)Wo) {
        const size_t e = ((size_t)(b * Ho + oh) * Wo + ow) * Co + cbase + c4 * 4;
        if (DYH) {
          const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(dyv) + e);
          v = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
        } else {
          v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dyv) + e);
        }
      }
      *reinterpret_cast<float4*>(&tile[pix * PS + c4 * 4]) = v;
    }
    for (int i = threadIdx.x; i < 16 * CI * c4n; i += 256) {          // [tap][ci][CH] <- w[tap][ci][cbase .. cbase+CH)
      const int row = i / c4n, c4 = i - row * c4n;
      reinterpret_cast<float4*>(wlds)[i] = *reinterpret_cast<const float4*>(w + (size_t)row * Co + cbase + c4 * 4);
    }
    }
    __syncthreads();
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
#pragma unroll
      for (int jw = 0; jw < 2; ++jw) {
        const int r = py + 1 + ph - jh, c = px + 1 + pw - jw;           // patch coordinates of the contributing dy pixel
        const float* src = &tile[(r * 10 + c) * PS];
        // this tap's filter rows in LDS: every lane of the wave reads the SAME address (broadcast, conflict-free).
        // (v1 read them with scalar loads straight from global: 384 dependent s_load per wave, ~35 us of the kernel's 65.)
        const float* wt = wlds + (((kh0 + 2 * jh) * 4 + (kw0 + 2 * jw)) * CI) * CH;
#pragma unroll 4
        for (int c4 = 0; c4 < c4n; ++c4) {
          const float4 d = *reinterpret_cast<const float4*>(src + c4 * 4);
#pragma unroll
          for (int ci = 0; ci < CI; ++ci) {
            const float4 f = *reinterpret_cast<const float4*>(wt + ci * CH + c4 * 4);
            acc[ci] = fmaf(d.x, f.x, fmaf(d.y, f.y, fmaf(d.z, f.z, fmaf(d.w, f.w, acc[ci]))));
          }
        }
      }
    }
  }
  const int ih = ih0 + 2 * py + ph, iw = iw0 + 2 * px + pw;
  float* o = dx + ((size_t)(b * H + ih) * W + iw) * CI;
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) o[ci] = apply_act(acc[ci] + (bias ? bias[ci] : 0.f), act, alpha);
}

// Measured and dropped (round 4): a 2x2 block of same-phase output pixels per lane (32x32 tiles, the 18x18 patch staged 32 or 64
// channels at a time: 0.11 instead of 0.33 LDS reads per FMA).  35 -> 47-48 us at B = 64, 88 -> 98-135 us at B = 192: the kernel is
// bound by STAGING the dy patch (global -> registers -> LDS with per-element index arithmetic, one workgroup per CU, passes separated
// by barriers), not by the LDS reads of the inner loop; four times fewer, four times larger workgroups made that worse.
bool thin_deconv_eligible(const t2i_conv_desc& d) {
  return d.KH == 4 && d.KW == 4 && d.SH == 2 && d.SW == 2 && d.pad_t == 1 && d.pad_l == 1 && d.Cin >= 1 && d.Cin <= 4 &&
         (d.Cout % 4) == 0 && d.Cout >= 16 && d.Cout <= 512 && (d.H % 16) == 0 && (d.W % 16) == 0 && d.Ho * 2 == d.H &&
         d.Wo * 2 == d.W;
}

hipError_t thin_deconv_launch(const t2i_conv_desc& d, const void* dy, const float* w, const float* bias, float* dx,
                              int act, float alpha, hipStream_t stream, bool dy_bf16) {
  int parts = tuning().thin_parts;                  // passes over the channels (1, 2 or 4): Co / parts are staged at a time
  if (parts < 1) parts = 1;
  while (parts > 1 && ((d.Cout % (4 * parts)) != 0 || d.Cout / parts < 32)) parts >>= 1;
  const int CH = d.Cout / parts;
  const size_t lds = ((size_t)100 * (CH + 4) + (size_t)16 * d.Cin * CH) * sizeof(float);
  dim3 grid(d.W / 16, d.H / 16, d.B);
#define T2I_THIN(CI)                                                                                              \
  case CI: {                                                                                                      \
    auto k = dy_bf16 ? thin_deconv_k4s2_kernel<CI, true> : thin_deconv_k4s2_kernel<CI, false>;                    \
    if (lds > 48 * 1024) {                                                                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return e;                                                                              \
    }                                                                                                             \
    hipLaunchKernelGGL(k, grid, dim3(256), lds, stream, dy, w, bias, dx, d.H, d.W, d.Cout, CH, act, alpha);       \
    break;                                                                                                        \
  }
  switch (d.Cin) {
    T2I_THIN(1) T2I_THIN(2) T2I_THIN(3) T2I_THIN(4)
    default: return hipErrorInvalidValue;
  }
#undef T2I_THIN
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// tiny conv: Cin <= 4 and Cout <= 4.  BWD = false: y[b,oh,ow,:] = act(b + sum x[b,oh*s-p+kh,ow*s-p+kw,:] w[kh,kw,:,:])
//                                     BWD = true (stride 1): dx[b,ih,iw,:] = sum dy[b,ih+p-kh,iw+p-kw,:] w[kh,kw,:,:]^T
// ------------------------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void tiny_conv_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        t2i_conv_desc d, int act, float alpha) {
  // FWD: in = x [B,H,W,Cin], out = y [B,Ho,Wo,Cout].   BWD: in = dy [B,Ho,Wo,Cout], out = dx [B,H,W,Cin]
  const int OH = BWD ? d.H : d.Ho, OW = BWD ? d.W : d.Wo, OC = BWD ? d.Cin : d.Cout;
  const int IH = BWD ? d.Ho : d.H, IW = BWD ? d.Wo : d.W, IC = BWD ? d.Cout : d.Cin;
  const size_t npix = (size_t)d.B * OH * OW;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
    const int ow = (int)(p % OW);
    const size_t t = p / OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < d.KH; ++kh) {
      const int ih = BWD ? oh + d.pad_t - kh : oh * d.SH - d.pad_t + kh;
      if ((unsigned)ih >= (unsigned)IH) continue;
      for (int kw = 0; kw < d.KW; ++kw) {
        const int iw = BWD ? ow + d.pad_l - kw : ow * d.SW - d.pad_l + kw;
        if ((unsigned)iw >= (unsigned)IW) continue;
        const float* src = in + ((size_t)(b * IH + ih) * IW + iw) * IC;
        const float* wt = w + (size_t)(kh * d.KW + kw) * d.Cin * d.Cout;    // [Cin][Cout], uniform
        for (int ic = 0; ic < IC; ++ic) {
          const float v = src[ic];
#pragma unroll
          for (int oc = 0; oc < 4; ++oc)
            if (oc < OC) acc[oc] = fmaf(v, BWD ? wt[oc * d.Cout + ic] : wt[ic * d.Cout + oc], acc[oc]);
        }
      }
    }
    float* o = out + p * OC;
#pragma unroll
    for (int oc = 0; oc < 4; ++oc)
      if (oc < OC) o[oc] = apply_act(acc[oc] + (bias ? bias[oc] : 0.f), act, alpha);
  }
}

bool tiny_conv_eligible(const t2i_conv_desc& d, bool bwd) {
  if (d.Cin > 4 || d.Cout > 4) return false;
  if (bwd && (d.SH != 1 || d.SW != 1)) return false;
  return true;
}

hipError_t tiny_conv_launch(const t2i_conv_desc& d, bool bwd, const float* in, const float* w, const float* bias,
                            float* out, int act, float alpha, hipStream_t stream) {
  const size_t npix = (size_t)d.B * (bwd ? d.H * d.W : d.Ho * d.Wo);
  size_t blocks = (npix + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (bwd)
    hipLaunchKernelGGL(tiny_conv_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, in, w, bias, out, d, act, alpha);
  else
    hipLaunchKernelGGL(tiny_conv_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, in, w, bias, out, d, act, alpha);
  return hipGetLastError();
}

}  // namespace t2i

// ------------------------------------------------------------------------------------------------------------------
// "head" conv: the whole HxWxCin map of one image is reduced to 1x1xCout with Cout <= 4 (the critic's logit layer, k4 s4
// VALID on a 4x4x1024 map: one 16384-long dot product per sample).  As a GEMM it is M = B, N = 1: pure latency.
//   head_fwd   y[b,co]  = act(bias[co] + sum_j x[b,j] * w[j,co])          one workgroup per sample, wave64 shuffle reduce
//   head_bwd_data    dx[b,j]  = sum_co dy[b,co] * w[j,co]                 elementwise
//   head_bwd_filter  dw[j,co] (+)= sum_b x[b,j] * dy[b,co]                one thread per j, coalesced over j
// ------------------------------------------------------------------------------------------------------------------
namespace t2i {

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, int K,
                                                       int Co, int act, float alpha) {
  __shared__ float red[4][4];
  const float* row = x + (size_t)blockIdx.x * K;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (Co == 1 && (K & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
    // the critic's logit layer: one 16-byte load of x and of w per lane and step, several in flight (the scalar loop below
    // took 26 us for 192 samples: 64 dependent 4-byte round trips per lane)
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int j = threadIdx.x; j < (K >> 2); j += 256) {
      const float4 v = reinterpret_cast<const float4*>(row)[j];
      const float4 q = reinterpret_cast<const float4*>(w)[j];
      a4.x = fmaf(v.x, q.x, a4.x); a4.y = fmaf(v.y, q.y, a4.y); a4.z = fmaf(v.z, q.z, a4.z); a4.w = fmaf(v.w, q.w, a4.w);
    }
    acc[0] = (a4.x + a4.y) + (a4.z + a4.w);
  } else
  for (int j = threadIdx.x; j < K; j += 256) {
    const float v = row[j];
#pragma unroll
    for (int co = 0; co < 4; ++co)
      if (co < Co) acc[co] = fmaf(v, w[(size_t)j * Co + co], acc[co]);
  }
#pragma unroll
  for (int co = 0; co < 4; ++co) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[co] += __shfl_xor(acc[co], o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][co] = acc[co];
  }
  __syncthreads();
  if (threadIdx.x < Co) {
    const int co = threadIdx.x;
    const float s = (red[0][co] + red[1][co]) + (red[2][co] + red[3][co]) + (bias ? bias[co] : 0.f);
    y[(size_t)blockIdx.x * Co + co] = apply_act(s, act, alpha);
  }
}

__global__ __launch_bounds__(256) void head_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, int B, int K, int Co) {
  const size_t n = (size_t)B * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / K;
    const int j = (int)(i - b * K);
    float s = 0.f;
    for (int co = 0; co < Co; ++co) s = fmaf(dy[b * Co + co], w[(size_t)j * Co + co], s);
    dx[i] = s;
  }
}

__global__ __launch_bounds__(256) void head_bwd_filter_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dw, int B, int K, int Co, int accumulate) {
  // 64 consecutive j per workgroup (coalesced), the batch split over 4 row lanes, joined through LDS in a fixed order
  __shared__ float red[4][64][4];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (j < K) {
    for (int b = ty; b < B; b += 4) {
      const float v = x[(size_t)b * K + j];
#pragma unroll
      for (int co = 0; co < 4; ++co)
        if (co < Co) acc[co] = fmaf(v, dy[b * Co + co], acc[co]);
    }
  }
#pragma unroll
  for (int co = 0; co < 4; ++co) red[ty][tx][co] = acc[co];
  __syncthreads();
  if (ty == 0 && j < K) {
#pragma unroll
    for (int co = 0; co < 4; ++co)
      if (co < Co) {
        const float s = (red[0][tx][co] + red[1][tx][co]) + (red[2][tx][co] + red[3][tx][co]);
        float* o = dw + (size_t)j * Co + co;
        *o = accumulate ? *o + s : s;
      }
  }
}

bool head_conv_eligible(const t2i_conv_desc& d) {
  return d.Ho == 1 && d.Wo == 1 && d.KH == d.H && d.KW == d.W && d.pad_t == 0 && d.pad_l == 0 && d.Cout <= 4;
}

hipError_t head_fwd_launch(const t2i_conv_desc& d, const float* x, const float* w, const float* bias, float* y, int act,
                           float alpha, hipStream_t stream) {
  hipLaunchKernelGGL(head_fwd_kernel, dim3(d.B), dim3(256), 0, stream, x, w, bias, y, d.H * d.W * d.Cin, d.Cout, act, alpha);
  return hipGetLastError();
}

hipError_t head_bwd_data_launch(const t2i_conv_desc& d, const float* dy, const float* w, float* dx, hipStream_t stream) {
  const int K = d.H * d.W * d.Cin;
  size_t blocks = ((size_t)d.B * K + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(head_bwd_data_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dy, w, dx, d.B, K, d.Cout);
  return hipGetLastError();
}

hipError_t head_bwd_filter_launch(const t2i_conv_desc& d, const float* x, const float* dy, float* dw, int accumulate,
                                  hipStream_t stream) {
  const int K = d.H * d.W * d.Cin;
  hipLaunchKernelGGL(head_bwd_filter_kernel, dim3((K + 63) / 64), dim3(256), 0, stream, x, dy, dw, d.B, K, d.Cout, accumulate);
  return hipGetLastError();
}

}  // namespace t2i

// ------------------------------------------------------------------------------------------------------------------
// First critic layer (and the input gradient of the generator's last deconv): y = act(conv_k4s2(x[B,H,W,CI<=4], w) + b)
// with Cout a multiple of 32.  As an implicit GEMM its K is 16*CI = 48: the general kernel spends its time in prologue
// and in 12 scalar gathers per float4 (28 us at B = 64 for 36.7 MB of traffic).  Here one workgroup owns 4 output rows of
// one image (4 x 32 pixels = a 128-row GEMM tile) and there is no K loop at all:
//   * the 10 input rows it needs (+ one zero pixel left and right) go to LDS with coalesced loads, the whole filter
//     [16*CI][Cout] next to them;
//   * wave w = output row, lane = output column: the MFMA A operand of step k = (kh, kw, ci) is one ds_read_b32 at
//     row 2w + kh, pixel 2 ow + kw (stride 6 floats across lanes: conflict free), the B operand a row of the filter;
//   * 8*CI MFMA steps (v_mfma_f32_32x32x2_f32) per 32 output channels, bias + activation in registers, 128-byte stores.
// The 3-channel ends of the nets are HBM-bound: this moves them from ~1.3 TB/s towards the stream rate.
// ------------------------------------------------------------------------------------------------------------------
namespace t2i {

typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int CI, int NB>      // NB = Cout / 32 accumulator blocks per wave (4 for the critic's 128 channels)
__global__ __launch_bounds__(256) void stem_k4s2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int H, int W,
                                                            int act, float alpha, __bf16* __restrict__ yh) {
  constexpr int K = 16 * CI, Co = 32 * NB;
  extern __shared__ __attribute__((aligned(16))) float lds_stem[];
  const int Wo = W >> 1, Ho = H >> 1;
  constexpr int cols = 66 * CI;                // 32 output columns need 66 input pixels (one of padding left and right)
  float* xs = lds_stem;                        // [10][cols]
  float* ws = lds_stem + 10 * cols;            // [K][Co]
  const int b = blockIdx.z, oh0 = blockIdx.y * 4, ow0 = blockIdx.x * 32;
  // ---- stage input rows 2*oh0-1 .. 2*oh0+8, columns 2*ow0-1 .. 2*ow0+64, zero outside the image ---------------------------
  // (all loads are issued before the first LDS store: a load -> store loop serialises ~14 global round trips per workgroup)
  constexpr int XN = (10 * cols + 255) / 256, WN = (K * Co / 4 + 255) / 256;
  float xv[XN];
  float4 wv[WN];
#pragma unroll
  for (int u = 0; u < XN; ++u) {
    const int i = threadIdx.x + 256 * u;
    const int r = i / cols, j = i - r * cols;
    const int px = j / CI, ci = j - px * CI;
    const int ih = 2 * oh0 - 1 + r, iw = 2 * ow0 - 1 + px;
    const bool ok = i < 10 * cols && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    xv[u] = ok ? x[((size_t)(b * H + ih) * W + iw) * CI + ci] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < WN; ++u) {
    const int i = threadIdx.x + 256 * u;
    wv[u] = i < K * Co / 4 ? reinterpret_cast<const float4*>(w)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < XN; ++u) {
    const int i = threadIdx.x + 256 * u;
    if (i < 10 * cols) xs[i] = xv[u];
  }
#pragma unroll
  for (int u = 0; u < WN; ++u) {
    const int i = threadIdx.x + 256 * u;
    if (i < K * Co / 4) reinterpret_cast<float4*>(ws)[i] = wv[u];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16_t acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  // k = (kh*4 + kw)*CI + ci; MFMA step s consumes k = 2s (lanes 0-31) and 2s+1 (lanes 32-63)
#pragma unroll
  for (int s2 = 0; s2 < K / 2; ++s2) {
    const int k = 2 * s2 + lh;
    const int tap = k / CI, ci = k - tap * CI;
    const int kh = tap >> 2, kw = tap & 3;
    const float a = xs[(2 * wave + kh) * cols + (2 * l31 + kw) * CI + ci];
#pragma unroll
    for (int j = 0; j < NB; ++j)
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ws[k * Co + j * 32 + l31], acc[j], 0, 0, 0);
  }
  // ---- epilogue: C/D layout col = lane&31 (channel), row = (e&3) + 8*(e>>2) + 4*lh (output column) -------------------------
  // Round 4: through LDS.  A lane holds ONE channel of 16 pixels per block, so the direct form issued 64 stores of 2-4 bytes per
  // thread — store-issue-bound like the GEMM epilogues (profiles/r04_bf16_gemm_fixed_cost.txt).  The staged input / filter are dead
  // after the barrier; each wave passes its 32 pixels x 128 channels through a private patch, 64 channels at a time, and a lane
  // then owns 8 consecutive channels of a pixel: one 16-byte store for the bf16 tensor, two for the fp32 one.  Same arithmetic.
  const int oh = oh0 + wave;
  if constexpr (NB % 2 == 0) {                  // 128 channels (wgancls) or 64 (gancls / StackGAN, round 5): 64 channels per pass
    constexpr int SROW = 68;                    // 64 channels + 4 floats of padding
    __syncthreads();                            // xs / ws are no longer read
    float* st = lds_stem + wave * (32 * SROW);
#pragma unroll
    for (int half = 0; half < NB / 2; ++half) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 16; ++e) st[((e & 3) + 8 * (e >> 2) + 4 * lh) * SROW + jj * 32 + l31] = acc[half * 2 + jj][e];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * 64 + lane, row = idx >> 3, c8 = idx & 7;
        const float4 v0 = *reinterpret_cast<const float4*>(&st[row * SROW + c8 * 8]);
        const float4 v1 = *reinterpret_cast<const float4*>(&st[row * SROW + c8 * 8 + 4]);
        const int ow = ow0 + row, ch = half * 64 + c8 * 8;
        if (oh < Ho && ow < Wo) {
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          if (bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(bias + ch), b1 = *reinterpret_cast<const float4*>(bias + ch + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += 0.f;          // the direct form adds bv = 0: keeps -0 -> +0 identical
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = apply_act(v[k], act, alpha);
          const size_t o = ((size_t)(b * Ho + oh) * Wo + ow) * Co + ch;
          if (y) {                                // y == NULL: bf16 storage, the bf16 tensor alone is written
            *reinterpret_cast<float4*>(y + o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(y + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
          }
          if (yh) {                               // bf16 twin for the conv that reads this tensor next, or THE tensor
            typedef float f2_t __attribute__((ext_vector_type(2)));
            typedef __bf16 h2_t __attribute__((ext_vector_type(2)));
            uint4 h;
            { f2_t t = {v[0], v[1]}; h.x = __builtin_bit_cast(unsigned, __builtin_convertvector(t, h2_t)); }
            { f2_t t = {v[2], v[3]}; h.y = __builtin_bit_cast(unsigned, __builtin_convertvector(t, h2_t)); }
            { f2_t t = {v[4], v[5]}; h.z = __builtin_bit_cast(unsigned, __builtin_convertvector(t, h2_t)); }
            { f2_t t = {v[6], v[7]}; h.w = __builtin_bit_cast(unsigned, __builtin_convertvector(t, h2_t)); }
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(yh) + o) = h;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  } else if (oh < Ho) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float bv = bias ? bias[j * 32 + l31] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ow = ow0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (ow < Wo) {
          const float v = apply_act(acc[j][e] + bv, act, alpha);
          const size_t o = ((size_t)(b * Ho + oh) * Wo + ow) * Co + j * 32 + l31;
          if (y) y[o] = v;                       // y == NULL: bf16 storage, the bf16 tensor alone is written
          if (yh) yh[o] = (__bf16)v;             // bf16 twin for the conv that reads this tensor next, or THE tensor
        }
      }
    }
  }
}

bool stem_fwd_eligible(const t2i_conv_desc& d) {
  return d.KH == 4 && d.KW == 4 && d.SH == 2 && d.SW == 2 && d.pad_t == 1 && d.pad_l == 1 && d.Cin == 3 && (d.Cout == 128 || d.Cout == 64) &&
         (d.H & 1) == 0 && (d.W & 1) == 0 && d.Ho * 2 == d.H && d.Wo * 2 == d.W;   // fp32 arithmetic in both math modes, like the other thin kernels
}

hipError_t stem_fwd_launch(const t2i_conv_desc& d, const float* x, const float* w, const float* bias, float* y, int act, float alpha,
                           hipStream_t stream, void* y_h) {
  size_t lds = ((size_t)10 * 66 * 3 + (size_t)48 * d.Cout) * sizeof(float);
  if (lds < (size_t)4 * 32 * 68 * sizeof(float)) lds = (size_t)4 * 32 * 68 * sizeof(float);      // the epilogue's staging patches
  auto k = d.Cout == 128 ? stem_k4s2_fwd_kernel<3, 4> : stem_k4s2_fwd_kernel<3, 2>;              // (the 3 -> 64 first layer of gancls / StackGAN: round 5)
  dim3 grid((d.Wo + 31) / 32, (d.Ho + 3) / 4, d.B);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, stream, x, w, bias, y, d.H, d.W, act, alpha, reinterpret_cast<__bf16*>(y_h));
  return hipGetLastError();
}

}  // namespace t2i

// Measured and dropped (round 1): direct VALU kernels for the Cin = 3 k4 s2 forward / filter gradient (critic layer 1).
// Three variants (LDS-broadcast patch, multi-row, SGPR patch via s_load) all ran 34-38 us at B = 64 against 31 us for the
// igemm path: 48 dependent scalar-fp32 FMAs per output sit at the non-packed VALU rate (~10 us) and the per-pixel
// overhead doubles it; the filter gradient came out equal to igemm (60 us).  See DESIGN.md section 4.5.
namespace t2i {

size_t col_reduce_ws(int64_t rows, int C);
hipError_t col_reduce_launch(const void*, const void*, const float*, int64_t, int, float*, float*, int, void*, hipStream_t, bool in_bf16 = false);

// ------------------------------------------------------------------------------------------------------------------
// tiny filter gradient (Cin, Cout <= 3, k <= 3: the generator's 3 -> 3 output conv): 81 sums over B*H*W pixels.
// ------------------------------------------------------------------------------------------------------------------
template <int CI, int CO>
__global__ __launch_bounds__(256) void tiny_bwdw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ part, t2i_conv_desc d) {
  __shared__ float red[4][9 * CI * CO];
  float acc[9][CI][CO];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
      for (int co = 0; co < CO; ++co) acc[t][ci][co] = 0.f;
  const size_t npix = (size_t)d.B * d.Ho * d.Wo;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
    const int ow = (int)(p % d.Wo);
    const size_t t = p / d.Wo;
    const int oh = (int)(t % d.Ho), b = (int)(t / d.Ho);
    float g[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) g[co] = dy[p * CO + co];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * d.SH - d.pad_t + kh;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * d.SW - d.pad_l + kw;
        if (kh < d.KH && kw < d.KW && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W) {
          const float* s = x + ((size_t)(b * d.H + ih) * d.W + iw) * CI;
#pragma unroll
          for (int ci = 0; ci < CI; ++ci) {
            const float v = s[ci];
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[kh * 3 + kw][ci][co] = fmaf(v, g[co], acc[kh * 3 + kw][ci][co]);
          }
        }
      }
    }
  }
  // wave64 shuffle reduction of each of the 9*CI*CO sums, then across the 4 waves through LDS
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        float v = acc[t][ci][co];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][(t * CI + ci) * CO + co] = v;
      }
  __syncthreads();
  if (threadIdx.x < 9 * CI * CO) {
    const int i = threadIdx.x;                 // (t, ci, co) with t = kh*3+kw over a 3x3 grid
    const int t = i / (CI * CO), r = i - t * (CI * CO);
    const int kh = t / 3, kw = t - kh * 3;
    if (kh < d.KH && kw < d.KW)
      part[(size_t)blockIdx.x * (d.KH * d.KW * CI * CO) + (kh * d.KW + kw) * (CI * CO) + r] =
          (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Filter gradient of the 3 -> 128 k4 s2 stem (the critic's first layer):
//     dw[kh][kw][ci][co] = sum over pixels (b, oh, ow) of  x[b, 2oh-1+kh, 2ow-1+kw, ci] * dy[b, oh, ow, co]
// a [48 x P] x [P x 128] product, P = B*Ho*Wo up to 2e5.  As a split-K launch of the general kernel it gathered its 48-row
// A operand with one scalar load per element (26 us at B = 64, 64 us at 3B, plus the reduce); the work is 0.8 GFLOP and the
// traffic is dy once (33.5 MB at B = 64).  Here one wave owns the whole 64 x 128 accumulator (rows 48..63 stay zero: 2 x 4
// MFMA 32x32x2 blocks, 128 registers) and walks its own run of pixels two at a time (k = pixel): per step a lane loads its
// A element (the x pixel under its tap for its k) twice (two row blocks), its B element (dy, 128-byte runs per pixel) four
// times, and issues 8 MFMAs; the next step's six loads are in flight meanwhile.  The four waves of a workgroup are joined in
// a fixed order through LDS, each workgroup writes one [48][128] partial and splitk_reduce sums the partials in slab order —
// deterministic, like every other reduction here.
// ------------------------------------------------------------------------------------------------------------------
typedef float stem_f32x16 __attribute__((ext_vector_type(16)));

template <bool DYH, int NB = 4>   // DYH: dy is a bf16 tensor (bf16 storage): 2-byte buffer loads, widened exactly; NB = Cout / 32 (4: wgancls, 2: gancls / StackGAN)
__global__ __launch_bounds__(256) void stem_k4s2_bwdf_kernel(const float* __restrict__ x, const void* __restrict__ dy, float* __restrict__ part,
                                                             int B, int H, int W, int Ho, int Wo, int pix_per_wave, FastDiv div_wo, FastDiv div_ho) {
  extern __shared__ __attribute__((aligned(16))) float red[];     // 2 x [64][CO]
  constexpr int CO = 32 * NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int P = B * Ho * Wo;                          // < 2^31: 32-bit index arithmetic throughout (a 64-bit divide is ~200 instructions)
  const int p_begin = (blockIdx.x * 4 + wave) * pix_per_wave;
  int p_end = p_begin + pix_per_wave;
  if (p_end > P) p_end = P;
  // this lane's two A rows: m = l31 and m = 32 + l31 (rows >= 48 are padding)
  int a_off[2]; bool a_ok[2]; int a_kh[2], a_kw[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int m = blk * 32 + l31;
    a_ok[blk] = m < 48;
    const int tap = m / 3, ci = m - tap * 3;
    a_kh[blk] = tap >> 2; a_kw[blk] = tap & 3;
    a_off[blk] = (a_kh[blk] * W + a_kw[blk]) * 3 + ci;
  }
  stem_f32x16 acc[2][NB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // loads go through buffer resources: padding taps, rows >= 48 and pixels past the wave's run select an out-of-range OFFSET
  // (the load returns 0), so nothing depends on a load's result until the MFMA that consumes it — a select on the loaded
  // value would put an s_waitcnt behind every load
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), (short)0, (int)((size_t)B * H * W * 3 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(dy), (short)0, (int)((size_t)P * CO * (DYH ? 2 : 4)), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  auto load = [&](int p, float (&a)[2], float (&b)[NB]) __attribute__((always_inline)) {
    const int pix = p + lh;                           // k = lh
    const bool ok = pix < p_end;
    const int pp = ok ? pix : p_begin;
    const int t = div_wo.div(pp);
    const int ow = pp - t * Wo;
    const int img = div_ho.div(t);
    const int oh = t - img * Ho;
    const int ih0 = 2 * oh - 1, iw0 = 2 * ow - 1;
    const int xbase = ((img * H + ih0) * W + iw0) * 3;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const bool in = ok & a_ok[blk] & ((unsigned)(ih0 + a_kh[blk]) < (unsigned)H) & ((unsigned)(iw0 + a_kw[blk]) < (unsigned)W);
      a[blk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, in ? (unsigned)(xbase + a_off[blk]) * 4u : OOB, 0, 0));
    }
    constexpr unsigned ES = DYH ? 2u : 4u;             // bytes per element of dy
    const unsigned ybase = ok ? ((unsigned)pp * (unsigned)CO + (unsigned)l31) * ES : OOB;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (DYH) b[j] = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(ry, ok ? ybase + j * 32u * ES : OOB, 0, 0) << 16);
      else b[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, ok ? ybase + j * 32u * ES : OOB, 0, 0));
    }
  };

  // 8 pixels (4 MFMA k-steps) per macro step; the 24 loads of the next macro step are issued before this one's 32 MFMAs,
  // so a full memory latency hides behind 2048 MFMA cycles (one wave per SIMD: nobody else would hide it)
  float a0[4][2], b0[4][NB], a1[4][2], b1[4][NB];
  auto load8 = [&](int p, float (&a)[4][2], float (&b)[4][NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 4; ++s) load(p + 2 * s, a[s], b[s]);
  };
  auto mma8 = [&](float (&a)[4][2], float (&b)[4][NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
  };
  load8(p_begin, a0, b0);                           // pixels beyond p_end load zeros (out-of-range offsets): no branch in the loop
  for (int p = p_begin; p < p_end; p += 16) {
    load8(p + 8, a1, b1);
    mma8(a0, b0);
    load8(p + 16, a0, b0);
    mma8(a1, b1);
  }

  // ---- join the four waves in a fixed order: (w0 + w2) + (w1 + w3) ---------------------------------------------------------
  auto put = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) buf[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * CO + j * 32 + l31] = acc[i][j][e];
  };
  auto add = [&](const float* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += buf[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * CO + j * 32 + l31];
  };
  float* buf0 = red;
  float* buf1 = red + 64 * CO;
  if (wave == 2) put(buf0);
  if (wave == 3) put(buf1);
  __syncthreads();
  if (wave == 0) add(buf0);
  if (wave == 1) add(buf1);
  __syncthreads();
  if (wave == 1) put(buf0);
  __syncthreads();
  if (wave == 0) {
    add(buf0);
    float* o = part + (size_t)blockIdx.x * 48 * CO;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (m < 48) o[m * CO + j * 32 + l31] = acc[i][j][e];
        }
  }
}

bool stem_bwdf_eligible(const t2i_conv_desc& d) { return stem_fwd_eligible(d); }

static int stem_bwdf_groups(const t2i_conv_desc& d, int* pix_per_wave) {
  const long P = (long)d.B * d.Ho * d.Wo;
  long ppw = (P + 1023) / 1024;                      // aim at 256 workgroups of 4 waves
  if (ppw < 16) ppw = 16;
  ppw = (ppw + 15) & ~15L;                           // whole double macro steps
  *pix_per_wave = (int)ppw;
  return (int)((P + 4 * ppw - 1) / (4 * ppw));
}

size_t stem_bwdf_ws(const t2i_conv_desc& d) {
  int ppw;
  return (size_t)stem_bwdf_groups(d, &ppw) * 48 * d.Cout * sizeof(float);
}

hipError_t stem_bwdf_launch(const t2i_conv_desc& d, const float* x, const void* dy, float* dw, int accumulate, void* ws, hipStream_t stream, bool dy_bf16) {
  int ppw;
  const int G = stem_bwdf_groups(d, &ppw);
  float* part = reinterpret_cast<float*>(ws);
  const size_t lds = (size_t)2 * 64 * d.Cout * sizeof(float);
  auto k = d.Cout == 128 ? (dy_bf16 ? stem_k4s2_bwdf_kernel<true, 4> : stem_k4s2_bwdf_kernel<false, 4>)
                         : (dy_bf16 ? stem_k4s2_bwdf_kernel<true, 2> : stem_k4s2_bwdf_kernel<false, 2>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  FastDiv dwo, dho;
  dwo.set((uint32_t)d.Wo); dho.set((uint32_t)d.Ho);
  hipLaunchKernelGGL(k, dim3(G), dim3(256), lds, stream, x, dy, part, d.B, d.H, d.W, d.Ho, d.Wo, ppw, dwo, dho);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return splitk_reduce_launch(part, G, (size_t)48 * d.Cout, nullptr, d.Cout, T2I_ACT_NONE, 0.f, dw, accumulate, stream);
}

bool tiny_bwdw_eligible(const t2i_conv_desc& d) { return d.Cin == 3 && d.Cout == 3 && d.KH <= 3 && d.KW <= 3; }

static int tiny_bwdw_blocks(const t2i_conv_desc& d) {
  size_t npix = (size_t)d.B * d.Ho * d.Wo;
  size_t blocks = (npix + 256 * 8 - 1) / (256 * 8);
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

size_t tiny_bwdw_ws(const t2i_conv_desc& d) {
  const int nb = tiny_bwdw_blocks(d), C = d.KH * d.KW * 9;
  return (size_t)nb * C * sizeof(float) + col_reduce_ws(nb, C) + 256;
}

hipError_t tiny_bwdw_launch(const t2i_conv_desc& d, const float* x, const float* dy, float* dw, int accumulate, void* ws,
                            hipStream_t stream) {
  const int nb = tiny_bwdw_blocks(d), C = d.KH * d.KW * 9;
  float* part = reinterpret_cast<float*>(ws);
  char* ws2 = reinterpret_cast<char*>(ws) + (((size_t)nb * C * sizeof(float) + 255) & ~(size_t)255);
  hipLaunchKernelGGL((tiny_bwdw_kernel<3, 3>), dim3(nb), dim3(256), 0, stream, x, dy, part, d);
  return col_reduce_launch(part, nullptr, nullptr, nb, C, dw, nullptr, accumulate, ws2, stream);
}

}  // namespace t2i
