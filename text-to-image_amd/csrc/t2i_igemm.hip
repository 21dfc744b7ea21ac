// t2i_igemm.hip — NHWC implicit-GEMM convolution family on the CDNA4 fp32 matrix pipe (gfx950 only).
//
// One templated kernel serves the three conv primitives (and dense layers as 1x1 convs):
//   FWD         y [M=B*Ho*Wo, N=Cout]   = im2col(x)[M,K=KH*KW*Cin]      * w[K,N]
//   BWD_DATA    dx[M=B*Hq*Wq, N=Cin]    = gather(dy)[M,K=taps*Cout]     * w^T[K,N]   (grid.z = stride phase)
//   BWD_FILTER  dw[M=KH*KW*Cin, N=Cout] = im2col(x)^T[M,K=B*Ho*Wo]      * dy[K,N]
// Arithmetic: v_mfma_f32_32x32x2_f32 — exact fp32 (an fmaf chain), 64 FLOP/clk/SIMD = the chip's fp32 peak.
// Structure: 256 threads = 4 waves (2x2), each wave owns WMT x WNT accumulator tiles of 32x32; BK = 32.
//   * operands are gathered with BUFFER loads (16 bytes where channels allow): padding taps, ragged edges and split-K
//     tails get an out-of-range offset and the hardware returns zeros — the K loop has no divergent branch at all
//     (a first version predicated each load with `ok ? *p : 0`; hipcc wrapped every load in an exec-mask branch, which
//     split the loop into ~40 basic blocks and serialised the address arithmetic in front of the MFMAs);
//   * tile t+1 sits in registers while tile t is multiplied; it is written to the other LDS buffer between the two
//     halves of tile t's MFMAs, then the loads of tile t+2 are issued: one barrier per K-tile, loads in flight for a
//     whole tile of MFMAs;
//   * MFMA operand fragments are double-buffered in registers (8-k chunk c+1 is read from LDS while chunk c is
//     multiplied), so no ds_read latency sits between MFMAs.
// Two LDS images, chosen per operand by which global dimension is contiguous:
//   K-inner  [rows][BK+4]   read with ds_read_b128 (4 k's per lane; row stride 36 dwords = 4*odd -> conflict free)
//   N-inner  [BK][cols]     read with ds_read_b32  (lane = column; consecutive lanes, conflict free).  (A k4-interleaved
//            [BK/4][cols][4] image read with ds_read_b128 was tried: 4x fewer LDS reads but 16 extra v_mov per tile for
//            the register transpose; measured 1-2% slower, so the plain image stays.)
// Both feed the same k-permutation: MFMA j of 8-k chunk c consumes k = 8c+j (lanes 0-31) and 8c+4+j (lanes 32-63).
// MATH = 1 (T2I_MATH_BF16): same gathers, same epilogue, but the operands are rounded to bf16 (RNE) on their way into LDS
// and multiplied by v_mfma_f32_32x32x16_bf16 with fp32 accumulation (BASELINE config 3: "bf16 MFMA + fp32 accumulate /
// master"; tensors in HBM stay fp32).  Both LDS images are then K-inner [rows][32 bf16 + 16 B pad] read with one
// ds_read_b128 per 16-k MFMA operand; an operand whose global layout is N-inner is transposed in registers by its
// loader (each thread fetches LD consecutive k-rows of one 4-column group and packs k-pairs), because the bf16 MFMA
// wants 8 consecutive k per lane for A and for B.
// Split-K (grid.y) writes full-layout partial slabs that splitk_reduce sums in a fixed order (deterministic),
// applying bias + activation there; without split-K the epilogue is fused here.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "t2i_internal.h"

namespace t2i {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;
constexpr int KPAD = 4;              // K-inner row stride = 36 dwords
constexpr int KSTRIDE = BK + KPAD;
constexpr unsigned OOB = 0xFFFFFFF0u;  // byte offset beyond every legal buffer: the load returns 0
constexpr int HSTRIDE = 20;          // bf16 image: row stride in dwords (32 bf16 = 16 dwords + 4 pad; 20r mod 64 is conflict
                                     // free for ds_read_b128's 16-lane groups)

template <int MODE, int WMT, int WNT, int MATH>
struct Smem {
  static constexpr int BM = 64 * WMT, BN = 64 * WNT;
  static constexpr bool A_KINNER = (MODE != MODE_BWD_FILTER);
  static constexpr bool B_KINNER = (MODE == MODE_BWD_DATA);
  static constexpr int A_ELEMS = MATH ? BM * HSTRIDE : (A_KINNER ? BM * KSTRIDE : BK * BM);   // dwords per buffer
  static constexpr int B_ELEMS = MATH ? BN * HSTRIDE : (B_KINNER ? BN * KSTRIDE : BK * BN);
  static constexpr int BYTES = 2 * (A_ELEMS + B_ELEMS) * 4;
};

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {      // v_cvt_pk_bf16_f32: round to nearest even
  f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, int elem_off, bool ok) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? (unsigned)elem_off * 4u : OOB, 0, 0);
  return __builtin_bit_cast(float4, v);
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, int elem_off, bool ok) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, ok ? (unsigned)elem_off * 4u : OOB, 0, 0));
}

// ------------------------------------------------------------------------------------------------------------------
// Operand address generators.  Each yields the element offset of a GEMM element and whether it exists.
// ------------------------------------------------------------------------------------------------------------------
struct RowFwd {  // an output pixel of FWD (A rows) or an input pixel of BWD_DATA (A rows): fixed per thread
  int base;      // pixel index of (b, 0, 0) in the tensor being gathered
  int h0, w0;    // FWD: oh*SH-pad_t, ow*SW-pad_l.  BWD_DATA: ihq+oh_off, iwq+ow_off
  bool ok;
};

template <int MODE>
__device__ __forceinline__ RowFwd make_row(const IgemmParams& p, const PhaseInfo& pi, int m) {
  RowFwd r;
  r.ok = m < p.M;
  int mm = r.ok ? m : 0;
  if (MODE == MODE_FWD) {
    int b = p.div_howo.div(mm);
    int rem = mm - b * p.howo;
    int oh = p.div_wo.div(rem);
    int ow = rem - oh * p.d.Wo;
    r.base = b * p.d.H * p.d.W;
    r.h0 = oh * p.d.SH - p.d.pad_t;
    r.w0 = ow * p.d.SW - p.d.pad_l;
  } else {  // BWD_DATA: m -> (b, ihq, iwq) within the phase
    int b = p.div_hqwq.div(mm);
    int rem = mm - b * p.hqwq;
    int ihq = p.div_wq.div(rem);
    int iwq = rem - ihq * p.Wq;
    r.base = b * p.d.Ho * p.d.Wo;
    r.h0 = ihq + pi.oh_off;
    r.w0 = iwq + pi.ow_off;
    // rows past the image edge (H not a multiple of SH) do not exist
    r.ok = r.ok && (ihq * p.d.SH + pi.ph < p.d.H) && (iwq * p.d.SW + pi.pw < p.d.W);
  }
  return r;
}

// A operand of FWD / BWD_DATA: (tap, c) = decode of reduction index k; `k` is a multiple of 4 on the vector path.
template <int MODE>
__device__ __forceinline__ void a_offset(const IgemmParams& p, const PhaseInfo& pi, const RowFwd& r, int k, int kend,
                                         int& off, bool& ok) {
  const int C = (MODE == MODE_FWD) ? p.d.Cin : p.d.Cout;  // channels of the gathered tensor
  int tap = p.div_c.div(k);
  int c = k - tap * C;
  if (MODE == MODE_FWD) {
    int kh = p.div_kw.div(tap);
    int kw = tap - kh * p.d.KW;
    int ih = r.h0 + kh, iw = r.w0 + kw;
    ok = r.ok & (k < kend) & ((unsigned)ih < (unsigned)p.d.H) & ((unsigned)iw < (unsigned)p.d.W);
    off = (r.base + ih * p.d.W + iw) * C + c;
  } else {
    int jh = pi.div_ntw.div(tap);
    int jw = tap - jh * pi.ntw;
    int oh = r.h0 - jh, ow = r.w0 - jw;
    ok = r.ok & (k < kend) & ((unsigned)oh < (unsigned)p.d.Ho) & ((unsigned)ow < (unsigned)p.d.Wo);
    off = (r.base + oh * p.d.Wo + ow) * C + c;
  }
}

// B operand of BWD_DATA (w^T): rows n = ci, reduction k = (jh, jw, co); contiguous along co.
__device__ __forceinline__ void bT_offset(const IgemmParams& p, const PhaseInfo& pi, int n, int k, int kend, int& off,
                                          bool& ok) {
  int tap = p.div_c.div(k);
  int co = k - tap * p.d.Cout;
  int jh = pi.div_ntw.div(tap);
  int jw = tap - jh * pi.ntw;
  int kh = pi.kh0 + jh * p.d.SH, kw = pi.kw0 + jw * p.d.SW;
  ok = (n < p.N) & (k < kend);
  off = ((kh * p.d.KW + kw) * p.d.Cin + n) * p.d.Cout + co;
}

// phase[idx] of the kernel's (only) argument, fetched from the kernel-argument segment in the constant address space
__device__ __forceinline__ PhaseInfo load_phase(int idx) {
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
// VAR: 0 = scalar gathers (channels not a multiple of 4); 1 = 16-byte gathers, per-element tap decode;
//      2 = 16-byte gathers with a TAP-UNIFORM K-tile (gathered channels % 32 == 0): one K-tile never straddles a filter
//          tap, so (kh,kw,c0) are computed once per tile on the scalar unit and each load costs ~8 VALU instructions
//          instead of ~25 (measured: the address arithmetic was the largest non-MFMA cost, 10-14% of the kernel).
template <int MODE, int WMT, int WNT, int VAR, int MATH>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmParams p) {
  constexpr bool VEC = VAR >= 1;
  constexpr bool UT = VAR == 2;
  constexpr bool BF = MATH == 1;
  using S = Smem<MODE, WMT, WNT, MATH>;
  constexpr int BM = S::BM, BN = S::BN;
  constexpr bool A_KINNER = S::A_KINNER, B_KINNER = S::B_KINNER;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                     // 2 buffers
  float* Bs = smem + 2 * S::A_ELEMS;    // 2 buffers

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- block -> (tile, split, phase).  blockIdx.x walks M tiles fastest so that consecutive workgroups (which the
  // dispatcher spreads over the 8 XCDs) share the same filter panel in every L2.
  const int tiles_m = p.tiles_m;
  // XCD-aware remap: the dispatcher places workgroup b on XCD b % 8 (each XCD has its own 4 MB L2).  Give every XCD a
  // contiguous run of tile ids, so the workgroups sharing one filter panel (consecutive M tiles) hit the same L2.
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;      // bijective for any nblk
  }
  // Grouped rasterisation inside each XCD's run: tile ids walk GROUP_N output-column tiles for every M tile before moving
  // down, so the ~64-96 workgroups an XCD runs at once form a block of (8-12 M tiles) x (GROUP_N N tiles) and share BOTH
  // operand panels through that XCD's 4 MB L2 (M-fastest order shared only the filter panel: every A panel was
  // re-fetched from HBM once per N tile — 220 MB of HBM-side traffic per launch against ~20 MB of operands).
  // Batched launches (Winograd's tile positions) with batch_lin: grid.x = nbatch * tiles and the XCD run above is cut
  // POSITION-major — an XCD works through whole positions one after the other, so a position's V panel and filter plane
  // (1-4 MB together) are fetched into ONE L2 once, instead of every XCD streaming a slice of all positions at the same
  // time (16 positions x (its A rows + the whole filter plane) = 10-36 MB against the 4 MB L2: every filter plane was
  // fetched by all eight XCDs).
  int bz_lin = 0;
  if (p.batch_lin) {
    const int tiles = tiles_m * p.tiles_n;
    bz_lin = bid / tiles;
    bid -= bz_lin * tiles;
  }
  int tile_m, tile_n;
  {
    const int g = p.group_n;                       // min(tiles_n, 8), host side
    const int per_group = g * tiles_m;
    const int grp = bid / per_group;
    const int r = bid - grp * per_group;
    const int n0 = grp * g;
    const int width = min(g, p.tiles_n - n0);
    tile_m = r / width;
    tile_n = n0 + (r - tile_m * width);
  }
  const int bm = tile_m * BM;
  const int bn = tile_n * BN;
  const int split = blockIdx.y;
  // The stride phase of this workgroup, read from the kernel-argument segment through an explicit constant-address-space
  // pointer (scalar loads with a uniform dynamic offset).  Indexing the by-value argument itself, `p.phase[blockIdx.z]`,
  // made hipcc copy the whole struct to scratch in the scalar-gather variant (private_segment 1680 B in igemm<1,2,2,0,0>).
  const PhaseInfo pi = load_phase((MODE == MODE_BWD_DATA && p.nbatch <= 1) ? blockIdx.z : 0);     // batched launches: one phase
  const int Kdim = (MODE == MODE_BWD_DATA) ? pi.K : p.K;
  const int kbeg = split * p.k_per_split;
  const int kend = min(Kdim, kbeg + p.k_per_split);
  const int ntiles = (kend - kbeg + BK - 1) / BK;

  // FWD launches may carry several independent GEMMs of one shape (grid.z): operand and output bases step per batch
  const int64_t bz = p.nbatch > 1 ? (p.batch_lin ? (int64_t)bz_lin : (int64_t)blockIdx.z) : 0;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a + bz * p.batch_a), (short)0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b + bz * p.batch_b), (short)0, (int)p.b_bytes, 0x00020000);

  // ---- per-thread loader state ---------------------------------------------------------------------------------
  // K-inner image: thread -> (k quad kq = tid&7, rows r0 + 32*i).   N-inner image: thread -> (col quad, k rows).
  constexpr int A_LD = A_KINNER ? BM / 32 : (BK * BM / 4) / 256;   // 16-byte pieces per thread per tile
  constexpr int B_LD = B_KINNER ? BN / 32 : (BK * BN / 4) / 256;
  const int kq = tid & 7, r0 = tid >> 3;
  RowFwd arow[A_KINNER ? A_LD : 1];
  if (A_KINNER) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) arow[i] = make_row<MODE>(p, pi, bm + r0 + 32 * i);
  }
  // tap-uniform fast path: everything that does not depend on the K-tile is folded into one per-row constant
  int a_rowoff[A_KINNER ? A_LD : 1];
  if (UT && A_KINNER) {
    const int Wsrc = (MODE == MODE_FWD) ? p.d.W : p.d.Wo;
    const int Csrc = (MODE == MODE_FWD) ? p.d.Cin : p.d.Cout;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) a_rowoff[i] = (arow[i].base + arow[i].h0 * Wsrc + arow[i].w0) * Csrc + kq * 4;
  }
  constexpr int A_C4 = BM / 4, B_C4 = BN / 4;          // float4 columns per k-row (N-inner images)
  // fp32: thread -> (column quad, k rows kr0 + KSTEP*i).  bf16: thread -> (k group of LD CONSECUTIVE rows, column quad),
  // k group fastest over lanes, so that the packed k-pairs of one column are a contiguous run of LDS words (conflict-free
  // transposing store) while a wave still fetches whole 128-byte lines.
  constexpr int A_KSTEP = BF ? 1 : 256 / A_C4, B_KSTEP = BF ? 1 : 256 / B_C4;
  constexpr int A_KG = BK / (A_KINNER ? 1 : A_LD), B_KG = BK / (B_KINNER ? 1 : B_LD);
  const int a_c4 = BF ? tid / A_KG : tid % A_C4, a_kr0 = BF ? (tid % A_KG) * (A_KINNER ? 1 : A_LD) : tid / A_C4;
  const int b_c4 = BF ? tid / B_KG : tid % B_C4, b_kr0 = BF ? (tid % B_KG) * (B_KINNER ? 1 : B_LD) : tid / B_C4;
  // BWD_FILTER A (x gathered, rows i=(kh,kw,ci) contiguous in ci): per-thread fixed (kh,kw,ci) for its 4 columns
  int fa_kh[4], fa_kw[4], fa_ci[4];
  bool fa_ok[4];
  if (MODE == MODE_BWD_FILTER) {
#pragma unroll
    for (int e = 0; e < (VEC ? 1 : 4); ++e) {
      int i = bm + a_c4 * 4 + e;
      fa_ok[e] = i < p.M;
      int ii = fa_ok[e] ? i : 0;
      int tap = p.div_c.div(ii);
      fa_ci[e] = ii - tap * p.d.Cin;
      fa_kh[e] = p.div_kw.div(tap);
      fa_kw[e] = tap - fa_kh[e] * p.d.KW;
    }
  }

  // B operand constants (vector paths)
  int b_const[B_LD];
  bool b_nok[B_LD];
  if (VEC) {
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      if (B_KINNER) {          // w^T: row n = ci
        const int n = bn + r0 + 32 * i;
        b_nok[i] = n < p.N;
        b_const[i] = n * p.d.Cout + kq * 4;
      } else {                 // row-major [K, N]
        const int n = bn + b_c4 * 4;
        b_nok[i] = n < p.N;
        b_const[i] = (b_kr0 + B_KSTEP * i) * p.N + n;
      }
    }
  }

  // BWD_FILTER "pixel walk" (VAR 2, Wo | 32): the reduction index advances by exactly 32 pixels per K-tile, so each
  // thread's gather position is decoded ONCE and then walked (oh += doh with carry into b) instead of two divisions per load
  int wk_b[A_KINNER ? 1 : A_LD], wk_oh[A_KINNER ? 1 : A_LD], wk_ofs[A_KINNER ? 1 : A_LD];
  bool wk_ok[A_KINNER ? 1 : A_LD];
  if (!A_KINNER && UT) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int r = kbeg + a_kr0 + A_KSTEP * i;
      int b = p.div_howo.div(r);
      int rem = r - b * p.howo;
      int oh = p.div_wo.div(rem);
      int ow = rem - oh * p.d.Wo;
      const int iw = ow * p.d.SW - p.d.pad_l + fa_kw[0];
      wk_b[i] = b; wk_oh[i] = oh;
      wk_ok[i] = fa_ok[0] & ((unsigned)iw < (unsigned)p.d.W);
      wk_ofs[i] = iw * p.d.Cin + fa_ci[0];
    }
  }

  float4 areg[A_LD], breg[B_LD];

  auto load_tile = [&](int t) __attribute__((always_inline)) {   // tiles past the end read as zeros (k >= kend), so the loop needs no tail branch
    const int k0 = kbeg + t * BK;
    // ---------------- A ----------------
    if (A_KINNER && UT) {
      const int Csrc = (MODE == MODE_FWD) ? p.d.Cin : p.d.Cout;
      const int tap = p.div_c.div(k0);            // wave-uniform: scalar unit
      const int c0 = k0 - tap * Csrc;
      const bool kok = (k0 + kq * 4) < kend;
      int s_off, dh, dw;
      if (MODE == MODE_FWD) {
        dh = p.div_kw.div(tap);
        dw = tap - dh * p.d.KW;
        s_off = (dh * p.d.W + dw) * Csrc + c0;
      } else {
        const int jh = pi.div_ntw.div(tap);
        const int jw = tap - jh * pi.ntw;
        dh = -jh; dw = -jw;
        s_off = c0 - (jh * p.d.Wo + jw) * Csrc;
      }
      const unsigned Hs = (MODE == MODE_FWD) ? p.d.H : p.d.Ho, Ws = (MODE == MODE_FWD) ? p.d.W : p.d.Wo;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const bool ok = arow[i].ok & kok & ((unsigned)(arow[i].h0 + dh) < Hs) & ((unsigned)(arow[i].w0 + dw) < Ws);
        areg[i] = bload4(ra, a_rowoff[i] + s_off, ok);
      }
    } else if (A_KINNER) {
      const int k = k0 + kq * 4;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        if (VEC) {
          int off; bool ok;
          a_offset<MODE>(p, pi, arow[i], k, kend, off, ok);
          areg[i] = bload4(ra, off, ok);
        } else {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int off; bool ok;
            a_offset<MODE>(p, pi, arow[i], k + e, kend, off, ok);
            v[e] = bload1(ra, off, ok);
          }
          areg[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    } else if (UT) {  // BWD_FILTER, pixel walk (load_tile is called with consecutive t)
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const int r = k0 + a_kr0 + A_KSTEP * i;
        const int ih = wk_oh[i] * p.d.SH - p.d.pad_t + fa_kh[0];
        const bool ok = (r < kend) & wk_ok[i] & ((unsigned)ih < (unsigned)p.d.H);
        areg[i] = bload4(ra, (wk_b[i] * p.d.H + ih) * p.d.W * p.d.Cin + wk_ofs[i], ok);
        int oh = wk_oh[i] + p.walk_doh;
        const bool carry = oh >= p.d.Ho;
        wk_oh[i] = carry ? oh - p.d.Ho : oh;
        wk_b[i] += p.walk_db + (carry ? 1 : 0);
      }
    } else {  // BWD_FILTER: A[i, r] = x gathered; the reduction index is r = (b,oh,ow)
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const int r = k0 + a_kr0 + A_KSTEP * i;
        const bool rok = r < kend;
        const int rr = rok ? r : 0;
        int b = p.div_howo.div(rr);
        int rem = rr - b * p.howo;
        int oh = p.div_wo.div(rem);
        int ow = rem - oh * p.d.Wo;
        const int pix = b * p.d.H * p.d.W;
        const int h0 = oh * p.d.SH - p.d.pad_t, w0 = ow * p.d.SW - p.d.pad_l;
        if (VEC) {
          int ih = h0 + fa_kh[0], iw = w0 + fa_kw[0];
          bool ok = rok & fa_ok[0] & ((unsigned)ih < (unsigned)p.d.H) & ((unsigned)iw < (unsigned)p.d.W);
          areg[i] = bload4(ra, (pix + ih * p.d.W + iw) * p.d.Cin + fa_ci[0], ok);
        } else {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int ih = h0 + fa_kh[e], iw = w0 + fa_kw[e];
            bool ok = rok & fa_ok[e] & ((unsigned)ih < (unsigned)p.d.H) & ((unsigned)iw < (unsigned)p.d.W);
            v[e] = bload1(ra, (pix + ih * p.d.W + iw) * p.d.Cin + fa_ci[e], ok);
          }
          areg[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    // ---------------- B ----------------
    if (B_KINNER && UT) {   // BWD_DATA, tap-uniform: off = ((kh*KW+kw)*Cin + n)*Cout + c0 + kq*4
      const int tap = p.div_c.div(k0);
      const int c0 = k0 - tap * p.d.Cout;
      const int jh = pi.div_ntw.div(tap);
      const int jw = tap - jh * pi.ntw;
      const int s_off = ((pi.kh0 + jh * p.d.SH) * p.d.KW + (pi.kw0 + jw * p.d.SW)) * p.d.Cin * p.d.Cout + c0;
      const bool kok = (k0 + kq * 4) < kend;
#pragma unroll
      for (int i = 0; i < B_LD; ++i) breg[i] = bload4(rb, b_const[i] + s_off, b_nok[i] & kok);
    } else if (B_KINNER) {  // BWD_DATA: w^T, rows n = ci
      const int k = k0 + kq * 4;
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        const int n = bn + r0 + 32 * i;
        if (VEC) {
          int off; bool ok;
          bT_offset(p, pi, n, k, kend, off, ok);
          breg[i] = bload4(rb, off, ok);
        } else {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int off; bool ok;
            bT_offset(p, pi, n, k + e, kend, off, ok);
            v[e] = bload1(rb, off, ok);
          }
          breg[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    } else {  // FWD: w[k, n];  BWD_FILTER: dy[r, n] — plain row-major [K, N]
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        const int k = k0 + b_kr0 + B_KSTEP * i;
        const int n = bn + b_c4 * 4;
        if (VEC) {
          breg[i] = bload4(rb, k0 * p.N + b_const[i], (k < kend) & b_nok[i]);
        } else {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = bload1(rb, k * p.N + n + e, (k < kend) & (n + e < p.N));
          breg[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  };

  auto store_tile = [&](int buf) __attribute__((always_inline)) {
    float* as = As + buf * S::A_ELEMS;
    float* bs = Bs + buf * S::B_ELEMS;
    if (BF) {
      unsigned* au = reinterpret_cast<unsigned*>(as);
      unsigned* bu = reinterpret_cast<unsigned*>(bs);
      if (A_KINNER) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
          *reinterpret_cast<uint2*>(&au[(r0 + 32 * i) * HSTRIDE + kq * 2]) =
              make_uint2(pk_bf16(areg[i].x, areg[i].y), pk_bf16(areg[i].z, areg[i].w));
      } else {   // areg[i] = 4 columns of k-row a_kr0 + i: transpose to [column][k]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned* dst = &au[(a_c4 * 4 + j) * HSTRIDE + (a_kr0 >> 1)];
#pragma unroll
          for (int i = 0; i < A_LD; i += 2)
            dst[i >> 1] = pk_bf16(reinterpret_cast<const float*>(&areg[i])[j], reinterpret_cast<const float*>(&areg[i + 1])[j]);
        }
      }
      if (B_KINNER) {
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
          *reinterpret_cast<uint2*>(&bu[(r0 + 32 * i) * HSTRIDE + kq * 2]) =
              make_uint2(pk_bf16(breg[i].x, breg[i].y), pk_bf16(breg[i].z, breg[i].w));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned* dst = &bu[(b_c4 * 4 + j) * HSTRIDE + (b_kr0 >> 1)];
#pragma unroll
          for (int i = 0; i < B_LD; i += 2)
            dst[i >> 1] = pk_bf16(reinterpret_cast<const float*>(&breg[i])[j], reinterpret_cast<const float*>(&breg[i + 1])[j]);
        }
      }
      return;
    }
    if (A_KINNER) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) *reinterpret_cast<float4*>(&as[(r0 + 32 * i) * KSTRIDE + kq * 4]) = areg[i];
    } else {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) *reinterpret_cast<float4*>(&as[(a_kr0 + A_KSTEP * i) * BM + a_c4 * 4]) = areg[i];
    }
    if (B_KINNER) {
#pragma unroll
      for (int i = 0; i < B_LD; ++i) *reinterpret_cast<float4*>(&bs[(r0 + 32 * i) * KSTRIDE + kq * 4]) = breg[i];
    } else {
#pragma unroll
      for (int i = 0; i < B_LD; ++i) *reinterpret_cast<float4*>(&bs[(b_kr0 + B_KSTEP * i) * BN + b_c4 * 4]) = breg[i];
    }
  };

  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  if constexpr (BF) {
    // ---- bf16 main loop: 2 MFMA k-steps of 16 per K-tile; the loop is bound by the fp32 operand fetch, not by the MFMAs
    // (8 per wave-tile, 32 cycles each), so the schedule is the plain one: fragments of tile t -> registers, tile t+1
    // registers -> the other LDS buffer, loads of tile t+2, MFMAs, one barrier.
    struct FragH { bf16x8 a[WMT][2]; bf16x8 b[WNT][2]; };
    auto read_h = [&](FragH& f, const float* as, const float* bs) __attribute__((always_inline)) {
      const unsigned* au = reinterpret_cast<const unsigned*>(as);
      const unsigned* bu = reinterpret_cast<const unsigned*>(bs);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < WMT; ++i)
          f.a[i][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
              &au[(wm * 32 * WMT + i * 32 + l31) * HSTRIDE + s * 8 + lh * 4]));
#pragma unroll
        for (int i = 0; i < WNT; ++i)
          f.b[i][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
              &bu[(wn * 32 * WNT + i * 32 + l31) * HSTRIDE + s * 8 + lh * 4]));
      }
    };
    load_tile(0);
    store_tile(0);
    load_tile(1);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
      FragH f;
      read_h(f, As + (t & 1) * S::A_ELEMS, Bs + (t & 1) * S::B_ELEMS);
      store_tile((t + 1) & 1);
      load_tile(t + 2);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
          for (int n = 0; n < WNT; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][s], f.b[n][s], acc[i][n], 0, 0, 0);
      __syncthreads();
    }
  } else {
  // fragments of one 8-k chunk: 4 MFMA steps x (WMT + WNT) operands
  struct Frag { float a[WMT][4]; float b[WNT][4]; };

  auto read_frag = [&](Frag& f, const float* as, const float* bs, int c) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
      const int row = wm * 32 * WMT + i * 32 + l31;
      if (A_KINNER) {
        float4 v = *reinterpret_cast<const float4*>(&as[row * KSTRIDE + c * 8 + lh * 4]);
        f.a[i][0] = v.x; f.a[i][1] = v.y; f.a[i][2] = v.z; f.a[i][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.a[i][j] = as[(c * 8 + j + 4 * lh) * BM + row];
      }
    }
#pragma unroll
    for (int i = 0; i < WNT; ++i) {
      const int col = wn * 32 * WNT + i * 32 + l31;
      if (B_KINNER) {
        float4 v = *reinterpret_cast<const float4*>(&bs[col * KSTRIDE + c * 8 + lh * 4]);
        f.b[i][0] = v.x; f.b[i][1] = v.y; f.b[i][2] = v.z; f.b[i][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[i][j] = bs[(c * 8 + j + 4 * lh) * BN + col];
      }
    }
  };

  auto mma_frag = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int n = 0; n < WNT; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][j], f.b[n][j], acc[i][n], 0, 0, 0);
  };

  // ---- main loop ---------------------------------------------------------------------------------------------------
  // registers: tile t+1 | LDS buf[t&1]: tile t | LDS buf[(t+1)&1]: free until this iteration's store.
  // ONE barrier per K-tile, placed mid-tile right after the store of tile t+1.  Every ds_read of tile t is issued before
  // it, so (a) the barrier publishes tile t+1 and (b) it also proves every wave is done reading the buffer that the NEXT
  // iteration's store will overwrite.  After it, the first two fragment chunks of tile t+1 are fetched while the last
  // two chunks of tile t multiply: no LDS latency and no barrier at the tile boundary.
  // sched_barrier(0) pins this order; left alone hipcc sinks every ds_read next to its MFMAs (fewer live registers)
  // and the LDS latency lands between MFMA groups — ~20% of the matrix pipe idle.
  Frag fa0, fa1, fb0, fb1;
  auto k_tile = [&](int t, Frag& c0, Frag& c1, Frag& n0, Frag& n1) __attribute__((always_inline)) {
    const float* as = As + (t & 1) * S::A_ELEMS;
    const float* bs = Bs + (t & 1) * S::B_ELEMS;
    const float* an = As + ((t + 1) & 1) * S::A_ELEMS;
    const float* bn_ = Bs + ((t + 1) & 1) * S::B_ELEMS;
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(c0);                                   // chunk 0 (fetched during the previous tile)
    __builtin_amdgcn_sched_barrier(0);
    read_frag(c0, as, bs, 2);
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(c1);                                   // chunk 1, with the LDS store of tile t+1 interleaved
    store_tile((t + 1) & 1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, (WMT * WNT * 4) / 8 > 0 ? (WMT * WNT * 4) / 8 : 1, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x200, (A_LD + B_LD + 7) / 8, 0);                               // DS write
    }
    __builtin_amdgcn_sched_barrier(0);
    read_frag(c1, as, bs, 3);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_frag(n0, an, bn_, 0);                      // tile t+1, chunks 0 and 1
    read_frag(n1, an, bn_, 1);
    __builtin_amdgcn_sched_barrier(0);
    load_tile(t + 2);                               // zeros past the end; address math + buffer loads behind chunk 2
    mma_frag(c0);                                   // chunk 2
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, (WMT * WNT * 4) / 8 > 0 ? (WMT * WNT * 4) / 8 : 1, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x020, (A_LD + B_LD + 7) / 8, 0);                               // VMEM read
    }
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(c1);                                   // chunk 3
  };
  load_tile(0);
  store_tile(0);
  load_tile(1);
  __syncthreads();
  read_frag(fa0, As, Bs, 0);
  read_frag(fa1, As, Bs, 1);
  int t = 0;
  for (; t + 1 < ntiles; t += 2) {
    k_tile(t, fa0, fa1, fb0, fb1);
    k_tile(t + 1, fb0, fb1, fa0, fa1);
  }
  if (t < ntiles) k_tile(t, fa0, fa1, fb0, fb1);
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  float* out = p.c + bz * p.batch_c + (p.splitk > 1 ? (size_t)split * p.out_elems : 0);
  const bool fused = (p.splitk == 1);
  const bool want_stats = MODE == MODE_FWD && fused && p.stats != nullptr;
  float cs[WNT];               // this lane's column sums over its rows (batch-norm statistics of the layer output)
#pragma unroll
  for (int j = 0; j < WNT; ++j) cs[j] = 0.f;
  // Round 4: stores through LDS.  The direct form below issues 16 WMT WNT four-byte stores per thread, and a bf16-kernel ablation
  // (profiles/r04_bf16_gemm_fixed_cost.txt) showed that form costing 7 us of every launch — store-issue latency of the workgroup,
  // not bandwidth.  Each wave passes its sub-tile through a private LDS patch 32 rows at a time (the operand tiles are dead after
  // the barrier) and a lane then stores 4 consecutive columns of a row with one 16-byte instruction: 4 WMT WNT stores per thread.
  // Same element arithmetic (bias, activation, accumulate) in the same order: bit-identical.  N % 4 == 0, aligned bases.
  constexpr bool STAGE_FITS = S::BYTES >= 4 * 32 * (32 * WNT + 4) * 4;      // (the bf16-math images of the 64x128 tile are smaller)
  const bool vec_epi = STAGE_FITS && p.vec_epi && (p.N & 3) == 0 && (p.out_elems & 3) == 0 && (p.batch_c & 3) == 0 &&
                       ((reinterpret_cast<uintptr_t>(p.c) | reinterpret_cast<uintptr_t>(p.bias)) & 15) == 0;
  if (vec_epi) {
    constexpr int SROW = 32 * WNT + 4, CH = 8 * WNT;      // staging row stride (floats); 4-column chunks per row
    __syncthreads();                                       // every wave is done with the operand tiles in LDS
    float* st = smem + wave * (32 * SROW);
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
      for (int it = 0; it < 4 * WNT; ++it) {
        const int idx = it * 64 + lane, row = idx / CH, c4 = idx % CH;
        float4 v = *reinterpret_cast<const float4*>(&st[row * SROW + c4 * 4]);
        const int m = bm + (wm * WMT + i) * 32 + row;
        const int n = bn + wn * 32 * WNT + c4 * 4;
        bool mok = m < p.M;
        int rowoff;
        if (MODE == MODE_BWD_DATA) {
          int mm = mok ? m : 0;
          int b = p.div_hqwq.div(mm);
          int rem = mm - b * p.hqwq;
          int ihq = p.div_wq.div(rem);
          int iwq = rem - ihq * p.Wq;
          int ih = ihq * p.d.SH + pi.ph, iw = iwq * p.d.SW + pi.pw;
          mok = mok && ih < p.d.H && iw < p.d.W;
          rowoff = ((b * p.d.H + ih) * p.d.W + iw) * p.N;
        } else {
          rowoff = m * p.N;
        }
        if (mok && n < p.N) {
          if (fused) {
            if (p.bias) {
              const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
              v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            }
            v.x = apply_act(v.x, p.act, p.alpha); v.y = apply_act(v.y, p.act, p.alpha);
            v.z = apply_act(v.z, p.act, p.alpha); v.w = apply_act(v.w, p.act, p.alpha);
            if (p.accumulate) {
              const float4 o = *reinterpret_cast<const float4*>(out + rowoff + n);
              v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
          }
          *reinterpret_cast<float4*>(out + rowoff + n) = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (want_stats) {            // the column sums of the values just stored, from the accumulators (same arithmetic)
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int n = bn + wn * 32 * WNT + j * 32 + l31;
        const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = bm + wm * 32 * WMT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            float v = acc[i][j][e];
            if (p.bias) v += bv;
            v = apply_act(v, p.act, p.alpha);
            if (m < p.M && n < p.N) cs[j] += v;
          }
      }
    }
  } else
#pragma unroll
  for (int i = 0; i < WMT; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = bm + wm * 32 * WMT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      bool mok = m < p.M;
      int rowoff;
      if (MODE == MODE_BWD_DATA) {
        int mm = mok ? m : 0;
        int b = p.div_hqwq.div(mm);
        int rem = mm - b * p.hqwq;
        int ihq = p.div_wq.div(rem);
        int iwq = rem - ihq * p.Wq;
        int ih = ihq * p.d.SH + pi.ph, iw = iwq * p.d.SW + pi.pw;
        mok = mok && ih < p.d.H && iw < p.d.W;
        rowoff = ((b * p.d.H + ih) * p.d.W + iw) * p.N;
      } else {
        rowoff = m * p.N;
      }
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int n = bn + wn * 32 * WNT + j * 32 + l31;
        if (mok && n < p.N) {
          float v = acc[i][j][e];
          if (fused) {
            if (p.bias) v += p.bias[n];
            v = apply_act(v, p.act, p.alpha);
            if (p.accumulate) v += out[rowoff + n];
          }
          out[rowoff + n] = v;
          if (want_stats) cs[j] += v;
        }
      }
    }
  }
  // ---- fused batch-norm statistics (the conv that feeds a batch norm hands it the partials, so the normalisation needs no
  // extra pass over the tensor): per column the tile's SUM and its second moment ABOUT THE TILE'S OWN MEAN, M2 = sum (y - m)^2.
  // Two passes over the accumulators, which are still in registers: sums first (lane l and l^32 hold the same column; the two
  // waves stacked along M meet in LDS), then the centred squares.  The batch norm merges the tiles with Chan's update
  // (t2i_bn_stats_tiles); raw sums of squares would lose var = E[y^2] - E[y]^2 to cancellation when |mean| >> std.
  if (want_stats) {
    __syncthreads();                                   // every wave is done with the operand tiles in LDS
    float* red = smem;                                 // [2 (wm)][BN] sums, then [2 (wm)][BN] centred squares
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const float s0 = cs[j] + __shfl_xor(cs[j], 32, 64);
      if (lh == 0) red[wm * BN + wn * 32 * WNT + j * 32 + l31] = s0;
    }
    __syncthreads();
    const float rows_tile = (float)min(BM, p.M - bm);
    float cq[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int col = wn * 32 * WNT + j * 32 + l31;
      const int n = bn + col;
      const float mean = (red[col] + red[BN + col]) / rows_tile;
      const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = bm + wm * 32 * WMT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          const float dlt = apply_act(acc[i][j][e] + bv, p.act, p.alpha) - mean;
          q += (m < p.M) ? dlt * dlt : 0.f;
        }
      cq[j] = q + __shfl_xor(q, 32, 64);
    }
    __syncthreads();                                   // the sums have been read by everyone
    float tsum = 0.f;
    if (tid < BN) tsum = red[tid] + red[BN + tid];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WNT; ++j)
      if (lh == 0) red[wm * BN + wn * 32 * WNT + j * 32 + l31] = cq[j];
    __syncthreads();
    if (tid < BN && bn + tid < p.N) {
      const size_t tm = (size_t)tile_m;
      p.stats[tm * p.N + bn + tid] = tsum;
      p.stats[((size_t)tiles_m + tm) * p.N + bn + tid] = red[tid] + red[BN + tid];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Split-K reduction: out[i] = act(sum_s slab[s][i] + bias[i % N]); fixed summation order => deterministic.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void splitk_reduce_body(const float* __restrict__ slabs, int splitk, size_t out_elems,
                                                   const float* __restrict__ bias, int N, int act, float alpha,
                                                   float* __restrict__ out, int accumulate, uint2* __restrict__ outh,
                                                   unsigned vb, unsigned nvb) {
  const size_t n4 = out_elems >> 2;
  for (size_t i = (size_t)vb * 256 + threadIdx.x; i < n4; i += (size_t)nvb * 256) {
    float4 s = reinterpret_cast<const float4*>(slabs)[i];
    for (int k = 1; k < splitk; ++k) {
      float4 v = reinterpret_cast<const float4*>(slabs + (size_t)k * out_elems)[i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
      int c = (int)((i * 4) % (size_t)N);   // N % 4 == 0 on this path
      s.x += bias[c]; s.y += bias[c + 1]; s.z += bias[c + 2]; s.w += bias[c + 3];
    }
    s.x = apply_act(s.x, act, alpha); s.y = apply_act(s.y, act, alpha);
    s.z = apply_act(s.z, act, alpha); s.w = apply_act(s.w, act, alpha);
    if (accumulate) {
      const float4 o = reinterpret_cast<const float4*>(out)[i];
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    if (out) reinterpret_cast<float4*>(out)[i] = s;       // out == NULL: bf16 storage, only the bf16 tensor is written
    if (outh) {                     // bf16 copy of the finished output (twin, or THE tensor under bf16 storage)
      typedef float f2 __attribute__((ext_vector_type(2)));
      typedef __bf16 h2 __attribute__((ext_vector_type(2)));
      f2 lo = {s.x, s.y}, hi = {s.z, s.w};
      outh[i] = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, h2)), __builtin_bit_cast(unsigned, __builtin_convertvector(hi, h2)));
    }
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int splitk,
                                                            size_t out_elems, const float* __restrict__ bias, int N,
                                                            int act, float alpha, float* __restrict__ out, int accumulate,
                                                            uint2* __restrict__ outh) {
  splitk_reduce_body(slabs, splitk, out_elems, bias, N, act, alpha, out, accumulate, outh, blockIdx.x, gridDim.x);
}

// The reductions of the two GEMMs of one launch (igemm_pair_kernel) in one launch: blocks [0, a.blocks) sum a's slabs, the rest b's —
// each exactly as splitk_reduce_kernel would (same order, same bits).
__global__ __launch_bounds__(256) void splitk_reduce2_kernel(ReduceJob a, ReduceJob b) {
  if (blockIdx.x < (unsigned)a.blocks)
    splitk_reduce_body(a.slabs, a.splitk, a.out_elems, a.bias, a.N, a.act, a.alpha, a.out, a.accumulate, reinterpret_cast<uint2*>(a.out_h), blockIdx.x, a.blocks);
  else
    splitk_reduce_body(b.slabs, b.splitk, b.out_elems, b.bias, b.N, b.act, b.alpha, b.out, b.accumulate, reinterpret_cast<uint2*>(b.out_h),
                       blockIdx.x - a.blocks, b.blocks);
}

// Deep splits of small outputs (the critic's first-layer filter gradient: 48x128 outputs, 128-256 slabs): one thread per
// output would walk hundreds of slabs in sequence (43 us, pure latency).  Here 16 slab lanes share an output: lane ty sums
// slabs ty, ty+16, ... and the 16 partial sums are joined in lane order through LDS — still a fixed order.
__global__ __launch_bounds__(256) void splitk_reduce_deep_kernel(const float* __restrict__ slabs, int splitk,
                                                                 size_t out_elems, const float* __restrict__ bias, int N,
                                                                 int act, float alpha, float* __restrict__ out, int accumulate,
                                                                 uint2* __restrict__ outh) {
  __shared__ float4 red[16][16];
  const size_t n4 = out_elems >> 2;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + tx;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    for (int k = ty; k < splitk; k += 16) {
      const float4 v = reinterpret_cast<const float4*>(slabs + (size_t)k * out_elems)[i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < n4) {
    s = red[0][tx];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 v = red[k][tx]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (bias) {
      int c = (int)((i * 4) % (size_t)N);
      s.x += bias[c]; s.y += bias[c + 1]; s.z += bias[c + 2]; s.w += bias[c + 3];
    }
    s.x = apply_act(s.x, act, alpha); s.y = apply_act(s.y, act, alpha);
    s.z = apply_act(s.z, act, alpha); s.w = apply_act(s.w, act, alpha);
    if (accumulate) {
      const float4 o = reinterpret_cast<const float4*>(out)[i];
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    if (out) reinterpret_cast<float4*>(out)[i] = s;
    if (outh) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      typedef __bf16 h2 __attribute__((ext_vector_type(2)));
      f2 lo = {s.x, s.y}, hi = {s.z, s.w};
      outh[i] = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, h2)), __builtin_bit_cast(unsigned, __builtin_convertvector(hi, h2)));
    }
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_scalar_kernel(const float* __restrict__ slabs, int splitk,
                                                                   size_t out_elems, const float* __restrict__ bias,
                                                                   int N, int act, float alpha,
                                                                   float* __restrict__ out, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < out_elems; i += (size_t)gridDim.x * blockDim.x) {
    float s = slabs[i];
    for (int k = 1; k < splitk; ++k) s += slabs[(size_t)k * out_elems + i];
    if (bias) s += bias[i % (size_t)N];
    s = apply_act(s, act, alpha);
    out[i] = accumulate ? s + out[i] : s;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host-side launch
// ------------------------------------------------------------------------------------------------------------------
template <int MODE, int WMT, int WNT, int VAR, int MATH>
static hipError_t launch_cfg(const IgemmParams& p, dim3 grid, hipStream_t stream) {
  using S = Smem<MODE, WMT, WNT, MATH>;
  auto k = igemm_kernel<MODE, WMT, WNT, VAR, MATH>;
  static bool attr_done = false;   // benign race: idempotent
  if (!attr_done && S::BYTES > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, S::BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(k, grid, dim3(256), S::BYTES, stream, p);
  return hipGetLastError();
}

template <int MODE, int MATH>
static hipError_t launch_mode(const IgemmParams& p, int wmt, int wnt, int var, dim3 grid, hipStream_t stream) {
#define T2I_CASE(a, b)                                                             \
  if (wmt == a && wnt == b) {                                                      \
    if (var == 2) return launch_cfg<MODE, a, b, 2, MATH>(p, grid, stream);         \
    return var >= 1 ? launch_cfg<MODE, a, b, 1, MATH>(p, grid, stream) : launch_cfg<MODE, a, b, 0, MATH>(p, grid, stream); \
  }
  T2I_CASE(2, 2)
  T2I_CASE(2, 1)
  T2I_CASE(1, 2)
  T2I_CASE(1, 1)
#undef T2I_CASE
  return hipErrorInvalidValue;
}

hipError_t igemm_launch(int mode, const IgemmParams& p, int wmt, int wnt, int var, hipStream_t stream) {
  dim3 grid(p.tiles_m * p.tiles_n, p.splitk, p.nbatch > 1 ? p.nbatch : (mode == MODE_BWD_DATA ? p.nphase : 1));
  if (p.nbatch > 1 && p.batch_lin) { grid.x *= p.nbatch; grid.z = 1; }
  if (p.d.math == T2I_MATH_BF16) {
    switch (mode) {
      case MODE_FWD: return launch_mode<MODE_FWD, 1>(p, wmt, wnt, var, grid, stream);
      case MODE_BWD_DATA: return launch_mode<MODE_BWD_DATA, 1>(p, wmt, wnt, var, grid, stream);
      case MODE_BWD_FILTER: return launch_mode<MODE_BWD_FILTER, 1>(p, wmt, wnt, var, grid, stream);
    }
    return hipErrorInvalidValue;
  }
  switch (mode) {
    case MODE_FWD: return launch_mode<MODE_FWD, 0>(p, wmt, wnt, var, grid, stream);
    case MODE_BWD_DATA: return launch_mode<MODE_BWD_DATA, 0>(p, wmt, wnt, var, grid, stream);
    case MODE_BWD_FILTER: return launch_mode<MODE_BWD_FILTER, 0>(p, wmt, wnt, var, grid, stream);
  }
  return hipErrorInvalidValue;
}

// true: both jobs take splitk_reduce_kernel's vector path (not the deep or the scalar one), so one launch can serve both
bool splitk_reduce2_ok(const ReduceJob& a, const ReduceJob& b) {
  auto plain = [](const ReduceJob& j) {
    return j.splitk > 1 && (j.out_elems & 3) == 0 && (j.N & 3) == 0 && !(j.splitk >= 32 && (j.out_elems >> 2) <= 65536);
  };
  return plain(a) && plain(b);
}

hipError_t splitk_reduce2_launch(ReduceJob a, ReduceJob b, hipStream_t stream) {
  auto nblocks = [](const ReduceJob& j) {
    int blocks = (int)(((j.out_elems >> 2) + 255) / 256);
    return blocks > 4096 ? 4096 : blocks < 1 ? 1 : blocks;
  };
  a.blocks = nblocks(a); b.blocks = nblocks(b);
  hipLaunchKernelGGL(splitk_reduce2_kernel, dim3(a.blocks + b.blocks), dim3(256), 0, stream, a, b);
  return hipGetLastError();
}

hipError_t splitk_reduce_launch(const float* slabs, int splitk, size_t out_elems, const float* bias, int N, int act,
                                float alpha, float* out, int accumulate, hipStream_t stream, void* out_h, bool* wrote_h) {
  if (wrote_h) *wrote_h = false;
  if ((out_elems & 3) == 0 && (N & 3) == 0) {
    size_t n4 = out_elems >> 2;
    if (splitk >= 32 && n4 <= 65536) {           // few outputs, many slabs: parallelise over slabs too
      hipLaunchKernelGGL(splitk_reduce_deep_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, stream, slabs, splitk, out_elems,
                         bias, N, act, alpha, out, accumulate, reinterpret_cast<uint2*>(out_h));
      if (wrote_h && out_h) *wrote_h = true;
      return hipGetLastError();
    }
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, slabs, splitk, out_elems, bias, N, act,
                       alpha, out, accumulate, reinterpret_cast<uint2*>(out_h));
    if (wrote_h && out_h) *wrote_h = true;
  } else {
    int blocks = (int)((out_elems + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(splitk_reduce_scalar_kernel, dim3(blocks), dim3(256), 0, stream, slabs, splitk, out_elems, bias,
                       N, act, alpha, out, accumulate);
  }
  return hipGetLastError();
}

}  // namespace t2i
