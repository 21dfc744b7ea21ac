// t2i_bgemm.hip — batched plain GEMMs of the Winograd paths on the fp32 matrix pipe, PERSISTENT workgroups (gfx950 only).
//
// The Winograd convolutions (t2i_winograd.hip) spend their matrix time in 9-36 independent GEMMs of one shape per launch,
// with K = the channel count only (256-1152): 8-36 K-tiles per output tile.  Launched through igemm_kernel (one workgroup
// per 64x64 output tile, 2-4 rounds of workgroups per launch) a K sweep measured  t = 12.5 us + 4.8 us x K-tiles  for the
// 2048-tile shape (profiles/r03_batched_gemm_ksweep.txt): the K loop itself runs at 72 % of the fp32 MFMA peak — what this
// chip's exact-fp32 MFMA sustains (cdna_hip_programming.md quotes 122 of 157 TF/s at 4096^3) — and the fixed 12.5 us are
// the launch ramp plus one PROLOGUE BUBBLE PER ROUND: the 4 workgroups sharing a CU start together, so they also finish
// together, and the 4 that replace them all sit in their prologue (address set-up, two dependent operand fetches, first
// LDS fill: ~2 us) at the same time with nothing to multiply.
// Here a launch has at most as many workgroups as fit on the chip at once (4 per CU), and each walks through its share of
// the output tiles WITHOUT leaving the K loop: the operand fetches of tile i+1's first two K-tiles are issued during the
// last two K-tiles of tile i (the loader state is switched on the fly), tile i's accumulators are stored while tile
// i+1's fragments are already in registers, and the matrix pipe never waits for a prologue again.
//   C[z][M,N] = op(A[z]) * op(B[z])   for z < nbatch, three operand layouts (what the three conv primitives need):
//     LAY 0  A [M][K] K-inner, B [K][N] N-inner      forward conv        (V [T,Cin]  x U [Cin,Cout])
//     LAY 1  A [M][K] K-inner, B [N][K] K-inner      input gradient      (V [T,Cout] x U [Cin,Cout]^T)
//     LAY 2  A [K][M] M-inner, B [K][N] N-inner      filter gradient     (V [T,Cin]^T x dY [T,Cout])
// Tile 64 WM x 64 WN (round 4: WM, WN in {1, 2}; rounds 1-3: 64x64 only), 4 waves (2x2) of WM x WN 32x32 accumulators, BK = 32,
// v_mfma_f32_32x32x2_f32: the LDS images, the k permutation of the fragments and the one-barrier-per-K-tile schedule are
// igemm_kernel's (t2i_igemm.hip) — an output element is the same fmaf chain over k in the same order whatever the tile, so
// results are bit-identical to the per-tile launch and across tile shapes.  Per multiply-add a 128x128 tile streams half the operand bytes of a 64x64 one into the CU (32 instead of 16 FLOP per L2
// byte) — the K loop is co-limited by exactly that stream (below) — at the price of 4x coarser work items, so bgemm_launch's
// caller picks the tile by item count (t2i_capi.hip: run_batched_gemm).
// What bounds the K loop (timing-only ablations of this kernel, 2048 tiles, profiles/r03_bgemm_ablation.txt): dropping the barrier
// changes nothing; a second accumulator per wave (no MFMA -> MFMA dependency) nothing; loads three K-tiles ahead through a
// second register set nothing; direct-to-LDS DMA (buffer_load ... lds into an XOR-swizzled unpadded image, issued from inline
// assembly so that hipcc does not wait for it before every ds_read) nothing — while dropping the operand fetches altogether is
// worth 14 % and the LDS stores another 4 %: the loop is co-limited by the L2 -> CU operand stream itself (64 KB per CU per
// K-tile round, ~12 B/clk/CU), not by latency, issue order or the staging path.
// Work order: items w = z * tiles + tile (position-major); XCD x owns a contiguous run of items (a position's operands
// meet in ONE 4 MB L2), its S workgroups take items run + s, run + s + S, ... so that at any time an XCD works on 1-2
// positions.  Ragged edges (M, N not multiples of 64; K not a multiple of 32) read zeros through the buffer range check.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "t2i_internal.h"

namespace t2i {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int BK = 32;
constexpr int KSTRIDE = BK + 4;            // K-inner LDS row stride (36 dwords: conflict-free ds_read_b128)
constexpr unsigned OOB = 0xFFFFFFF0u;

// NW waves as (NW / 2) x 2: 4 waves = the 64 WM x 64 WN tile of rounds 3-4; 8 waves (round 5) = 128 WM x 64 WN on 512 threads — per wave the
// same accumulators, fragments and K loop, per workgroup 25 % fewer operand bytes per multiply-add (the A panel serves two more wave rows)
template <int LAY, int WM, int WN, int NW = 4>
struct BSmem {
  static constexpr bool A_KIN = LAY != 2, B_KIN = LAY == 1;
  static constexpr int BM = 32 * (NW / 2) * WM, BN = 64 * WN;
  static constexpr int A_ELEMS = A_KIN ? BM * KSTRIDE : BK * BM;
  static constexpr int B_ELEMS = B_KIN ? BN * KSTRIDE : BK * BN;
  static constexpr int BYTES = 2 * (A_ELEMS + B_ELEMS) * 4;
};

__device__ __forceinline__ float4 bl4(__amdgpu_buffer_rsrc_t r, int elem_off, bool ok) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? (unsigned)elem_off * 4u : OOB, 0, 0);
  return __builtin_bit_cast(float4, v);
}
}  // namespace

template <int LAY, int WM, int WN, int NW = 4>
__global__ __launch_bounds__(64 * NW) void bgemm_kernel(BgemmParams p) {
  using S = BSmem<LAY, WM, WN, NW>;
  constexpr bool A_KIN = S::A_KIN, B_KIN = S::B_KIN;
  constexpr int BM = S::BM, BN = S::BN;
  constexpr int NT = 64 * NW, RPP = NT / 8;                // threads; rows of a K-inner image one pass of the loader covers
  constexpr int A_LD = BM * 8 / NT, B_LD = BN * 8 / NT;    // 16-byte pieces per thread and K-tile
  // M/N-inner image [32 k][cols]: cols / 4 threads per k-row, NT / (cols / 4) k-rows per pass
  constexpr int A_C4 = BM / 4, A_KR = NT / A_C4, B_C4 = BN / 4, B_KR = NT / B_C4;
  extern __shared__ __attribute__((aligned(16))) float smem_b[];
  float* As = smem_b;
  float* Bs = smem_b + 2 * S::A_ELEMS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- this workgroup's items -----------------------------------------------------------------------------------------
  const int nslot = gridDim.x >> 3;                       // workgroups per XCD (gridDim.x is a multiple of 8)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q = p.items >> 3, r = p.items & 7;
  const int run0 = xcd * q + min(xcd, r), runlen = q + (xcd < r ? 1 : 0);
  if (slot >= runlen) return;
  const int n_items = (runlen - slot + nslot - 1) / nslot;
  const int tiles = p.tiles_m * p.tiles_n;

  auto coords = [&](int w, int& bz, int& bm, int& bn) __attribute__((always_inline)) {
    bz = w / tiles;
    const int t = w - bz * tiles;
    const int g = p.group_n, per_group = g * p.tiles_m;   // grouped rasterisation inside a position (as igemm_kernel)
    const int grp = t / per_group, rr = t - grp * per_group, n0 = grp * g;
    const int width = min(g, p.tiles_n - n0);
    const int tm = rr / width;
    bm = tm * BM;
    bn = (n0 + rr - tm * width) * BN;
  };

  // ---- loader state (switched per item) ----------------------------------------------------------------------------------
  // K-inner image: thread -> (k quad kq, rows r0 + 32 i).   M/N-inner image: thread -> (column quad c4, k rows kr + 16 i).
  const int kq = tid & 7, r0 = tid >> 3;
  const int a_c4 = tid % A_C4, a_kr = tid / A_C4, b_c4 = tid % B_C4, b_kr = tid / B_C4;
  __amdgpu_buffer_rsrc_t ra, rb;
  int a_off[A_LD], b_off[B_LD];
  bool a_ok[A_LD], b_ok[B_LD];
  int l_item = -1, l_t = 0;            // the item being fetched, and its next K-tile

  auto next_item = [&]() __attribute__((always_inline)) {
    ++l_item;
    l_t = 0;
    if (l_item >= n_items) {           // past the last item: the pipeline's look-ahead fetches zeros
#pragma unroll
      for (int i = 0; i < A_LD; ++i) a_ok[i] = false;
#pragma unroll
      for (int i = 0; i < B_LD; ++i) b_ok[i] = false;
      return;
    }
    int bz, bm, bn;
    coords(run0 + slot + l_item * nslot, bz, bm, bn);
    ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a + (int64_t)bz * p.sa), (short)0, (int)p.a_bytes, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b + (int64_t)bz * p.sb), (short)0, (int)p.b_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      if (A_KIN) {
        const int m = bm + r0 + RPP * i;
        a_ok[i] = m < p.M;
        a_off[i] = m * p.K + kq * 4;
      } else {
        const int m = bm + a_c4 * 4;
        a_ok[i] = m < p.M;
        a_off[i] = (a_kr + A_KR * i) * p.M + m;
      }
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      if (B_KIN) {
        const int n = bn + r0 + RPP * i;
        b_ok[i] = n < p.N;
        b_off[i] = n * p.K + kq * 4;
      } else {
        const int n = bn + b_c4 * 4;
        b_ok[i] = n < p.N;
        b_off[i] = (b_kr + B_KR * i) * p.N + n;
      }
    }
  };

  float4 areg[A_LD], breg[B_LD];
  auto load_into = [&](float4* ar, float4* br) __attribute__((always_inline)) {   // the next K-tile of the item sequence (next_item() is the K loop's business)
    const int k0 = l_t * BK;
    ++l_t;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      if (A_KIN) ar[i] = bl4(ra, a_off[i] + k0, a_ok[i] & (k0 + kq * 4 < p.K));
      else ar[i] = bl4(ra, a_off[i] + k0 * p.M, a_ok[i] & (k0 + a_kr + A_KR * i < p.K));
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      if (B_KIN) br[i] = bl4(rb, b_off[i] + k0, b_ok[i] & (k0 + kq * 4 < p.K));
      else br[i] = bl4(rb, b_off[i] + k0 * p.N, b_ok[i] & (k0 + b_kr + B_KR * i < p.K));
    }
  };
  auto load_tile = [&]() __attribute__((always_inline)) { load_into(areg, breg); };
  auto store_from = [&](int buf, const float4* ar, const float4* br) __attribute__((always_inline)) {
    float* as = As + buf * S::A_ELEMS;
    float* bs = Bs + buf * S::B_ELEMS;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      if (A_KIN) *reinterpret_cast<float4*>(&as[(r0 + RPP * i) * KSTRIDE + kq * 4]) = ar[i];
      else *reinterpret_cast<float4*>(&as[(a_kr + A_KR * i) * BM + a_c4 * 4]) = ar[i];
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      if (B_KIN) *reinterpret_cast<float4*>(&bs[(r0 + RPP * i) * KSTRIDE + kq * 4]) = br[i];
      else *reinterpret_cast<float4*>(&bs[(b_kr + B_KR * i) * BN + b_c4 * 4]) = br[i];
    }
  };
  auto store_tile = [&](int buf) __attribute__((always_inline)) { store_from(buf, areg, breg); };

  // fragments of one 8-k chunk: MFMA j consumes k = 8c + j (lanes 0-31) and 8c + 4 + j (lanes 32-63)
  struct Frag { float a[WM][4]; float b[WN][4]; };
  auto read_frag = [&](Frag& f, const float* as, const float* bs, int c) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int row = (wm * WM + i) * 32 + l31;
      if (A_KIN) {
        const float4 v = *reinterpret_cast<const float4*>(&as[row * KSTRIDE + c * 8 + lh * 4]);
        f.a[i][0] = v.x; f.a[i][1] = v.y; f.a[i][2] = v.z; f.a[i][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.a[i][j] = as[(c * 8 + j + 4 * lh) * BM + row];
      }
    }
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      const int col = (wn * WN + i) * 32 + l31;
      if (B_KIN) {
        const float4 v = *reinterpret_cast<const float4*>(&bs[col * KSTRIDE + c * 8 + lh * 4]);
        f.b[i][0] = v.x; f.b[i][1] = v.y; f.b[i][2] = v.z; f.b[i][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[i][j] = bs[(c * 8 + j + 4 * lh) * BN + col];
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][n][e] = 0.f;
  // j outermost: with more than one accumulator consecutive MFMAs never depend on each other; every accumulator still sees
  // its k in the order j = 0..3 of chunk 0, 1, 2, 3 — the 64x64 kernel's chain
  auto mma_frag = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int n = 0; n < WN; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][j], f.b[n][j], acc[i][n], 0, 0, 0);
  };
  constexpr int N_MMA = 4 * WM * WN, N_PIECE = A_LD + B_LD, MMA_PER_PIECE = N_MMA / N_PIECE > 0 ? N_MMA / N_PIECE : 1;

  // One K-tile (igemm_kernel's schedule): registers hold tile t+1, LDS buf[t&1] tile t; the barrier sits mid-tile right after
  // the store of tile t+1, the first two fragment chunks of tile t+1 are fetched while the last two of tile t multiply.
  auto k_tile = [&](int t, Frag& c0, Frag& c1, Frag& n0, Frag& n1) __attribute__((always_inline)) {
    const float* as = As + (t & 1) * S::A_ELEMS;
    const float* bs = Bs + (t & 1) * S::B_ELEMS;
    const float* an = As + ((t + 1) & 1) * S::A_ELEMS;
    const float* bn_ = Bs + ((t + 1) & 1) * S::B_ELEMS;
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(c0);                                   // chunk 0
    __builtin_amdgcn_sched_barrier(0);
    read_frag(c0, as, bs, 2);
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(c1);                                   // chunk 1, with the LDS store of tile t+1 interleaved
    store_tile((t + 1) & 1);
#pragma unroll
    for (int q2 = 0; q2 < N_PIECE; ++q2) {
      __builtin_amdgcn_sched_group_barrier(0x008, MMA_PER_PIECE, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);               // DS write
    }
    __builtin_amdgcn_sched_barrier(0);
    read_frag(c1, as, bs, 3);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_frag(n0, an, bn_, 0);                      // tile t+1, chunks 0 and 1
    read_frag(n1, an, bn_, 1);
    __builtin_amdgcn_sched_barrier(0);
    load_tile();                                    // tile t+2 (the next item's first tiles at an item's end)
    mma_frag(c0);                                   // chunk 2
#pragma unroll
    for (int q2 = 0; q2 < N_PIECE; ++q2) {
      __builtin_amdgcn_sched_group_barrier(0x008, MMA_PER_PIECE, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);               // VMEM read
    }
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(c1);                                   // chunk 3
  };

  next_item();
  {   // the only prologue of the launch: both first K-tiles are requested before the first is waited for
    float4 a0[A_LD], b0[B_LD];
    load_into(a0, b0);
    load_tile();
    store_from(0, a0, b0);
  }
  __syncthreads();
  Frag fa0, fa1, fb0, fb1;
  read_frag(fa0, As, Bs, 0);
  read_frag(fa1, As, Bs, 1);
  for (int it = 0; it < n_items; ++it) {
    for (int t = 0; t < p.ntiles; t += 2) {          // ntiles is even (host side): buffer parity restarts at 0 for every item
      // the loads run two K-tiles ahead: K-tiles t and t+1 are on their way, so from here on the loader belongs to the next
      // item.  The switch sits HERE, between two K-tile pairs, and not inside load_tile: a branch in the middle of a K-tile
      // splits the region the sched_barriers order (measured: -3..5 % on the long-K shapes).
      if (t + 2 == p.ntiles) next_item();
      k_tile(0, fa0, fa1, fb0, fb1);
      k_tile(1, fb0, fb1, fa0, fa1);
    }
    // ---- this item's tile -> memory.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    int bz, bm, bn;
    coords(run0 + slot + it * nslot, bz, bm, bn);
    float* out = p.c + (int64_t)bz * p.sc;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int n = bn + (wn * WN + j) * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = bm + (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (m < p.M && n < p.N) out[(size_t)m * p.N + n] = acc[i][j][e];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// bgemm9_kernel (round 5): the nine position GEMMs of one F(2x2, 2x2) output tile in ONE work item, the output transform in the epilogue.
// The 4x4 stride-2 layers (t2i_winograd.hip: forward = 9 GEMMs, input gradient = 4 phases x 9) write their 9 / 36 product planes M to the
// workspace (2.25 x the output tensor) and read them back in a separate output-transform kernel.  Here a workgroup keeps the nine 64x64
// products of its tile in registers (9 x 16 accumulator registers per lane: two workgroups per CU instead of four), walks the nine K loops
// back to back without leaving the software pipeline (the loader switches operand planes the way bgemm_kernel switches items), and applies
// A^T M A + bias + activation to its own accumulators: lane = output channel, register = tile, so the transform is register arithmetic in
// exactly the order of wino2_output_kernel / wino2b_output_kernel — results are bit-identical to the unfused path.  M never exists.
// Work items: phases x tiles_m x tiles_n (9 x coarser than bgemm_kernel's): the caller takes this path only where that still fills the chip.
//   LAY 0  forward:        A = V[xi] [T][4 Cin] K-inner, B = U[xi] [4 Cin][Cout] N-inner, plane slot = xi
//   LAY 1  input gradient: A = V[slot] [T][Cout] K-inner, B = U[slot] [Cin][Cout] K-inner, slot = ((2 - r) 3 + (2 - c)) 4 + (3 - phase)
// XF (LAY 1 only, round 5): the INPUT transform in the A loader too.  The unfused form writes V = B^T d B of the incoming gradient once per output
// phase (9 x |dy| through the workspace, the largest single stream of this layer class) and reads it back here.  With XF the loader fetches the
// (up to) four dy pixels a V element is made of — V(r,c) = (d[r][c] - d[1][c]) - (d[r][1] - d[1][1]), with the taps of row / column 1 absent for
// r = 1 / c = 1 — straight from dy (which stays in L2 / MALL: 1/9 of V's size) and forms the element in registers, in wino2b_input_kernel's order:
// x - 0 is exact, so the bits are the same.  Padding taps and rows beyond T read zero through the buffer range check, as everywhere here.
template <int LAY, bool XF = false>
__global__ __launch_bounds__(256, 2) void bgemm9_kernel(Bgemm9Params q) {
  static_assert(!XF || LAY == 1, "the in-loader input transform exists for the input-gradient form");
  const BgemmParams& p = q.g;
  using S = BSmem<LAY, 1, 1, 4>;
  constexpr bool B_KIN = S::B_KIN;                          // A is K-inner in both layouts
  constexpr int BM = 64, BN = 64, A_LD = 2, B_LD = 2;
  constexpr int B_C4 = BN / 4, B_KR = 256 / B_C4;
  extern __shared__ __attribute__((aligned(16))) float smem_b[];
  float* As = smem_b;
  float* Bs = smem_b + 2 * S::A_ELEMS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int nslot = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int qq = p.items >> 3, rr_ = p.items & 7;
  const int run0 = xcd * qq + min(xcd, rr_), runlen = qq + (xcd < rr_ ? 1 : 0);
  if (slot >= runlen) return;
  const int n_items = (runlen - slot + nslot - 1) / nslot;
  const int tiles = p.tiles_m * p.tiles_n;

  auto coords = [&](int w, int& phs, int& bm, int& bn) __attribute__((always_inline)) {
    phs = w / tiles;
    const int t = w - phs * tiles;
    const int g = p.group_n, per_group = g * p.tiles_m;
    const int grp = t / per_group, rr = t - grp * per_group, n0 = grp * g;
    const int width = min(g, p.tiles_n - n0);
    const int tm = rr / width;
    bm = tm * BM;
    bn = (n0 + rr - tm * width) * BN;
  };
  auto plane_slot = [&](int pos, int phs) __attribute__((always_inline)) {
    if (LAY == 0) return pos;
    const int r = pos / 3, c = pos - 3 * r;
    return ((2 - r) * 3 + (2 - c)) * 4 + (3 - phs);
  };

  const int kq = tid & 7, r0 = tid >> 3;
  const int b_c4 = tid % B_C4, b_kr = tid / B_C4;
  __amdgpu_buffer_rsrc_t ra, rb;
  int a_off[A_LD], b_off[B_LD];
  bool a_ok[A_LD], b_ok[B_LD];
  unsigned a_win[A_LD];           // XF: which of the 3 x 3 window pixels of the row's tile lie inside the dy map (bit r' * 3 + c')
  int x_tap[4];                   // XF: element offsets of the position's four taps relative to the window origin; x_bit: their window bits, 0 = tap absent
  unsigned x_bit[4];
  int l_item = -1, l_pos = 8, l_t = 0, l_phs = 0, l_bm = 0, l_bn = 0;
  if (XF) ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.dy), (short)0, (int)p.a_bytes, 0x00020000);

  auto next_sub = [&]() __attribute__((always_inline)) {          // the loader moves on to the next (item, position)
    l_t = 0;
    bool new_item = false;
    if (++l_pos == 9) {
      l_pos = 0;
      ++l_item;
      new_item = true;
      if (l_item < n_items) coords(run0 + slot + l_item * nslot, l_phs, l_bm, l_bn);
    }
    if (l_item >= n_items) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) { a_ok[i] = false; a_win[i] = 0u; }
#pragma unroll
      for (int i = 0; i < B_LD; ++i) b_ok[i] = false;
      return;
    }
    const int ps = plane_slot(l_pos, l_phs);
    if (!XF) ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a + (int64_t)ps * p.sa), (short)0, (int)p.a_bytes, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b + (int64_t)ps * p.sb), (short)0, (int)p.b_bytes, 0x00020000);
    if (XF) {
      const int r = l_pos / 3, c = l_pos - 3 * r;
      const int rowpx = q.dWo * p.K;                      // elements per dy row
      x_tap[0] = r * rowpx + c * p.K;  x_bit[0] = 1u << (r * 3 + c);
      x_tap[1] = 1 * rowpx + c * p.K;  x_bit[1] = r != 1 ? 1u << (3 + c) : 0u;
      x_tap[2] = r * rowpx + 1 * p.K;  x_bit[2] = c != 1 ? 1u << (r * 3 + 1) : 0u;
      x_tap[3] = 1 * rowpx + 1 * p.K;  x_bit[3] = (r != 1 && c != 1) ? 1u << 4 : 0u;
      if (new_item) {
        const int ph = l_phs >> 1, pw = l_phs & 1;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
          const int m = l_bm + r0 + 32 * i;
          const int b = m / (q.Th * q.Tw), rem = m - b * (q.Th * q.Tw);
          const int ty = rem / q.Tw, tx = rem - ty * q.Tw;
          const int oh0 = 2 * ty + ph - 1, ow0 = 2 * tx + pw - 1;
          a_off[i] = ((b * q.dHo + oh0) * q.dWo + ow0) * p.K + kq * 4;       // may point before the row / the map: only used under its window bit
          unsigned w = 0u;
#pragma unroll
          for (int rr2 = 0; rr2 < 3; ++rr2)
#pragma unroll
            for (int cc2 = 0; cc2 < 3; ++cc2)
              if ((unsigned)(oh0 + rr2) < (unsigned)q.dHo && (unsigned)(ow0 + cc2) < (unsigned)q.dWo) w |= 1u << (rr2 * 3 + cc2);
          a_win[i] = m < p.M ? w : 0u;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const int m = l_bm + r0 + 32 * i;
        a_ok[i] = m < p.M;
        a_off[i] = m * p.K + kq * 4;
      }
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      if (B_KIN) {
        const int n = l_bn + r0 + 32 * i;
        b_ok[i] = n < p.N;
        b_off[i] = n * p.K + kq * 4;
      } else {
        const int n = l_bn + b_c4 * 4;
        b_ok[i] = n < p.N;
        b_off[i] = (b_kr + B_KR * i) * p.N + n;
      }
    }
  };

  float4 areg[A_LD], breg[B_LD];
  auto load_into = [&](float4* ar, float4* br) __attribute__((always_inline)) {
    const int k0 = l_t * BK;
    ++l_t;
    if (XF) {
      const bool kin = k0 + kq * 4 < p.K;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        const float4 ta = bl4(ra, a_off[i] + x_tap[0] + k0, kin & ((a_win[i] & x_bit[0]) != 0u));
        const float4 tb = bl4(ra, a_off[i] + x_tap[1] + k0, kin & ((a_win[i] & x_bit[1]) != 0u));
        const float4 tc = bl4(ra, a_off[i] + x_tap[2] + k0, kin & ((a_win[i] & x_bit[2]) != 0u));
        const float4 td = bl4(ra, a_off[i] + x_tap[3] + k0, kin & ((a_win[i] & x_bit[3]) != 0u));
        // (d[r][c] - d[1][c]) - (d[r][1] - d[1][1]): wino2b_input_kernel's tt / V arithmetic; absent taps are zeros
        ar[i] = make_float4((ta.x - tb.x) - (tc.x - td.x), (ta.y - tb.y) - (tc.y - td.y), (ta.z - tb.z) - (tc.z - td.z), (ta.w - tb.w) - (tc.w - td.w));
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) ar[i] = bl4(ra, a_off[i] + k0, a_ok[i] & (k0 + kq * 4 < p.K));
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      if (B_KIN) br[i] = bl4(rb, b_off[i] + k0, b_ok[i] & (k0 + kq * 4 < p.K));
      else br[i] = bl4(rb, b_off[i] + k0 * p.N, b_ok[i] & (k0 + b_kr + B_KR * i < p.K));
    }
  };
  auto store_from = [&](int buf, const float4* ar, const float4* br) __attribute__((always_inline)) {
    float* as = As + buf * S::A_ELEMS;
    float* bs = Bs + buf * S::B_ELEMS;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) *reinterpret_cast<float4*>(&as[(r0 + 32 * i) * KSTRIDE + kq * 4]) = ar[i];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      if (B_KIN) *reinterpret_cast<float4*>(&bs[(r0 + 32 * i) * KSTRIDE + kq * 4]) = br[i];
      else *reinterpret_cast<float4*>(&bs[(b_kr + B_KR * i) * BN + b_c4 * 4]) = br[i];
    }
  };

  struct Frag { float a[4]; float b[4]; };
  auto read_frag = [&](Frag& f, const float* as, const float* bs, int c) __attribute__((always_inline)) {
    {
      const int row = wm * 32 + l31;
      const float4 v = *reinterpret_cast<const float4*>(&as[row * KSTRIDE + c * 8 + lh * 4]);
      f.a[0] = v.x; f.a[1] = v.y; f.a[2] = v.z; f.a[3] = v.w;
    }
    {
      const int col = wn * 32 + l31;
      if (B_KIN) {
        const float4 v = *reinterpret_cast<const float4*>(&bs[col * KSTRIDE + c * 8 + lh * 4]);
        f.b[0] = v.x; f.b[1] = v.y; f.b[2] = v.z; f.b[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = bs[(c * 8 + j + 4 * lh) * BN + col];
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // one K-tile on accumulator P (bgemm_kernel's schedule)
  auto k_tile = [&](auto P_, int t, Frag& c0, Frag& c1, Frag& n0, Frag& n1) __attribute__((always_inline)) {
    constexpr int P = decltype(P_)::value;
    const float* as = As + (t & 1) * S::A_ELEMS;
    const float* bs = Bs + (t & 1) * S::B_ELEMS;
    const float* an = As + ((t + 1) & 1) * S::A_ELEMS;
    const float* bn_ = Bs + ((t + 1) & 1) * S::B_ELEMS;
    auto mma = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[j], f.b[j], acc[P], 0, 0, 0);
    };
    __builtin_amdgcn_sched_barrier(0);
    mma(c0);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(c0, as, bs, 2);
    __builtin_amdgcn_sched_barrier(0);
    mma(c1);
    store_from((t + 1) & 1, areg, breg);
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    read_frag(c1, as, bs, 3);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_frag(n0, an, bn_, 0);
    read_frag(n1, an, bn_, 1);
    __builtin_amdgcn_sched_barrier(0);
    load_into(areg, breg);
    mma(c0);
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    mma(c1);
  };

  next_sub();
  {
    float4 a0[A_LD], b0[B_LD];
    load_into(a0, b0);
    load_into(areg, breg);
    store_from(0, a0, b0);
  }
  __syncthreads();
  Frag fa0, fa1, fb0, fb1;
  read_frag(fa0, As, Bs, 0);
  read_frag(fa1, As, Bs, 1);
  auto run_pos = [&](auto P_) __attribute__((always_inline)) {
    for (int t = 0; t < p.ntiles; t += 2) {
      if (t + 2 == p.ntiles) next_sub();               // the loads run two K-tiles ahead: from here on the loader belongs to the next position
      k_tile(P_, 0, fa0, fa1, fb0, fb1);
      k_tile(P_, 1, fb0, fb1, fa0, fa1);
    }
  };
  const int plane_px = q.Th * q.Tw;
  for (int it = 0; it < n_items; ++it) {
    run_pos(std::integral_constant<int, 0>{}); run_pos(std::integral_constant<int, 1>{}); run_pos(std::integral_constant<int, 2>{});
    run_pos(std::integral_constant<int, 3>{}); run_pos(std::integral_constant<int, 4>{}); run_pos(std::integral_constant<int, 5>{});
    run_pos(std::integral_constant<int, 6>{}); run_pos(std::integral_constant<int, 7>{}); run_pos(std::integral_constant<int, 8>{});
    // ---- A^T M A + bias + activation on the accumulators (the arithmetic, and its order, of wino2_output_kernel / wino2b_output_kernel)
    int phs, bm, bn;
    coords(run0 + slot + it * nslot, phs, bm, bn);
    const int ph = phs >> 1, pw = phs & 1;
    const int n = bn + wn * 32 + l31;
    const float bs_ = (q.bias && n < p.N) ? q.bias[n] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = bm + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      float z[2][3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        z[0][c] = acc[0 * 3 + c][e] + acc[1 * 3 + c][e];
        z[1][c] = acc[1 * 3 + c][e] + acc[2 * 3 + c][e];
      }
      if (m < p.M && n < p.N) {
        const int b = m / plane_px, rem = m - b * plane_px;
        const int ty = rem / q.Tw, tx = rem - ty * q.Tw;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const float y0 = apply_act((z[r][0] + z[r][1]) + bs_, q.act, q.alpha);
          const float y1 = apply_act((z[r][1] + z[r][2]) + bs_, q.act, q.alpha);
          const size_t oh = (size_t)q.sr * (2 * ty + r) + ph, ow0 = (size_t)q.sr * (2 * tx) + pw;
          float* o = q.out + (((size_t)b * q.OH + oh) * q.OW + ow0) * p.N + n;
          o[0] = y0;
          o[(size_t)q.sr * p.N] = y1;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  }
}

hipError_t bgemm9_launch(int lay, const Bgemm9Params& q, hipStream_t stream) {
  if (lay != 0 && lay != 1) return hipErrorInvalidValue;
  const int bytes = lay == 0 ? BSmem<0, 1, 1, 4>::BYTES : BSmem<1, 1, 1, 4>::BYTES;
  const int per_xcd = (q.g.items + 7) / 8;
  int nslot = per_xcd < 32 * 2 ? per_xcd : 32 * 2;             // two workgroups per CU (144 accumulator registers per lane)
  if (nslot < 1) nslot = 1;
  if (lay == 0) hipLaunchKernelGGL((bgemm9_kernel<0, false>), dim3(nslot * 8), dim3(256), bytes, stream, q);
  else if (q.dy) hipLaunchKernelGGL((bgemm9_kernel<1, true>), dim3(nslot * 8), dim3(256), bytes, stream, q);
  else hipLaunchKernelGGL((bgemm9_kernel<1, false>), dim3(nslot * 8), dim3(256), bytes, stream, q);
  return hipGetLastError();
}

template <int LAY, int WM, int WN, int NW = 4>
static hipError_t launch_b(const BgemmParams& p, hipStream_t stream) {
  using S = BSmem<LAY, WM, WN, NW>;
  auto k = bgemm_kernel<LAY, WM, WN, NW>;
  static bool attr_done = false;   // benign race: idempotent
  if (!attr_done && S::BYTES > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, S::BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  // workgroups resident per CU: 4 of the 64x64 tile (its registers and LDS), 2 of the larger ones
  const int per_cu = (WM * WN == 1 && NW == 4) ? 4 : 2;
  const int per_xcd = (p.items + 7) / 8;
  int nslot = per_xcd < 32 * per_cu ? per_xcd : 32 * per_cu;   // fewer, evenly loaded workgroups (e.g. 72 x 2 items for 144) lose to the
  if (nslot < 1) nslot = 1;                                    // CU granularity: 72 workgroups on 32 CUs leave some CUs with 3, some with 2
  hipLaunchKernelGGL(k, dim3(nslot * 8), dim3(64 * NW), S::BYTES, stream, p);
  return hipGetLastError();
}

// lay: 0 forward, 1 input gradient, 2 filter gradient (see the header); wm, wn in {1, 2}: tile 64 wm x 64 wn.
// p.items / p.ntiles / tiles_m / tiles_n are filled by the caller FOR THAT TILE.
hipError_t bgemm_launch(int lay, int wm, int wn, const BgemmParams& p, hipStream_t stream) {
  // wm == 4: the 128 x 64 tile on 8 waves (4 x 2 waves of one 32x32 accumulator each)
  if (wm == 4 && wn == 1) {
    if (lay == 0) return launch_b<0, 1, 1, 8>(p, stream);
    if (lay == 1) return launch_b<1, 1, 1, 8>(p, stream);
    if (lay == 2) return launch_b<2, 1, 1, 8>(p, stream);
  }
#define T2I_B(L, a, b) if (lay == L && wm == a && wn == b) return launch_b<L, a, b>(p, stream);
  T2I_B(0, 1, 1) T2I_B(0, 2, 1) T2I_B(0, 1, 2) T2I_B(0, 2, 2)
  T2I_B(1, 1, 1) T2I_B(1, 2, 1) T2I_B(1, 1, 2) T2I_B(1, 2, 2)
  T2I_B(2, 1, 1) T2I_B(2, 2, 1) T2I_B(2, 1, 2) T2I_B(2, 2, 2)
#undef T2I_B
  return hipErrorInvalidValue;
}

}  // namespace t2i
