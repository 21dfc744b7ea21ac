// t2i_internal.h — shared between the kernel translation units of libt2i_hip.so (not part of the C ABI).
#ifndef T2I_INTERNAL_H
#define T2I_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/t2i_hip.h"

namespace t2i {

enum { MODE_FWD = 0, MODE_BWD_DATA = 1, MODE_BWD_FILTER = 2 };

// Division by a launch-constant via multiply-high (valid for 0 <= n < 2^31): the im2col index decode must not cost a
// 30-instruction integer division per gathered element.
struct FastDiv {
  uint32_t mul, shr, add;   // q = (umulhi(n, mul) + (n & add)) >> shr;  d == 1 is {0, 0, ~0}: branch-free
  __host__ void set(uint32_t d) {
    if (d == 1) { mul = 0; shr = 0; add = 0xFFFFFFFFu; return; }
    add = 0;
    uint32_t l = 0;
    while ((1u << l) < d) ++l;                       // ceil(log2 d)
    uint64_t pw = 1ull << (31 + l);
    mul = (uint32_t)((pw + d - 1) / d);
    shr = l - 1;
  }
  __host__ __device__ __forceinline__ int div(int n) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)((__umulhi((uint32_t)n, mul) + ((uint32_t)n & add)) >> shr);
#else
    return (int)(((uint32_t)(((uint64_t)(uint32_t)n * mul) >> 32) + ((uint32_t)n & add)) >> shr);
#endif
  }
};

struct PhaseInfo {      // one stride phase of BWD_DATA: output pixels with (ih % SH, iw % SW) == (ph, pw)
  int32_t ph, pw;       // phase
  int32_t kh0, kw0;     // first filter tap that lands on this phase; taps are kh0 + SH*jh
  int32_t oh_off, ow_off;  // oh = ihq + oh_off - jh
  int32_t nth, ntw;     // taps per phase
  int32_t K;            // nth*ntw*Cout
  FastDiv div_ntw;
};

struct IgemmParams {
  t2i_conv_desc d;
  const float* a;
  const float* b;
  float* c;
  void* c_h;                  // bf16-operand kernels, unsplit: also write the output's bf16 twin here (or NULL)
  const float* bias;
  int32_t act;
  float alpha;
  int32_t M, N, K;            // GEMM extents (BWD_DATA: M per phase; K = max over phases, per-phase K in phase[])
  int32_t tiles_m, tiles_n;
  int32_t group_n;            // rasterisation: N tiles walked per M tile before moving to the next M tile (L2 reuse)
  int32_t splitk, k_per_split;
  size_t out_elems;           // slab stride for split-K
  uint32_t a_bytes, b_bytes;  // extents of the two operand buffers (buffer-load range check: out of range reads 0)
  int32_t howo, hqwq, Wq;     // Ho*Wo; Hq*Wq; Wq  (Hq = ceil(H/SH))
  FastDiv div_howo, div_wo, div_hqwq, div_wq, div_c, div_kw;
  int32_t nphase;
  int32_t walk_doh, walk_db;  // BWD_FILTER pixel walk: advancing the reduction index by one K-tile (32 pixels, Wo | 32)
                              // moves oh by walk_doh (mod Ho, carry into b) and b by walk_db
  int32_t accumulate;         // epilogue adds the existing contents of the output (dw += ...: gradient accumulation)
  int32_t nbatch;             // FWD: independent GEMMs of identical shape in one launch (grid.z; Winograd's 16 tile positions)
  int64_t batch_a, batch_b, batch_c;   // element strides between them
  int32_t batch_lin;          // batched: grid.x = nbatch * tiles, positions assigned to XCDs in contiguous runs (see igemm_kernel)
  int32_t vec_epi;            // bf16-operand kernels: epilogue through LDS with 16-byte stores (store_tile_h) when N % 8 == 0
  float* stats;               // FWD, unsplit: per-M-tile column partials [2][tiles_m][N] (sum, sum of squares) of the output
  PhaseInfo phase[16];
};

__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
  switch (act) {
    case T2I_ACT_LRELU: return v > 0.f ? v : alpha * v;
    case T2I_ACT_RELU: return v > 0.f ? v : 0.f;
    case T2I_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// Winograd F(2x2, 3x3) for 3x3 stride-1 SAME convolutions with many channels (t2i_winograd.hip)
bool winograd_eligible(const t2i_conv_desc& d, bool bwd_data);
bool winograd_filter_eligible(const t2i_conv_desc& d);      // the filter gradient's own size rule on top of winograd_eligible(d, false)
size_t winograd_ws(const t2i_conv_desc& d, bool bwd_data);
int winograd_conv(const t2i_conv_desc& d, bool bwd_data, const float* in, const float* w, const float* bias, float* out, int act,
                  float alpha, void* ws, size_t ws_bytes, hipStream_t stream, float* Vkeep = nullptr);
bool winograd_k4s2_eligible(const t2i_conv_desc& d, bool bwd_data);
size_t winograd_k4s2_filter_grad_ws(const t2i_conv_desc& d);
int winograd_k4s2_filter_grad(const t2i_conv_desc& d, const float* x, const float* dy, float* dw, int accumulate, void* ws, size_t ws_bytes,
                              hipStream_t stream, const float* Vhave = nullptr, int valid_rows = 0x7fffffff);
int filter_cache_enable(int on);
int filter_cache_attach(void* buf, size_t bytes);
void filter_cache_invalidate(const void* p, size_t bytes);
size_t filter_cache_bytes();
int filter_cache_refresh(const void* p, size_t bytes, hipStream_t stream);
int filter_cache_assume(const void* p, size_t bytes, hipStream_t stream);
size_t winograd_k4s2_bwd_ws(const t2i_conv_desc& d);
int winograd_k4s2_bwd_data(const t2i_conv_desc& d, const float* dy, const float* w, const float* bias, float* dx, int act, float alpha,
                           void* ws, size_t ws_bytes, hipStream_t stream);
size_t winograd_k4s2_ws(const t2i_conv_desc& d);
int winograd_k4s2_fwd(const t2i_conv_desc& d, const float* x, const float* w, const float* bias, float* y, int act, float alpha, void* ws,
                      size_t ws_bytes, hipStream_t stream, float* Vkeep = nullptr);
size_t winograd_filter_grad_ws(const t2i_conv_desc& d);
int winograd_filter_grad(const t2i_conv_desc& d, const float* x, const float* dy, float* dw, int accumulate, void* ws, size_t ws_bytes,
                         hipStream_t stream, const float* Vhave = nullptr, int valid_rows = 0x7fffffff, int plane_rows = 0);
int run_batched_gemm(const t2i_conv_desc& gd, int gmode, int nbatch, const float* a, const float* b, float* c, int64_t sa, int64_t sb, int64_t sc,
                     hipStream_t stream, const char* what);

hipError_t igemm_launch(int mode, const IgemmParams& p, int wmt, int wnt, int var, hipStream_t stream);
// bf16 operands in memory (t2i_igemm_h.hip): staging kernels + the GEMM
hipError_t igemm_h_launch(int mode, const IgemmParams& p, int wmt, int wnt, hipStream_t stream);
hipError_t igemm_h_filter_launch(const IgemmParams& p, int wmt, int wnt, hipStream_t stream);
// one launch for a layer's input gradient (or forward-type conv) AND its filter gradient (igemm_pair_kernel)
bool igemm_pair_fusable(const IgemmParams& pb, int wmt_b, int wnt_b);
hipError_t igemm_pair_launch(int mode, const IgemmParams& pa, int wmt, int wnt, const IgemmParams& pb, hipStream_t stream);
hipError_t cast_bf16_launch(const float* x, size_t n, void* y, hipStream_t stream);
hipError_t cast_bf16_any_launch(const float* x, size_t n, void* y, hipStream_t stream);      // any n / alignment
hipError_t cast_f32_launch(const void* x_bf16, size_t n, float* y, hipStream_t stream);    // exact widening (staging copies, bf16 storage)
hipError_t wcast_launch(const float* w, int taps, int Ci, int Co, int transpose, void* out, hipStream_t stream);
// slot of the caller-owned filter-cache arena for (filter, kind) — nullptr when the cache cannot serve it (t2i_winograd.hip)
float* filter_cache_get(const float* w, int kind, int Cin, int Cout, size_t bytes, hipStream_t stream, bool* fill);
// one split-K reduction as splitk_reduce_launch's arguments (the pair launch issues two in one kernel)
struct ReduceJob {
  const float* slabs; const float* bias; float* out; void* out_h; size_t out_elems; float alpha; int splitk, N, act, accumulate, blocks;
};
bool splitk_reduce2_ok(const ReduceJob& a, const ReduceJob& b);
hipError_t splitk_reduce2_launch(ReduceJob a, ReduceJob b, hipStream_t stream);
hipError_t splitk_reduce_launch(const float* slabs, int splitk, size_t out_elems, const float* bias, int N, int act,
                                float alpha, float* out, int accumulate, hipStream_t stream, void* out_h = nullptr, bool* wrote_h = nullptr);

void set_error(const char* fmt, ...);

// Tuning / diagnostic switches, read from the T2I_* environment ONCE (first use) so that the planner never depends on the
// environment at call time: two calls with the same descriptor always take the same path within a process.
// batched plain GEMMs with persistent workgroups (t2i_bgemm.hip): C[z][M,N] = op(A[z]) * op(B[z])
struct BgemmParams {
  const float* a;
  const float* b;
  float* c;
  int32_t M, N, K;
  int32_t tiles_m, tiles_n, group_n;
  int32_t ntiles;             // K-tiles of 32 (even)
  int32_t items;              // nbatch * tiles_m * tiles_n
  int64_t sa, sb, sc;         // element strides between the batch members
  uint32_t a_bytes, b_bytes;  // extents of ONE member's operands (buffer-load range check)
};
hipError_t bgemm_launch(int lay, int wm, int wn, const BgemmParams& p, hipStream_t stream);   // tile 64 wm x 64 wn
// the nine position GEMMs of an F(2x2,2x2) tile + the output transform in one work item (t2i_bgemm.hip: bgemm9_kernel)
struct Bgemm9Params {
  BgemmParams g;              // a = V planes, b = U planes (sa / sb = plane strides), M = tiles T, N, K, tiles_m/n, ntiles, items = phases * tiles_m * tiles_n; c unused
  const float* bias;          // [N] or NULL
  float* out;                 // the conv's output tensor [B, OH, OW, N]
  int32_t OH, OW, Th, Tw;     // output map, tiles per image
  int32_t sr;                 // 1: forward (pixel (2 ty + r, 2 tx + c));  2: input gradient (pixel (2 (2 ty + r) + ph, 2 (2 tx + c) + pw))
  int32_t act;
  float alpha;
  const float* dy;            // LAY 1, optional: the raw incoming gradient [B, dHo, dWo, K] — the loader forms V itself (g.a unused, g.a_bytes = dy's extent)
  int32_t dHo, dWo;
};
hipError_t bgemm9_launch(int lay, const Bgemm9Params& q, hipStream_t stream);

struct Tuning {
  int force_tile, force_splitk, debug_plan, group_n, no_ut, no_thin;
  int winograd, winograd_minc, winograd_maxhw, winograd_k4s2, winograd_k4s2_minc, winograd_k4s2_bwd_minc, winograd_k4s2_bwdf;
  int adam_blocks, max_chain, bf16_operands, cache_refresh, thin_parts, batch_lin, bgemm, winograd_minwork, bf16_dma, hft_boost, hft_ovh, bgemm_tile, bgemm_big_items, vec_epi, pair, pair_cus, pair_max_px, dma_ovh, dma_split_us, tile8_eff, colred_wgs, colred_cap, h_stats, pair_reduce, bf16_waves, bf16_pair_tiles, bn_fuse, wino_fuse, wino_fuse_items, wino_fuse_xf, winograd_k4s2_minwork, winograd_k4s2_minitems;
  double split_cost;
};
const Tuning& tuning();

}  // namespace t2i
#endif
