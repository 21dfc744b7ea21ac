// t2i_capi.hip — the extern "C" surface of libt2i_hip.so (declared in include/t2i_hip.h): argument validation, GEMM
// planning (tile shape, split-K, stride phases, vector-path eligibility) and launches.  No allocation, no
// synchronisation: every entry point only enqueues kernels on the caller's stream, so a whole training step can be
// captured into a hipGraph.
#include <hip/hip_runtime.h>
#include <atomic>
#include <initializer_list>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "t2i_internal.h"

namespace t2i {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// launchers implemented in t2i_aux.hip
size_t col_reduce_ws(int64_t rows, int C);
hipError_t col_reduce_launch(const void*, const void*, const float*, int64_t, int, float*, float*, int, void*, hipStream_t, bool in_bf16 = false);
hipError_t bn_stats_launch(const void*, int64_t, int, float*, float*, const float*, const float*, float, float, float*, float*, float*,
                           float*, float*, float*, void*, hipStream_t, bool x_bf16 = false);
hipError_t bn_stats_tiles_launch(const float*, const float*, int, int, int64_t, int, float*, float*, const float*, const float*, float, float,
                                 float*, float*, float*, float*, float*, float*, hipStream_t);
hipError_t bn_bwd_fused_launch(const void*, const void*, const void*, const float*, const float*, const float*, int64_t, int, int, float,
                               void*, float*, float*, float*, int, void*, hipStream_t, void* dx_h, bool in_bf16);
hipError_t bn_finalize_launch(const float*, const float*, int64_t, int, const float*, const float*, float, float, float*,
                              float*, float*, float*, float*, float*, hipStream_t);
hipError_t bn_apply_launch(const void*, const float*, const float*, int64_t, int, int, float, float*, hipStream_t, void* y_h, bool x_bf16);
size_t bn_grouped_ws(int64_t rows_g, int C, int groups);
hipError_t bn_fwd_grouped_launch(const void*, int64_t, int, int, const float*, const float*, float, float, float*, float*, float*, float*, float*, float*,
                                 int, float, float*, void*, void*, hipStream_t, bool, const float*, const float*, int, int, int, int);
hipError_t bn_bwd_grouped_launch(const void*, const void*, const void*, const float*, const float*, const float*, int64_t, int, int, int, float, void*,
                                 float*, float*, float*, int, void*, hipStream_t, void*, bool);
hipError_t bn_bwd_launch(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                         int64_t, int, float*, float*, float*, float*, int, hipStream_t);
hipError_t ew_launch(int, const void*, const void*, size_t, int, float, float, float*, hipStream_t, void* y_h, bool in_bf16);
hipError_t interp_launch(const float*, const float*, const float*, int, int64_t, float*, hipStream_t);
hipError_t concat_tile_fwd_launch(const void*, const void*, int, int, int, int, void*, hipStream_t, bool bf16);
hipError_t concat_tile_bwd_launch(const void*, int, int, int, int, void*, void*, hipStream_t, bool bf16);
hipError_t transpose_launch(const void*, int, int, int, void*, hipStream_t, bool bf16);
hipError_t gp_slopes_launch(const void*, int, int64_t, float*, hipStream_t, bool bf16);
hipError_t row_scale_launch(const void*, const float*, int, int64_t, void*, hipStream_t, bool bf16 = false, const float* den = nullptr);
hipError_t crop_flip_normalize_launch(const uint8_t*, int, const int32_t*, const int32_t*, const int32_t*, const int32_t*, int, int, float*, hipStream_t);
hipError_t gather_mean_launch(const float*, int, int, const int32_t*, const int32_t*, int, int, float*, hipStream_t);
hipError_t resample2_launch(bool, const float*, int, int, int, int, float, float*, hipStream_t);
hipError_t row_moments_launch(const float*, const float*, int, int64_t, float*, float*, void*, hipStream_t);
size_t row_moments_ws(int B);
hipError_t row_fma2_launch(const float*, const float*, const float*, const float*, const float*, int, int64_t, float*, hipStream_t);
hipError_t wgan_d_head_launch(const float*, const float*, const float*, const float*, int, float, float*, float*, float*, float*, hipStream_t);
hipError_t sigmoid_ce_head_launch(const float* const*, const float*, const float*, float* const*, float* const*, int, float*, hipStream_t);
hipError_t ca_kl_fwd_launch(const float*, const float*, const float*, int, float*, float*, hipStream_t);
hipError_t ca_kl_bwd_launch(const float*, const float*, const float*, const float*, const float*, int, float*, float*, hipStream_t);
hipError_t lerp_dev_launch(const float*, const float*, const float*, int, size_t, float*, hipStream_t);
hipError_t col_reduce_partials_launch(const float*, const float*, int, int, float*, float*, int, hipStream_t);
hipError_t adam_tf_launch(float*, const float*, float*, float*, int64_t, float, const float*, float, float, float, float,
                          hipStream_t);
hipError_t kt_sgd_launch(float*, const float*, float, float, hipStream_t);
hipError_t zero_ranges_launch(float*, const long long*, int, hipStream_t);
hipError_t trunc_normal_launch(float*, size_t, unsigned long long, unsigned long long, float, float, float, float, hipStream_t);
hipError_t act_bwd_colsum_launch(const void*, const void*, const void*, const float*, int64_t, int, int, float, float*, float*,
                                 float*, int, void*, hipStream_t, void* dx_h, bool in_bf16);
// direct kernels for the 3-channel layers (t2i_thin.hip)
bool thin_deconv_eligible(const t2i_conv_desc& d);
hipError_t thin_deconv_launch(const t2i_conv_desc&, const void*, const float*, const float*, float*, int, float, hipStream_t, bool dy_bf16 = false);
bool tiny_bwdw_eligible(const t2i_conv_desc& d);
size_t tiny_bwdw_ws(const t2i_conv_desc& d);
hipError_t tiny_bwdw_launch(const t2i_conv_desc&, const float*, const float*, float*, int, void*, hipStream_t);
bool head_conv_eligible(const t2i_conv_desc& d);
hipError_t head_fwd_launch(const t2i_conv_desc&, const float*, const float*, const float*, float*, int, float, hipStream_t);
hipError_t head_bwd_data_launch(const t2i_conv_desc&, const float*, const float*, float*, hipStream_t);
hipError_t head_bwd_filter_launch(const t2i_conv_desc&, const float*, const float*, float*, int, hipStream_t);
bool stem_fwd_eligible(const t2i_conv_desc& d);
bool stem_bwdf_eligible(const t2i_conv_desc& d);
size_t stem_bwdf_ws(const t2i_conv_desc& d);
hipError_t stem_bwdf_launch(const t2i_conv_desc&, const float*, const void*, float*, int, void*, hipStream_t, bool dy_bf16 = false);
hipError_t stem_fwd_launch(const t2i_conv_desc&, const float*, const float*, const float*, float*, int, float, hipStream_t, void* y_h = nullptr);
bool tiny_conv_eligible(const t2i_conv_desc& d, bool bwd);
hipError_t tiny_conv_launch(const t2i_conv_desc&, bool, const float*, const float*, const float*, float*, int, float, hipStream_t);

static int check(hipError_t e, const char* what) {
  if (e == hipSuccess) return T2I_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return T2I_ERR_LAUNCH;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int validate_desc(const t2i_conv_desc* d) {
  if (!d) { set_error("null descriptor"); return T2I_ERR_INVALID; }
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0 || d->KH <= 0 ||
      d->KW <= 0 || d->SH <= 0 || d->SW <= 0 || d->pad_t < 0 || d->pad_l < 0) {
    set_error("non-positive extent in conv descriptor");
    return T2I_ERR_INVALID;
  }
  if (d->SH > 4 || d->SW > 4) { set_error("stride > 4 unsupported (16 phases max)"); return T2I_ERR_INVALID; }
  if (d->math != T2I_MATH_F32 && d->math != T2I_MATH_BF16) { set_error("unknown math mode %d in conv descriptor", d->math); return T2I_ERR_INVALID; }
  // the last output pixel must start inside the padded image (true for TF SAME / VALID geometries)
  if ((int64_t)(d->Ho - 1) * d->SH - d->pad_t >= d->H || (int64_t)(d->Wo - 1) * d->SW - d->pad_l >= d->W) {
    set_error("output extent inconsistent with input/stride/pad");
    return T2I_ERR_INVALID;
  }
  // buffer loads address with 32-bit byte offsets.  The element counts are formed factor by factor against the limit: four
  // 31-bit extents overflow int64 (found by tools/sanitize_host.sh: UBSan on a hostile descriptor), and every size the planner
  // derives later relies on these products being small.
  const int64_t lim = (1LL << 30) - 16;
  auto fits = [lim](int64_t a, int64_t b, int64_t c, int64_t e) {
    int64_t n = a;
    for (int64_t f : {b, c, e}) {
      if (n > lim / f) return false;      // n * f > lim without forming the product
      n *= f;
    }
    return n <= lim;
  };
  if (!fits(d->B, d->H, d->W, d->Cin) || !fits(d->B, d->Ho, d->Wo, d->Cout) || !fits(d->KH, d->KW, d->Cin, d->Cout)) {
    set_error("tensor exceeds 2^30-16 elements");
    return T2I_ERR_INVALID;
  }
  return T2I_OK;
}

struct Plan {
  int wmt, wnt, splitk, k_per_split, tiles_m, tiles_n;
  size_t ws_bytes;
};

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

static Tuning& tuning_mut() {
  static Tuning t = [] {
    Tuning v;
    v.force_tile = env_int("T2I_FORCE_TILE", 0);            // e.g. 22, 12, 21, 11 (tuning hooks)
    v.force_splitk = env_int("T2I_FORCE_SPLITK", 0);
    v.debug_plan = env_int("T2I_DEBUG_PLAN", 0);
    v.group_n = env_int("T2I_GROUP_N", 8);
    v.bf16_dma = env_int("T2I_BF16_DMA", 1);               // bf16-operand GEMM (fwd / input gradient): operand tiles by LDS DMA (igemm_hd_kernel)
    v.hft_boost = env_int("T2I_HFT_BOOST", 130);           // x0.01: planner efficiency of igemm_hft_kernel's 128x128 tile against igemm_h_filter_kernel's
    v.hft_ovh = env_int("T2I_HFT_OVH", 80);                // x0.1 K-tile steps: its fixed cost per workgroup
    v.bgemm = env_int("T2I_BGEMM", 1);                     // batched (Winograd) fp32 GEMMs: persistent workgroups (t2i_bgemm.hip); 0: one workgroup per tile (igemm_kernel)
    v.bgemm_tile = env_int("T2I_BGEMM_TILE", 0);           // persistent batched GEMM tile: 11 / 21 / 12 / 22 = 64 a x 64 b; 0 = by item count.  Rounds 4-5: the larger tiles lost at every
                                                           // batch size (profiles/r04_bgemm_tiles.txt) and 11 was the default; round 6: the stacked critic pass (4B rows) has launches with
                                                           // >= 8192 64x64 items, where the 128x128 tile (half the operand bytes per multiply-add) wins: 12.73 -> 12.64 ms on the step
    v.bgemm_big_items = env_int("T2I_BGEMM_BIG_ITEMS", 2048);   // ... a larger tile is taken when it still leaves at least this many work items (sweep 1024 .. 6144: profiles/r06_stacked_pass.txt, section 6)
    v.dma_ovh = env_int("T2I_DMA_OVH", 120);               // x0.1 K-tile steps: prologue + epilogue of an igemm_hd_kernel workgroup in the planner's model
    v.dma_split_us = env_int("T2I_DMA_SPLIT_US", 29);      // x0.1 us: fixed cost of its split-K reduction launch
    v.tile8_eff = env_int("T2I_TILE8_EFF", 0);             // x0.01: planner efficiency of the 8-wave 256x128 bf16 tile relative to 128x128 (0: only when forced with force_tile = 42)
    v.colred_wgs = env_int("T2I_COLRED_WGS", 768);         // column reductions, stage 1: workgroups in flight
    v.wino_fuse = env_int("T2I_WINO_FUSE", 1);             // 4x4 stride-2 Winograd: the nine position GEMMs + output transform in one work item (bgemm9_kernel); 0: never, 1: where it still
    v.wino_fuse_items = env_int("T2I_WINO_FUSE_ITEMS", 512);   // has >= wino_fuse_items items (input gradient; forward: 4x that — measured, profiles/r05_winograd_fused.txt), 2: always
    v.wino_fuse_xf = env_int("T2I_WINO_FUSE_XF", 1);       // ... and the input transform of dy in its A loader (input gradient only): no V planes either
    v.bn_fuse = env_int("T2I_BN_FUSE", 1);                 // batch norm: second stage of the statistics in the normalisation's prologue when <= 64 partial rows (t2i_aux.hip)
    v.colred_cap = env_int("T2I_COLRED_CAP", 192);         // ... and the most row chunks (= partials the second stage sums per column)
    v.bf16_waves = env_int("T2I_BF16_WAVES", 8);           // 8: the 128x128 bf16 tile runs on 8 waves of 32x64 (two per SIMD) instead of 4 of 64x64
    v.bf16_pair_tiles = env_int("T2I_BF16_PAIR_TILES", 0); // > 0: 128x128 bf16 launches of at most this many workgroups run the paired-K-tile loop (two K-tiles per barrier pair)
    v.pair_reduce = env_int("T2I_PAIR_REDUCE", 1);         // the split-K reductions of a pair launch's two GEMMs in one launch (splitk_reduce2_kernel)
    v.h_stats = env_int("T2I_H_STATS", 0);                 // bf16-operand forward GEMM: batch-norm statistics from the epilogue (0, default: the batch norm reduces the tensor itself; measured round 4: 10 871 vs 10 827 img/s, the reduce is cheaper than the longer epilogue)
    v.pair_cus = env_int("T2I_PAIR_CUS", 128);             // ... each of the two GEMMs is planned for this many CUs (they share the chip)
    v.pair_max_px = env_int("T2I_PAIR_MAX_PX", 49152);     // ... only for layers with at most this many input pixels (B * H * W)
    v.pair = env_int("T2I_PAIR", 1);                       // t2i_conv2d_bwd_pair: the two GEMMs in one launch where both are bf16-operand DMA kernels (0: two launches)
    v.vec_epi = env_int("T2I_VEC_EPI", 1);                 // bf16-operand GEMMs: epilogue through LDS, 16-byte stores (0: one store per element)
    v.batch_lin = env_int("T2I_BATCH_LIN", 1);             // batched (Winograd) GEMMs: positions in XCD-contiguous runs (0: grid.z = position)
    v.no_ut = env_int("T2I_NO_UT", 0);
    v.no_thin = env_int("T2I_NO_THIN", 0);
    v.winograd = env_int("T2I_WINOGRAD", 1);
    v.winograd_minc = env_int("T2I_WINOGRAD_MINC", 128);
    v.winograd_maxhw = env_int("T2I_WINOGRAD_MAXHW", 1024);
    v.winograd_minwork = env_int("T2I_WINOGRAD_MINWORK", 50000000);   // T * Cin * Cout below which 16 GEMMs + transforms lose to one direct GEMM
    v.winograd_k4s2 = env_int("T2I_WINOGRAD_K4S2", 1);
    v.winograd_k4s2_minc = env_int("T2I_WINOGRAD_K4S2_MINC", 128);
    v.winograd_k4s2_bwd_minc = env_int("T2I_WINOGRAD_K4S2_BWD_MINC", 128);
    v.winograd_k4s2_bwdf = env_int("T2I_WINOGRAD_K4S2_BWDF", 1);
    v.winograd_k4s2_minwork = env_int("T2I_WINOGRAD_K4S2_MINWORK", 160000000);   // T * 4 Cin * Cout below which one direct (split-K) GEMM beats the 9 / 36 position GEMMs + transforms
    v.winograd_k4s2_minitems = env_int("T2I_WINOGRAD_K4S2_MINITEMS", 400);       // ... and the least number of 64x64 work items of the forward form
    v.adam_blocks = env_int("T2I_ADAM_BLOCKS", 2048);
    v.cache_refresh = env_int("T2I_CACHE_REFRESH", 0);     // 1: t2i_adam_tf itself regenerates the cached filter images of its arena (else the caller: t2i_filter_cache_refresh)
    v.thin_parts = env_int("T2I_THIN_PARTS", 2);          // 128 -> 3 k4s2 transposed conv: passes over the channels (Co / parts staged at a time)
    v.bf16_operands = env_int("T2I_BF16_OPERANDS", 1);     // bf16 math: stage bf16 operand copies (t2i_igemm_h.hip) where eligible
    v.max_chain = env_int("T2I_MAX_CHAIN", 8192);           // longest unsplit reduction on the 128x128 tile (see make_plan)
    const char* sc = getenv("T2I_SPLIT_COST");
    v.split_cost = (sc && *sc) ? atof(sc) : 4.0;            // us per extra launch
    return v;
  }();
  return t;
}

const Tuning& tuning() { return tuning_mut(); }

// Tile / split-K choice by a makespan model.  Measured on MI355X: a balanced launch of this kernel sustains ~72% of
// the fp32 matrix peak, but launches whose workgroup count is not a multiple of the 256 CUs lose up to 45% to the
// last partial round (each CU works through its workgroups at a fixed MFMA rate; co-resident workgroups time-share).
// So: for every tile shape and split factor estimate  rounds x (K-tiles per split + fixed overhead) x tile work
// (+ the split-K reduction traffic) and take the cheapest.
// bf16 math: the MFMAs are 16x faster but the operands are still fetched as fp32, so the kernel is bound by the L2 -> LDS
// operand stream (measured ~16 TB/s chip-wide on 128x128 tiles): bytes per FLOP scale with 1/tile edge, which is what
// rel_eff_bf16 encodes (sweep: profiles/r01_bf16_tile_split_sweep.txt; 128x128 wins almost everywhere).
static Plan make_plan(int64_t M, int64_t N, int64_t K, int nphase, size_t out_elems, int split_cap = 32, int math = 0, bool batched = false,
                      int bk = 32, int m_unit = 0,        // m_unit > 0: the M tile must divide it (tiles that stay inside one filter tap)
                      bool dma = false,                   // the bf16-operand kernel that moves its tiles by LDS DMA (igemm_hd_kernel)
                      int cus = 256) {                    // CUs this GEMM can count on: 256, or fewer when a sibling GEMM shares the launch (t2i_conv2d_bwd_pair)
  static const int cand[5][2] = {{2, 2}, {2, 1}, {1, 2}, {1, 1}, {4, 2}};      // {4, 2}: the 8-wave 256x128 tile of igemm_hd8_kernel (dma only)
  // MFMA efficiency relative to the 128x128 tile (re-fitted on the sweep taken with Winograd active: the 128x64 / 64x128
  // shapes lose to 128x128 on the 128-channel direct layers by 7-10 %, and to 64x64 when many workgroups are wanted)
  static const double rel_eff_f32[4] = {1.0, 0.93, 0.93, 0.94};
  static const double rel_eff_bf16[4] = {1.0, 0.74, 0.74, 0.55};
  static const int max_resident_f32[4] = {2, 3, 3, 4};            // co-resident workgroups per CU (LDS / VGPR limited)
  static const int max_resident_bf16[4] = {3, 4, 4, 4};
  // sustained rate of a CU as a function of how many workgroups it time-shares: one workgroup alone leaves the matrix
  // pipe idle during its barrier / LDS-fill phases (measured: 128x128 tiles, 256 vs 512 workgroups: 0.80 vs 0.95)
  static const double share_eff_f32[5] = {0.0, 0.80, 0.95, 1.0, 1.0};
  static const double share_eff_bf16[5] = {0.0, 0.62, 0.86, 1.0, 1.0};
  // igemm_hd_kernel (no staging registers, no ds_write pass): the small tiles lose less to the 128x128 one, a lone workgroup
  // per CU loses less, and unsplit launches of 64x128 tiles beat split 128x128 ones on most B = 64 layers.  Fitted on a sweep of
  // every tile x split over 56 forward / input-gradient cases (profiles/r03_bf16_planner_fit.txt): regret against the per-case
  // optimum 7.4 % with the constants above, 1.0 % with these.
  const double rel_eff_dma[5] = {1.0, 0.82, 0.83, 0.64, tuning().tile8_eff * 0.01};   // [4]: measured by tools/bench_conv.py --math bf16 (profiles/r04_bf16_tile8.txt); 0 = never chosen
  static const int max_resident_dma[5] = {3, 3, 3, 6, 1};
  static const double share_eff_dma[7] = {0.0, 0.75, 1.0, 1.0, 1.0, 1.0, 1.0};
  // the filter gradient (m_unit > 0) runs igemm_hft_kernel on 128x128 tiles when the tile fits a tap (m_unit % 128 == 0) and the
  // register-transposing kernel on the other shapes: the 128x128 entry of rel_eff is raised by hft_boost for it
  const bool hft = math && m_unit > 0 && (m_unit % 128) == 0 && tuning().bf16_dma;
  const double hft_boost = hft ? tuning().hft_boost * 0.01 : 1.0;
  const double* rel_eff = dma ? rel_eff_dma : (math ? rel_eff_bf16 : rel_eff_f32);
  const int* max_resident = dma ? max_resident_dma : (math ? max_resident_bf16 : max_resident_f32);
  const double* share_eff = dma ? share_eff_dma : (math ? share_eff_bf16 : share_eff_f32);
  const double unit_us = math ? 0.135 : 0.52;       // one 64x64x32 tile-step on one CU at the sustained rate
  const double overhead_tiles = dma ? tuning().dma_ovh * 0.1 : (math ? 8.0 : 3.0);   // prologue + epilogue of a workgroup, in K-tile steps
  const int64_t ktiles = (K + bk - 1) / bk;        // K-tiles of the kernel that will run (32; 64 for the bf16-operand kernel)
  const double tile_w = bk / 32.0;                 // ... in units of the 32-wide tile-step the cost constants are quoted for
  int64_t maxsplit = ktiles / 4;
  if (maxsplit < 1) maxsplit = 1;
  if (maxsplit > split_cap) maxsplit = split_cap;
  int ft = tuning().force_tile;
  if (ft == 42 && (!dma || batched || m_unit > 0 || M < 256 || N <= 64)) ft = 22;      // the 8-wave tile exists for the LDS-DMA forward / input-gradient kernel only
  const int fs = tuning().force_splitk;
  Plan best;
  double best_t = 1e300;
  for (int c = 0; c < 5; ++c) {
    const int wmt = cand[c][0], wnt = cand[c][1];
    if (c == 4 && (!dma || batched || m_unit > 0 || (!ft && rel_eff_dma[4] <= 0.0) || M < 256 || N <= 64)) continue;
    if (m_unit > 0 && (m_unit % (64 * wmt)) != 0) continue;
    if (ft) { if (ft != wmt * 10 + wnt) continue; }
    else if (batched && !math) {
      // batched (Winograd) GEMMs: K = channels only, 9-36 GEMMs per launch -> short K loops, many workgroups; 64x64 tiles
      // (4 co-resident workgroups hide each other's prologue/epilogue) won or tied every case of the sweep (up to -13 %)
      if (wmt != 1 || wnt != 1) continue;
    } else {
      if (wmt == 2 && M <= 64) continue;
      if (wnt == 2 && N <= 64) continue;
    }
    const int64_t tm = (M + 64 * wmt - 1) / (64 * wmt), tn = (N + 64 * wnt - 1) / (64 * wnt);
    const int64_t tiles = tm * tn * nphase;
    for (int64_t sk = 1; sk <= maxsplit; ++sk) {
      if (fs > 0 && sk != fs && !(fs > maxsplit && sk == maxsplit)) continue;
      const int64_t per = (ktiles + sk - 1) / sk;
      const int64_t sk_eff = (ktiles + per - 1) / per;
      if (sk_eff != sk) continue;                       // same plan as a smaller sk
      // fp32 accuracy: the MFMA accumulates an output element's K products as ONE sequential fmaf chain, whose rounding
      // error grows ~sqrt(K).  Chains are kept <= max_chain products by splitting K: the slabs are then joined by
      // splitk_reduce in a fixed order (a K-chunked partial sum; a second accumulator set inside the kernel would cost the
      // 128x128 tile its second resident workgroup: 228 of 256 registers are taken).
      if (!math && !batched && per * bk > tuning().max_chain && sk < maxsplit && !fs) continue;
      const int64_t blocks = tiles * sk_eff;
      const int64_t rounds = (blocks + cus - 1) / cus;
      int resident = (int)(rounds < max_resident[c] ? rounds : max_resident[c]);
      if (cus < 256 && resident < 2) resident = 2;      // the sibling's workgroups are co-resident: nobody is alone on a CU
      const double ovh_t = (hft && c == 0) ? tuning().hft_ovh * 0.1 : overhead_tiles;
      const double eff_c = (c == 4 && rel_eff[c] <= 0.0) ? 1.0 : rel_eff[c];       // (forced 8-wave tile without a fitted efficiency)
      double t = (double)rounds * ((double)per * tile_w + ovh_t) * (wmt * wnt) * unit_us /
                 (eff_c * (c == 0 ? hft_boost : 1.0) * (c == 4 ? 1.0 : share_eff[resident]));      // 8 waves: two per SIMD hide each other like two workgroups
      const double split_cost = tuning().split_cost;
      if (sk_eff > 1) t += (dma ? tuning().dma_split_us * 0.1 : (math ? 2.0 : split_cost)) + (double)out_elems * 4.0 * (double)(sk_eff + 1) / (dma ? 5.5e6 : (math ? 6.0e6 : 4.0e6));   // slabs out + in
      if (t < best_t) {
        best_t = t;
        best.wmt = wmt; best.wnt = wnt;
        best.splitk = (int)sk_eff; best.k_per_split = (int)(per * bk);
        best.tiles_m = (int)tm; best.tiles_n = (int)tn;
        best.ws_bytes = sk_eff > 1 ? (size_t)sk_eff * out_elems * sizeof(float) : 0;
      }
    }
  }
  if (tuning().debug_plan)
    fprintf(stderr, "[t2i plan] M=%lld N=%lld K=%lld phases=%d -> tile %dx%d splitk=%d (k/split=%d) model %.1f us\n",
            (long long)M, (long long)N, (long long)K, nphase, 64 * best.wmt, 64 * best.wnt, best.splitk, best.k_per_split, best_t);
  return best;
}

static void fill_common(IgemmParams& p, const t2i_conv_desc* d) {
  memset(&p, 0, sizeof(p));
  p.d = *d;
  p.howo = d->Ho * d->Wo;
  p.div_howo.set(p.howo);
  p.div_wo.set(d->Wo);
  p.div_kw.set(d->KW);
  const int Hq = (d->H + d->SH - 1) / d->SH;
  p.Wq = (d->W + d->SW - 1) / d->SW;
  p.hqwq = Hq * p.Wq;
  p.div_hqwq.set(p.hqwq);
  p.div_wq.set(p.Wq);
  p.nphase = 1;
  p.vec_epi = tuning().vec_epi;
}

// stride phases of the transposed conv: dx pixels with (ih % SH, iw % SW) == (ph, pw) see taps kh = kh0 + SH*jh
static int fill_phases(IgemmParams& p) {
  const t2i_conv_desc& d = p.d;
  int np = 0, kmax = 0;
  for (int ph = 0; ph < d.SH; ++ph)
    for (int pw = 0; pw < d.SW; ++pw) {
      PhaseInfo& pi = p.phase[np++];
      pi.ph = ph; pi.pw = pw;
      pi.kh0 = (ph + d.pad_t) % d.SH;
      pi.kw0 = (pw + d.pad_l) % d.SW;
      pi.nth = pi.kh0 < d.KH ? (d.KH - pi.kh0 + d.SH - 1) / d.SH : 0;
      pi.ntw = pi.kw0 < d.KW ? (d.KW - pi.kw0 + d.SW - 1) / d.SW : 0;
      pi.oh_off = (ph + d.pad_t - pi.kh0) / d.SH;
      pi.ow_off = (pw + d.pad_l - pi.kw0) / d.SW;
      pi.K = pi.nth * pi.ntw * d.Cout;
      pi.div_ntw.set(pi.ntw > 0 ? pi.ntw : 1);
      if (pi.K > kmax) kmax = pi.K;
    }
  p.nphase = np;
  return kmax;
}

// the filter gradient reduces over B*Ho*Wo (up to ~2e5 rows) into a small output: allow deep splits there
static inline int split_cap_for(int mode) { return mode == MODE_BWD_FILTER ? 256 : 32; }

static int run_gemm(int mode, IgemmParams& p, size_t out_elems, int var, float* out, const float* bias, int act,
                    float alpha, void* ws, size_t ws_bytes, hipStream_t stream, const char* what, int accumulate = 0,
                    int* stats_chunks = nullptr, int* stats_tile_rows = nullptr) {
  Plan pl = make_plan(p.M, p.N, p.K, p.nphase, out_elems, split_cap_for(mode), p.d.math);
  if (stats_chunks) {          // epilogue statistics exist only on the unsplit path; the caller falls back otherwise
    if (pl.splitk > 1) { p.stats = nullptr; *stats_chunks = 0; }
    else {
      *stats_chunks = pl.tiles_m;
      if (stats_tile_rows) *stats_tile_rows = pl.wmt * 64;       // the M-tile height of THIS plan (2 x 2 waves of wmt 32-row blocks)
    }
  }
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n;
  { const int g = tuning().group_n; p.group_n = pl.tiles_n < g ? pl.tiles_n : g; if (p.group_n < 1) p.group_n = 1; }
  p.splitk = pl.splitk; p.k_per_split = pl.k_per_split;
  p.out_elems = out_elems;
  if (pl.splitk > 1) {
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) {
      set_error("%s: workspace %zu B < %zu B required (or misaligned)", what, ws_bytes, pl.ws_bytes);
      return T2I_ERR_WORKSPACE;
    }
    p.c = reinterpret_cast<float*>(ws);
    p.bias = nullptr; p.act = T2I_ACT_NONE; p.alpha = 0.f; p.accumulate = 0;
  } else {
    p.c = out; p.bias = bias; p.act = act; p.alpha = alpha; p.accumulate = accumulate;
  }
  if (tuning().no_ut && var == 2) var = 1;
  int rc = check(igemm_launch(mode, p, pl.wmt, pl.wnt, var, stream), what);
  if (rc != T2I_OK) return rc;
  if (pl.splitk > 1)
    rc = check(splitk_reduce_launch(reinterpret_cast<const float*>(ws), pl.splitk, out_elems, bias, p.N, act, alpha, out,
                                    accumulate, stream), what);
  return rc;
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 math with bf16 operands in memory (t2i_igemm_h.hip): forward conv and input gradient / transposed conv whose
// gathered tensor has a multiple of 64 channels.  Workspace = [bf16 copy of the gathered tensor][bf16 filter image,
// unless the filter cache holds it][split-K slabs].
// ------------------------------------------------------------------------------------------------------------------
static inline size_t al256c(size_t n) { return (n + 255) & ~(size_t)255; }

// bytes of the input transform the forward conv of `d` and its filter gradient share (0: they do not both take a Winograd path)
static size_t xform_bytes(const t2i_conv_desc& d) {
  if (winograd_filter_eligible(d)) return (size_t)16 * ((size_t)d.B * (d.H / 2) * (d.W / 2)) * d.Cin * 4;
  if (winograd_k4s2_eligible(d, false) && tuning().winograd_k4s2_bwdf) return (size_t)9 * ((size_t)d.B * (d.Ho / 2) * (d.Wo / 2)) * 4 * d.Cin * 4;
  return 0;
}

static bool h_eligible(const t2i_conv_desc& d, bool bwd_data) {
  const int C = bwd_data ? d.Cout : d.Cin;
  // (taps: the LDS-DMA loader keeps a row's in-image taps in 15 + 15 mask bits; filter image: its poisoned offsets need bytes < 2^30)
  return d.math == T2I_MATH_BF16 && tuning().bf16_operands && C >= 64 && (C % 64) == 0 && d.KH <= 15 && d.KW <= 15 &&
         (size_t)d.KH * d.KW * d.Cin * d.Cout * 2 < ((size_t)1 << 30);
}

static void h_problem(IgemmParams& p, const t2i_conv_desc* d, int mode, size_t* n_in, size_t* out_elems) {
  fill_common(p, d);
  if (mode == MODE_FWD) {
    p.M = d->B * d->Ho * d->Wo; p.N = d->Cout; p.K = d->KH * d->KW * d->Cin;
    p.div_c.set(d->Cin);
    *n_in = (size_t)d->B * d->H * d->W * d->Cin;
    *out_elems = (size_t)p.M * p.N;
  } else {
    p.K = fill_phases(p);
    p.M = d->B * p.hqwq; p.N = d->Cin;
    p.div_c.set(d->Cout);
    *n_in = (size_t)d->B * d->Ho * d->Wo * d->Cout;
    *out_elems = (size_t)d->B * d->H * d->W * d->Cin;
  }
}

static size_t conv_h_ws(const t2i_conv_desc* d, int mode) {
  IgemmParams p;
  size_t n_in, out_elems;
  h_problem(p, d, mode, &n_in, &out_elems);
  const Plan pl = make_plan(p.M, p.N, p.K, p.nphase, out_elems, 32, 1, false, 64, 0, tuning().bf16_dma != 0);
  return al256c(n_in * 2) + al256c((size_t)d->KH * d->KW * d->Cin * d->Cout * 2) + pl.ws_bytes;
}

static std::atomic<long long> g_stat_pair_fused{0};     // t2i_stat("pair_fused"): pairs that went out as ONE launch

// t2i_conv2d_bwd_pair: the two GEMMs of one layer's backward are prepared by conv_h / conv_h_filter as usual (staging, filter
// images, plan, workspace) but their GEMM launches and split-K reductions are handed back in this record instead of being
// issued, so that the caller can put both GEMMs into ONE launch (igemm_pair_kernel) and issue the reductions behind it.
struct PendingGemm {
  bool set = false;
  int mode = 0, wmt = 0, wnt = 0;
  IgemmParams p;
  // the split-K reduction that follows (splitk > 1)
  const float* slabs = nullptr; const float* bias = nullptr; int act = 0; float alpha = 0.f; float* out = nullptr; int accumulate = 0;
  void* out_h = nullptr; int32_t* out_h_written = nullptr;
};

static int finish_pending(const PendingGemm& g, hipStream_t stream, const char* what) {
  if (g.p.splitk <= 1) return T2I_OK;
  bool wrote = false;
  const int rc = check(splitk_reduce_launch(g.slabs, g.p.splitk, g.p.out_elems, g.bias, g.p.N, g.act, g.alpha, g.out, g.accumulate, stream, g.out_h, &wrote), what);
  if (wrote && g.out_h_written) *g.out_h_written = 1;
  return rc;
}

static int conv_h(int mode, const t2i_conv_desc* d, const float* in, const void* in_h, const float* w, const float* bias, float* out, void* out_h,
                  int32_t* out_h_written, int act, float alpha, void* ws, size_t ws_bytes, hipStream_t stream, const char* what,
                  PendingGemm* defer = nullptr, int cus = 256, float* stats = nullptr, int* stats_chunks = nullptr, int* stats_tile_rows = nullptr) {
  IgemmParams p;
  size_t n_in, out_elems;
  h_problem(p, d, mode, &n_in, &out_elems);
  const size_t nw = (size_t)d->KH * d->KW * d->Cin * d->Cout;
  Plan pl = make_plan(p.M, p.N, p.K, p.nphase, out_elems, 32, 1, false, 64, 0, tuning().bf16_dma != 0, cus);
  const size_t off_w = al256c(n_in * 2), off_s = off_w + al256c(nw * 2);
  if (cus != 256 && off_s + pl.ws_bytes > ws_bytes) // the shared-launch plan must fit the workspace sized for the plain one
    pl = make_plan(p.M, p.N, p.K, p.nphase, out_elems, 32, 1, false, 64, 0, tuning().bf16_dma != 0);
  const size_t need = off_s + pl.ws_bytes;
  if (!ws || ws_bytes < need || !aligned16(ws)) {
    set_error("%s: workspace %zu B < %zu B required (or misaligned)", what, ws_bytes, need);
    return T2I_ERR_WORKSPACE;
  }
  char* base = reinterpret_cast<char*>(ws);
  int rc = T2I_OK;
  if (!in_h || !aligned16(in_h)) {                  // no caller-held image of the gathered tensor: stage one
    if (tuning().debug_plan) fprintf(stderr, "[t2i stage] cast_bf16 operand (conv_h): B=%d %dx%dx%d->%d k%d s%d\n", d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->SH);
    rc = check(cast_bf16_launch(in, n_in, base, stream), what);
    if (rc != T2I_OK) return rc;
    in_h = base;
  }
  bool fill = true;
  void* wh = filter_cache_get(w, mode == MODE_FWD ? 4 : 5, d->Cin, d->Cout, nw * 2, stream, &fill);
  if (!wh) { wh = base + off_w; fill = true; }
  if (fill) {
    rc = check(wcast_launch(w, d->KH * d->KW, d->Cin, d->Cout, mode == MODE_FWD ? 1 : 0, wh, stream), what);
    if (rc != T2I_OK) return rc;
  }
  p.a = reinterpret_cast<const float*>(in_h); p.b = reinterpret_cast<const float*>(wh);
  p.a_bytes = (uint32_t)(n_in * 2); p.b_bytes = (uint32_t)(nw * 2);
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n;
  { const int g = tuning().group_n; p.group_n = pl.tiles_n < g ? pl.tiles_n : g; if (p.group_n < 1) p.group_n = 1; }
  p.splitk = pl.splitk; p.k_per_split = pl.k_per_split;
  p.out_elems = out_elems;
  if (pl.splitk > 1) {
    p.c = reinterpret_cast<float*>(base + off_s);
    p.bias = nullptr; p.act = T2I_ACT_NONE; p.alpha = 0.f; p.accumulate = 0;
  } else {
    p.c = out; p.bias = bias; p.act = act; p.alpha = alpha; p.accumulate = 0;
    if (out_h && aligned16(out_h)) { p.c_h = out_h; if (out_h_written) *out_h_written = 1; }
    // batch-norm statistics from the epilogue (tile_stats_h): unsplit forward launches of the LDS-DMA kernel whose epilogue goes through LDS
    if (stats && mode == MODE_FWD && !defer && tuning().h_stats && tuning().bf16_dma && tuning().vec_epi && (p.N % 8) == 0 && (p.out_elems % 4) == 0 &&
        aligned16(p.c) && aligned16(p.c_h) && aligned16(bias)) {
      p.stats = stats;
      if (stats_chunks) *stats_chunks = pl.tiles_m;
      if (stats_tile_rows) *stats_tile_rows = 64 * pl.wmt;
    }
  }
  if (defer) {
    defer->set = true; defer->mode = mode; defer->wmt = pl.wmt; defer->wnt = pl.wnt; defer->p = p;
    defer->slabs = reinterpret_cast<const float*>(base + off_s); defer->bias = bias; defer->act = act; defer->alpha = alpha; defer->out = out;
    defer->accumulate = 0; defer->out_h = (out_h && aligned16(out_h)) ? out_h : nullptr; defer->out_h_written = out_h_written;
    return T2I_OK;
  }
  rc = check(igemm_h_launch(mode, p, pl.wmt, pl.wnt, stream), what);
  if (rc != T2I_OK) return rc;
  if (pl.splitk > 1) {
    bool wrote = false;
    rc = check(splitk_reduce_launch(reinterpret_cast<const float*>(base + off_s), pl.splitk, out_elems, bias, p.N, act, alpha, out, 0, stream,
                                    (out_h && aligned16(out_h)) ? out_h : nullptr, &wrote), what);
    if (wrote && out_h_written) *out_h_written = 1;
  }
  return rc;
}

// filter gradient with bf16 operand copies: workspace = [x_h][dy_h][split-K slabs]
static bool h_filter_eligible(const t2i_conv_desc& d) {
  return d.math == T2I_MATH_BF16 && tuning().bf16_operands && d.Cin >= 64 && (d.Cin % 64) == 0 && (d.Cout % 8) == 0 && (d.Wo % 4) == 0 &&
         (!tuning().force_tile || (d.Cin % (64 * (tuning().force_tile / 10 > 2 ? 2 : tuning().force_tile / 10))) == 0);
}

static Plan h_filter_plan(const t2i_conv_desc* d, int cus = 256) {
  const int64_t M = (int64_t)d->KH * d->KW * d->Cin, N = d->Cout, K = (int64_t)d->B * d->Ho * d->Wo;
  return make_plan(M, N, K, 1, (size_t)(M * N), split_cap_for(MODE_BWD_FILTER), 1, false, 64, d->Cin, false, cus);
}

static size_t conv_h_filter_ws(const t2i_conv_desc* d) {
  const size_t nx = (size_t)d->B * d->H * d->W * d->Cin, ny = (size_t)d->B * d->Ho * d->Wo * d->Cout;
  return al256c(nx * 2) + al256c(ny * 2) + h_filter_plan(d).ws_bytes;
}

static int conv_h_filter(const t2i_conv_desc* d, const float* x, const float* dy, const void* x_h, const void* dy_h, float* dw, int accumulate,
                         void* ws, size_t ws_bytes, hipStream_t stream, PendingGemm* defer = nullptr, int cus = 256) {
  const char* what = "t2i_conv2d_bwd_filter(bf16 operands)";
  IgemmParams p;
  fill_common(p, d);
  const size_t nx = (size_t)d->B * d->H * d->W * d->Cin, ny = (size_t)d->B * d->Ho * d->Wo * d->Cout;
  p.M = d->KH * d->KW * d->Cin; p.N = d->Cout; p.K = d->B * d->Ho * d->Wo;
  p.div_c.set(d->Cin);
  const size_t out_elems = (size_t)p.M * p.N;
  Plan pl = h_filter_plan(d, cus);
  const size_t off_y = al256c(nx * 2), off_s = off_y + al256c(ny * 2);
  if (cus != 256 && off_s + pl.ws_bytes > ws_bytes) pl = h_filter_plan(d);
  const size_t need = off_s + pl.ws_bytes;
  if (!ws || ws_bytes < need || !aligned16(ws)) {
    set_error("%s: workspace %zu B < %zu B required (or misaligned)", what, ws_bytes, need);
    return T2I_ERR_WORKSPACE;
  }
  char* base = reinterpret_cast<char*>(ws);
  int rc = T2I_OK;
  if (!x_h || !aligned16(x_h)) {
    if (tuning().debug_plan) fprintf(stderr, "[t2i stage] cast_bf16 x (filter gradient): B=%d %dx%dx%d->%d k%d s%d\n", d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->SH);
    rc = check(cast_bf16_launch(x, nx, base, stream), what);
    if (rc != T2I_OK) return rc;
    x_h = base;
  }
  if (!dy_h || !aligned16(dy_h)) {
    if (tuning().debug_plan) fprintf(stderr, "[t2i stage] cast_bf16 dy (filter gradient): B=%d %dx%dx%d->%d k%d s%d\n", d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->SH);
    rc = check(cast_bf16_launch(dy, ny, base + off_y, stream), what);
    if (rc != T2I_OK) return rc;
    dy_h = base + off_y;
  }
  p.a = reinterpret_cast<const float*>(x_h); p.b = reinterpret_cast<const float*>(dy_h);
  p.a_bytes = (uint32_t)(nx * 2); p.b_bytes = (uint32_t)(ny * 2);
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n; p.group_n = 1;
  p.splitk = pl.splitk; p.k_per_split = pl.k_per_split;
  p.out_elems = out_elems;
  p.bias = nullptr; p.act = T2I_ACT_NONE; p.alpha = 0.f;
  if (pl.splitk > 1) { p.c = reinterpret_cast<float*>(base + off_s); p.accumulate = 0; }
  else { p.c = dw; p.accumulate = accumulate; }
  if (defer) {
    defer->set = true; defer->mode = MODE_BWD_FILTER; defer->wmt = pl.wmt; defer->wnt = pl.wnt; defer->p = p;
    defer->slabs = reinterpret_cast<const float*>(base + off_s); defer->bias = nullptr; defer->act = T2I_ACT_NONE; defer->alpha = 0.f; defer->out = dw;
    defer->accumulate = accumulate; defer->out_h = nullptr; defer->out_h_written = nullptr;
    return T2I_OK;
  }
  rc = check(igemm_h_filter_launch(p, pl.wmt, pl.wnt, stream), what);
  if (rc != T2I_OK) return rc;
  if (pl.splitk > 1)
    rc = check(splitk_reduce_launch(reinterpret_cast<const float*>(base + off_s), pl.splitk, out_elems, nullptr, p.N, T2I_ACT_NONE, 0.f, dw,
                                    accumulate, stream), what);
  return rc;
}

// nbatch independent GEMMs of one shape in one launch (Winograd's 16 tile positions), as 1x1 convolutions over T "pixels":
//   fwd: c[T, Cout] = a[T, Cin] * b[Cin, Cout]        (MODE_FWD,      b row-major [K][N])
//   bwd: c[T, Cin]  = a[T, Cout] * b[Cin, Cout]^T     (MODE_BWD_DATA, b read as its K-inner image [n][k])
// grid.z = batch.  Never split: nbatch x tiles already fill the chip.
//   flt: c[Cin, Cout] = a[T, Cin]^T * b[T, Cout]     (MODE_BWD_FILTER, reduction over the T "pixels")
int run_batched_gemm(const t2i_conv_desc& gd, int gmode, int nbatch, const float* a, const float* b, float* c, int64_t sa, int64_t sb,
                     int64_t sc, hipStream_t stream, const char* what) {
  {   // fp32, 16-byte operands, an even number of K-tiles: persistent workgroups that never leave the K loop
    const int M = gmode == MODE_BWD_FILTER ? gd.Cin : gd.B;
    const int N = gmode == MODE_BWD_DATA ? gd.Cin : gd.Cout;
    const int K = gmode == MODE_BWD_FILTER ? gd.B : (gmode == MODE_BWD_DATA ? gd.Cout : gd.Cin);
    const int ntiles = (K + 31) / 32;
    const int64_t a_elems = (int64_t)gd.B * (gmode == MODE_BWD_DATA ? gd.Cout : gd.Cin);
    const int64_t b_elems = gmode == MODE_BWD_FILTER ? (int64_t)gd.B * gd.Cout : (int64_t)gd.Cin * gd.Cout;
    if (tuning().bgemm && gd.math != T2I_MATH_BF16 && !tuning().force_tile && (ntiles & 1) == 0 && M % 4 == 0 && N % 4 == 0 && K % 4 == 0 &&
        aligned16(a) && aligned16(b) && sa % 4 == 0 && sb % 4 == 0 && a_elems < (1LL << 30) && b_elems < (1LL << 30)) {
      BgemmParams q;
      memset(&q, 0, sizeof(q));
      q.a = a; q.b = b; q.c = c;
      q.M = M; q.N = N; q.K = K;
      // Tile: the K loop is co-limited by the L2 -> CU operand stream (t2i_bgemm.hip), which a 128x128 tile halves per multiply-add —
      // but its work items are 4x coarser and only two of its workgroups fit a CU, so it is taken when the launch still has
      // enough items to keep 512 resident workgroups evenly busy; 128x64 / 64x128 in between; 64x64 (4 per CU) otherwise.
      int wm = 1, wn = 1;
      {
        auto items_of = [&](int a_, int b_) { return (int64_t)nbatch * ((M + 64 * a_ - 1) / (64 * a_)) * ((N + 64 * b_ - 1) / (64 * b_)); };
        const int ft = tuning().bgemm_tile;
        const int64_t big = tuning().bgemm_big_items;
        if (ft == 22 || ft == 21 || ft == 12 || ft == 11) { wm = ft / 10; wn = ft % 10; }
        else if (ft == 41 && M >= 128) { wm = 4; wn = 1; }          // 128 x 64 on EIGHT waves (round 5): wm = 4 names that kernel, its tile is 128 rows
        else if (M >= 128 && N >= 128 && items_of(2, 2) >= big) { wm = 2; wn = 2; }
        else if (M >= 128 && items_of(2, 1) >= big) { wm = 2; wn = 1; }
        else if (N >= 128 && items_of(1, 2) >= big) { wm = 1; wn = 2; }
      }
      const int bm_rows = wm == 4 ? 128 : 64 * wm;
      q.tiles_m = (M + bm_rows - 1) / bm_rows; q.tiles_n = (N + 64 * wn - 1) / (64 * wn);
      { const int g = tuning().group_n; q.group_n = q.tiles_n < g ? q.tiles_n : g; if (q.group_n < 1) q.group_n = 1; }
      q.ntiles = ntiles;
      const int64_t items = (int64_t)nbatch * q.tiles_m * q.tiles_n;
      if (items < (1LL << 30)) {
        q.items = (int32_t)items;
        q.sa = sa; q.sb = sb; q.sc = sc;
        q.a_bytes = (uint32_t)(a_elems * 4); q.b_bytes = (uint32_t)(b_elems * 4);
        if (tuning().debug_plan)
          fprintf(stderr, "[t2i plan] batched x%d M=%d N=%d K=%d mode %d -> persistent %dx%d, %lld items\n", nbatch, M, N, K, gmode, bm_rows, 64 * wn, (long long)items);
        return check(bgemm_launch(gmode == MODE_FWD ? 0 : (gmode == MODE_BWD_DATA ? 1 : 2), wm, wn, q, stream), what);
      }
    }
  }
  IgemmParams p;
  fill_common(p, &gd);
  p.a = a; p.b = b;
  if (gmode == MODE_BWD_FILTER) {
    p.a_bytes = (uint32_t)((size_t)gd.B * gd.Cin * 4);
    p.b_bytes = (uint32_t)((size_t)gd.B * gd.Cout * 4);
    p.M = gd.Cin; p.N = gd.Cout; p.K = gd.B;
    p.div_c.set(gd.Cin);
    p.walk_db = 32; p.walk_doh = 0;               // 1x1 maps: one K-tile = 32 consecutive "images"
    p.nbatch = nbatch; p.batch_a = sa; p.batch_b = sb; p.batch_c = sc; p.batch_lin = tuning().batch_lin;
    Plan pl = make_plan(p.M, p.N, p.K, nbatch, (size_t)p.M * p.N, 1, gd.math, true);
    p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n;
    { const int g = tuning().group_n; p.group_n = pl.tiles_n < g ? pl.tiles_n : g; if (p.group_n < 1) p.group_n = 1; }
    p.splitk = 1; p.k_per_split = pl.k_per_split;
    p.out_elems = (size_t)p.M * p.N;
    p.c = c; p.bias = nullptr; p.act = T2I_ACT_NONE; p.alpha = 0.f; p.accumulate = 0;
    const bool vec = (gd.Cin % 4 == 0) && (gd.Cout % 4 == 0) && aligned16(a) && aligned16(b);
    return check(igemm_launch(MODE_BWD_FILTER, p, pl.wmt, pl.wnt, vec ? 2 : 0, stream), what);
  }
  const bool bwd = gmode == MODE_BWD_DATA;
  const int K = bwd ? gd.Cout : gd.Cin, N = bwd ? gd.Cin : gd.Cout;
  p.a_bytes = (uint32_t)((size_t)gd.B * K * 4);
  p.b_bytes = (uint32_t)((size_t)gd.Cin * gd.Cout * 4);
  p.M = gd.B; p.N = N; p.K = K;
  if (bwd) {
    const int kmax = fill_phases(p);          // 1x1 stride 1: one phase, K = Cout
    p.K = kmax;
    p.div_c.set(gd.Cout);
  } else {
    p.div_c.set(gd.Cin);
  }
  p.nbatch = nbatch; p.batch_a = sa; p.batch_b = sb; p.batch_c = sc; p.batch_lin = tuning().batch_lin;
  Plan pl = make_plan(p.M, p.N, p.K, nbatch, (size_t)p.M * p.N, 1, gd.math, true);
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n;
  { const int g = tuning().group_n; p.group_n = pl.tiles_n < g ? pl.tiles_n : g; if (p.group_n < 1) p.group_n = 1; }
  p.splitk = 1; p.k_per_split = pl.k_per_split;
  p.out_elems = (size_t)p.M * p.N;
  p.c = c; p.bias = nullptr; p.act = T2I_ACT_NONE; p.alpha = 0.f; p.accumulate = 0;
  const bool vec = (gd.Cin % 4 == 0) && (gd.Cout % 4 == 0) && aligned16(a) && aligned16(b);
  const int var = !vec ? 0 : ((K % 32 == 0) ? 2 : 1);
  return check(igemm_launch(bwd ? MODE_BWD_DATA : MODE_FWD, p, pl.wmt, pl.wnt, var, stream), what);
}

}  // namespace t2i

using namespace t2i;

extern "C" {

int t2i_version(void) { return 9; }

const char* t2i_last_error(void) { return g_err; }

int t2i_device_info(int device, int32_t* cu_count, int32_t* clock_khz, char* arch, size_t arch_len) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return check(e, "t2i_device_info");
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (arch && arch_len) { strncpy(arch, prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
  return T2I_OK;
}

size_t t2i_conv2d_workspace_bytes(const t2i_conv_desc* d) {
  if (validate_desc(d) != T2I_OK) return 0;
  IgemmParams p;
  fill_common(p, d);
  const size_t nx = (size_t)d->B * d->H * d->W * d->Cin, ny = (size_t)d->B * d->Ho * d->Wo * d->Cout,
               nw = (size_t)d->KH * d->KW * d->Cin * d->Cout;
  size_t need = make_plan((int64_t)d->B * d->Ho * d->Wo, d->Cout, (int64_t)d->KH * d->KW * d->Cin, 1, ny, 32, d->math).ws_bytes;
  int kmax = fill_phases(p);
  size_t b = make_plan((int64_t)d->B * p.hqwq, d->Cin, kmax, p.nphase, nx, 32, d->math).ws_bytes;
  if (b > need) need = b;
  b = make_plan((int64_t)d->KH * d->KW * d->Cin, d->Cout, (int64_t)d->B * d->Ho * d->Wo, 1, nw, split_cap_for(MODE_BWD_FILTER), d->math).ws_bytes;
  if (b > need) need = b;
  if (h_eligible(*d, false) && conv_h_ws(d, MODE_FWD) > need) need = conv_h_ws(d, MODE_FWD);
  if (h_eligible(*d, true) && conv_h_ws(d, MODE_BWD_DATA) > need) need = conv_h_ws(d, MODE_BWD_DATA);
  if (h_filter_eligible(*d) && conv_h_filter_ws(d) > need) need = conv_h_filter_ws(d);
  if (tiny_bwdw_eligible(*d) && tiny_bwdw_ws(*d) > need) need = tiny_bwdw_ws(*d);
  if (stem_bwdf_eligible(*d) && stem_bwdf_ws(*d) > need) need = stem_bwdf_ws(*d);
  if (winograd_eligible(*d, false) && winograd_ws(*d, false) > need) need = winograd_ws(*d, false);
  if (winograd_eligible(*d, true) && winograd_ws(*d, true) > need) need = winograd_ws(*d, true);
  if (winograd_filter_eligible(*d) && winograd_filter_grad_ws(*d) > need) need = winograd_filter_grad_ws(*d);
  if (winograd_k4s2_eligible(*d, false) && winograd_k4s2_ws(*d) > need) need = winograd_k4s2_ws(*d);
  if (winograd_k4s2_eligible(*d, true) && winograd_k4s2_bwd_ws(*d) > need) need = winograd_k4s2_bwd_ws(*d);
  if (winograd_k4s2_eligible(*d, false) && winograd_k4s2_filter_grad_ws(*d) > need) need = winograd_k4s2_filter_grad_ws(*d);
  // bf16 storage (t2i_conv_opts.in_dtype / out_dtype): a call whose path has no bf16-tensor loader (thin / head / generic kernels)
  // stages fp32 copies of its two activation tensors in front of the path's own workspace
  // — room for them only where such a call can happen: a descriptor whose three primitives all have bf16-tensor loaders
  // (every 64-channel-multiple layer: the bulk of the bytes) never stages
  if (d->math == T2I_MATH_BF16 && !(h_eligible(*d, false) && h_eligible(*d, true) && h_filter_eligible(*d) && !head_conv_eligible(*d)))
    need += al256c(nx * 4) + al256c(ny * 4) + 512;
  return need;
}

size_t t2i_conv2d_stats_bytes(const t2i_conv_desc* d) {
  if (validate_desc(d) != T2I_OK) return 0;
  const size_t tiles = ((size_t)d->B * d->Ho * d->Wo + 63) / 64;        // smallest M tile
  return tiles * 2 * (size_t)d->Cout * sizeof(float);
}

static int conv2d_fwd_impl(const t2i_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act,
                           float alpha, float* stats, int* stats_chunks, int* stats_tile_rows, t2i_conv_opts* opts, void* ws, size_t ws_bytes,
                           t2i_stream_t stream);

// ---- bf16 storage around the three primitives ---------------------------------------------------------------------------
// opts->in_dtype bit 0 / bit 1: the first / second ACTIVATION operand is a bf16 tensor; opts->out_dtype: the output is.  The
// bf16-operand GEMMs (and the 3 -> 128 stem) read / write such tensors directly; every other path runs on fp32 staging copies
// in the workspace (exact widening in front, one RNE rounding behind) — correct everywhere, fast where it matters.
static inline bool in_h(const t2i_conv_opts* o, int which) { return o && ((o->in_dtype >> which) & 1); }
static inline bool out_h(const t2i_conv_opts* o) { return o && o->out_dtype == T2I_DT_BF16; }
static int storage_check(const t2i_conv_desc* d, const t2i_conv_opts* o, const char* what) {
  if (o && (o->in_dtype & ~3)) { set_error("%s: in_dtype is a mask of bit 0 / bit 1", what); return T2I_ERR_INVALID; }
  if (o && o->out_dtype != T2I_DT_F32 && o->out_dtype != T2I_DT_BF16) { set_error("%s: out_dtype must be T2I_DT_F32 or T2I_DT_BF16", what); return T2I_ERR_INVALID; }
  if (d->math != T2I_MATH_BF16) { set_error("%s: bf16 tensors need t2i_conv_desc.math = T2I_MATH_BF16", what); return T2I_ERR_INVALID; }
  return T2I_OK;
}
struct Staging {           // fp32 copies carved from the front of the workspace
  char* base; size_t bytes, off;
  float* take(size_t elems) {
    const size_t need = al256c(elems * 4);
    if (!base || off + need > bytes) return nullptr;
    float* p = reinterpret_cast<float*>(base + off);
    off += need;
    return p;
  }
};

static int conv2d_fwd_storage(const t2i_conv_desc* d, const void* xv, const float* w, const float* bias, void* yv, int act,
                              float alpha, t2i_conv_opts* opts, void* ws, size_t ws_bytes, t2i_stream_t stream,
                              float* stats, int* stats_chunks, int* stats_tile_rows);

int t2i_conv2d_fwd(const t2i_conv_desc* d, const void* xv, const float* w, const float* bias, void* yv, int act,
                   float alpha, t2i_conv_opts* opts, void* ws, size_t ws_bytes, t2i_stream_t stream) {
  return conv2d_fwd_storage(d, xv, w, bias, yv, act, alpha, opts, ws, ws_bytes, stream, nullptr, nullptr, nullptr);
}

static int conv2d_fwd_storage(const t2i_conv_desc* d, const void* xv, const float* w, const float* bias, void* yv, int act,
                              float alpha, t2i_conv_opts* opts, void* ws, size_t ws_bytes, t2i_stream_t stream,
                              float* stats, int* stats_chunks, int* stats_tile_rows) {
  const bool xh = in_h(opts, 0), yh = out_h(opts);
  if (!xh && !yh)
    return conv2d_fwd_impl(d, reinterpret_cast<const float*>(xv), w, bias, reinterpret_cast<float*>(yv), act, alpha, nullptr, nullptr, nullptr, opts,
                           ws, ws_bytes, stream);
  int rc = validate_desc(d);
  if (rc) return rc;
  if ((rc = storage_check(d, opts, "t2i_conv2d_fwd"))) return rc;
  if (!xv || !w || !yv) { set_error("t2i_conv2d_fwd: null tensor"); return T2I_ERR_INVALID; }
  opts->out_image_written = 0; opts->xform_kept = 0;
  if (!tuning().no_thin && stem_fwd_eligible(*d) && aligned16(w) && aligned16(yv) && (!bias || aligned16(bias)) && !xh && yh)   // fp32 image in, bf16 activation out (vector epilogue: 16-byte stores, float4 bias loads)
    return check(stem_fwd_launch(*d, reinterpret_cast<const float*>(xv), w, bias, nullptr, act, alpha, (hipStream_t)stream, yv), "t2i_conv2d_fwd(stem)");
  const bool head = !tuning().no_thin && head_conv_eligible(*d);
  if (!head && h_eligible(*d, false) && (d->Cout % 4) == 0 && aligned16(xv) && aligned16(w) && aligned16(yv))
    return conv_h(MODE_FWD, d, xh ? nullptr : reinterpret_cast<const float*>(xv), xh ? xv : opts->a_image, w, bias,
                  yh ? nullptr : reinterpret_cast<float*>(yv), yh ? yv : opts->out_image, yh ? nullptr : &opts->out_image_written, act, alpha, ws,
                  ws_bytes, (hipStream_t)stream, "t2i_conv2d_fwd(bf16 operands)", nullptr, 256, stats, stats_chunks, stats_tile_rows);
  const size_t nx = (size_t)d->B * d->H * d->W * d->Cin, ny = (size_t)d->B * d->Ho * d->Wo * d->Cout;
  Staging st{reinterpret_cast<char*>(ws), ws_bytes, 0};
  const float* x32 = reinterpret_cast<const float*>(xv);
  float* y32 = reinterpret_cast<float*>(yv);
  if (xh) {
    float* t = st.take(nx);
    if (!t) { set_error("t2i_conv2d_fwd: workspace too small for the fp32 staging copy"); return T2I_ERR_WORKSPACE; }
    if (tuning().debug_plan) fprintf(stderr, "[t2i stage] cast_f32 t2i_conv2d_fwd(stage): B=%d %dx%dx%d->%d k%d s%d\n", d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->SH);
    if ((rc = check(cast_f32_launch(xv, nx, t, (hipStream_t)stream), "t2i_conv2d_fwd(stage)"))) return rc;
    x32 = t;
  }
  if (yh && !(y32 = st.take(ny))) { set_error("t2i_conv2d_fwd: workspace too small for the fp32 staging copy"); return T2I_ERR_WORKSPACE; }
  t2i_conv_opts o2 = *opts;
  o2.in_dtype = 0; o2.out_dtype = T2I_DT_F32; o2.a_image = nullptr; o2.out_image = nullptr;
  rc = conv2d_fwd_impl(d, x32, w, bias, y32, act, alpha, nullptr, nullptr, nullptr, &o2, st.base + st.off, ws_bytes - st.off, stream);
  if (rc == T2I_OK && yh) rc = check(cast_bf16_any_launch(y32, ny, yv, (hipStream_t)stream), "t2i_conv2d_fwd(unstage)");
  return rc;
}

int t2i_conv2d_fwd_stats(const t2i_conv_desc* d, const void* x, const float* w, const float* bias, void* y, int act,
                         float alpha, float* stats, size_t stats_bytes, int32_t* chunks, int32_t* tile_rows, t2i_conv_opts* opts,
                         void* ws, size_t ws_bytes, t2i_stream_t stream) {
  if (!stats || !chunks || !tile_rows || stats_bytes < t2i_conv2d_stats_bytes(d)) { set_error("t2i_conv2d_fwd_stats: stats buffer missing or too small"); return T2I_ERR_INVALID; }
  if (in_h(opts, 0) || out_h(opts)) {         // bf16 storage: statistics from the bf16-operand kernel's epilogue where the launch is unsplit (round 4)
    *chunks = 0; *tile_rows = 0;
    int c = 0, tr = 0;
    const int rc = conv2d_fwd_storage(d, x, w, bias, y, act, alpha, opts, ws, ws_bytes, stream, stats, &c, &tr);
    *chunks = c; *tile_rows = tr;
    return rc;
  }
  int c = 0, tr = 0;          // number of M-tiles and their height, reported by the plan that launched (run_gemm)
  const int rc = conv2d_fwd_impl(d, reinterpret_cast<const float*>(x), w, bias, reinterpret_cast<float*>(y), act, alpha, stats, &c, &tr, opts, ws, ws_bytes, stream);
  *chunks = c;
  *tile_rows = tr;
  return rc;
}

static int conv2d_fwd_impl(const t2i_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act,
                           float alpha, float* stats, int* stats_chunks, int* stats_tile_rows, t2i_conv_opts* opts, void* ws, size_t ws_bytes,
                           t2i_stream_t stream) {
  const void* x_h = opts ? opts->a_image : nullptr;
  void* y_h = opts ? opts->out_image : nullptr;
  int32_t scratch_flag = 0;
  int32_t* y_h_written = opts ? &opts->out_image_written : &scratch_flag;
  if (opts) { opts->out_image_written = 0; opts->xform_kept = 0; }
  int rc = validate_desc(d);
  if (rc) return rc;
  if (!x || !w || !y) { set_error("t2i_conv2d_fwd: null tensor"); return T2I_ERR_INVALID; }
  float* vkeep = (opts && opts->xform_mode == T2I_XFORM_KEEP && opts->xform && aligned16(opts->xform) && xform_bytes(*d) &&
                  opts->xform_bytes >= xform_bytes(*d)) ? reinterpret_cast<float*>(opts->xform) : nullptr;
  if (stats_chunks) *stats_chunks = 0;
  if (stats_tile_rows) *stats_tile_rows = 0;
  if (!tuning().no_thin) {
    if (head_conv_eligible(*d))
      return check(head_fwd_launch(*d, x, w, bias, y, act, alpha, (hipStream_t)stream), "t2i_conv2d_fwd(head)");
    if (tiny_conv_eligible(*d, false))
      return check(tiny_conv_launch(*d, false, x, w, bias, y, act, alpha, (hipStream_t)stream), "t2i_conv2d_fwd(tiny)");
    if (stem_fwd_eligible(*d) && aligned16(w) && aligned16(y) && (!y_h || aligned16(y_h)) && (!bias || aligned16(bias))) {   // (the LDS epilogue stores 16 bytes per lane)
      if (y_h) *y_h_written = 1;
      return check(stem_fwd_launch(*d, x, w, bias, y, act, alpha, (hipStream_t)stream, y_h), "t2i_conv2d_fwd(stem)");
    }
  }
  if (winograd_eligible(*d, false) && aligned16(x) && aligned16(w) && aligned16(y) && (!bias || aligned16(bias)))
  {
    if (opts) opts->xform_kept = vkeep ? 1 : 0;
    return winograd_conv(*d, false, x, w, bias, y, act, alpha, ws, ws_bytes, (hipStream_t)stream, vkeep);
  }
  if (winograd_k4s2_eligible(*d, false) && aligned16(x) && aligned16(w) && aligned16(y) && (!bias || aligned16(bias)))
  {
    if (opts) opts->xform_kept = vkeep ? 1 : 0;
    return winograd_k4s2_fwd(*d, x, w, bias, y, act, alpha, ws, ws_bytes, (hipStream_t)stream, vkeep);
  }
  if (h_eligible(*d, false) && aligned16(x) && aligned16(w))
    return conv_h(MODE_FWD, d, x, x_h, w, bias, y, y_h, y_h_written, act, alpha, ws, ws_bytes, (hipStream_t)stream, "t2i_conv2d_fwd(bf16 operands)");
  IgemmParams p;
  fill_common(p, d);
  p.a = x; p.b = w;
  p.a_bytes = (uint32_t)((size_t)d->B * d->H * d->W * d->Cin * 4);
  p.b_bytes = (uint32_t)((size_t)d->KH * d->KW * d->Cin * d->Cout * 4);
  p.M = d->B * d->Ho * d->Wo; p.N = d->Cout; p.K = d->KH * d->KW * d->Cin;
  p.div_c.set(d->Cin);
  const bool vec = (d->Cin % 4 == 0) && (d->Cout % 4 == 0) && aligned16(x) && aligned16(w);
  const int var = !vec ? 0 : ((d->Cin % 32 == 0) ? 2 : 1);     // 2: a 32-wide K-tile never straddles a filter tap
  p.stats = stats;
  return run_gemm(MODE_FWD, p, (size_t)p.M * p.N, var, y, bias, act, alpha, ws, ws_bytes, (hipStream_t)stream,
                  "t2i_conv2d_fwd", 0, stats_chunks, stats_tile_rows);
}

static int conv2d_bwd_data_impl(const t2i_conv_desc* d, const float* dy, const float* w, const float* bias, float* dx, int act,
                                float alpha, t2i_conv_opts* opts, void* ws, size_t ws_bytes, t2i_stream_t stream);

int t2i_conv2d_bwd_data(const t2i_conv_desc* d, const void* dyv, const float* w, const float* bias, void* dxv, int act,
                        float alpha, t2i_conv_opts* opts, void* ws, size_t ws_bytes, t2i_stream_t stream) {
  const bool gh = in_h(opts, 0), xh = out_h(opts);
  if (!gh && !xh)
    return conv2d_bwd_data_impl(d, reinterpret_cast<const float*>(dyv), w, bias, reinterpret_cast<float*>(dxv), act, alpha, opts, ws, ws_bytes, stream);
  int rc = validate_desc(d);
  if (rc) return rc;
  if ((rc = storage_check(d, opts, "t2i_conv2d_bwd_data"))) return rc;
  if (!dyv || !w || !dxv) { set_error("t2i_conv2d_bwd_data: null tensor"); return T2I_ERR_INVALID; }
  opts->out_image_written = 0; opts->xform_kept = 0;
  if (!tuning().no_thin && thin_deconv_eligible(*d) && gh && !xh && aligned16(dyv) && aligned16(w) &&
      !(head_conv_eligible(*d) && !bias && act == T2I_ACT_NONE) && !tiny_conv_eligible(*d, true))      // 128 -> 3: bf16 activation in, fp32 image side out
    return check(thin_deconv_launch(*d, dyv, w, bias, reinterpret_cast<float*>(dxv), act, alpha, (hipStream_t)stream, true), "t2i_conv2d_bwd_data(thin)");
  const bool direct = !tuning().no_thin && ((head_conv_eligible(*d) && !bias && act == T2I_ACT_NONE) || tiny_conv_eligible(*d, true) || thin_deconv_eligible(*d));
  if (!direct && h_eligible(*d, true) && (d->Cin % 4) == 0 && aligned16(dyv) && aligned16(w) && aligned16(dxv))
    return conv_h(MODE_BWD_DATA, d, gh ? nullptr : reinterpret_cast<const float*>(dyv), gh ? dyv : opts->a_image, w, bias,
                  xh ? nullptr : reinterpret_cast<float*>(dxv), xh ? dxv : opts->out_image, xh ? nullptr : &opts->out_image_written, act, alpha, ws,
                  ws_bytes, (hipStream_t)stream, "t2i_conv2d_bwd_data(bf16 operands)");
  const size_t nx = (size_t)d->B * d->H * d->W * d->Cin, ny = (size_t)d->B * d->Ho * d->Wo * d->Cout;
  Staging st{reinterpret_cast<char*>(ws), ws_bytes, 0};
  const float* dy32 = reinterpret_cast<const float*>(dyv);
  float* dx32 = reinterpret_cast<float*>(dxv);
  if (gh) {
    float* t = st.take(ny);
    if (!t) { set_error("t2i_conv2d_bwd_data: workspace too small for the fp32 staging copy"); return T2I_ERR_WORKSPACE; }
    if (tuning().debug_plan) fprintf(stderr, "[t2i stage] cast_f32 t2i_conv2d_bwd_data(stage): B=%d %dx%dx%d->%d k%d s%d\n", d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->SH);
    if ((rc = check(cast_f32_launch(dyv, ny, t, (hipStream_t)stream), "t2i_conv2d_bwd_data(stage)"))) return rc;
    dy32 = t;
  }
  if (xh && !(dx32 = st.take(nx))) { set_error("t2i_conv2d_bwd_data: workspace too small for the fp32 staging copy"); return T2I_ERR_WORKSPACE; }
  t2i_conv_opts o2 = *opts;
  o2.in_dtype = 0; o2.out_dtype = T2I_DT_F32; o2.a_image = nullptr; o2.out_image = nullptr;
  rc = conv2d_bwd_data_impl(d, dy32, w, bias, dx32, act, alpha, &o2, st.base + st.off, ws_bytes - st.off, stream);
  if (rc == T2I_OK && xh) rc = check(cast_bf16_any_launch(dx32, nx, dxv, (hipStream_t)stream), "t2i_conv2d_bwd_data(unstage)");
  return rc;
}

static int conv2d_bwd_data_impl(const t2i_conv_desc* d, const float* dy, const float* w, const float* bias, float* dx, int act,
                                float alpha, t2i_conv_opts* opts, void* ws, size_t ws_bytes, t2i_stream_t stream) {
  const void* dy_h = opts ? opts->a_image : nullptr;
  void* dx_h = opts ? opts->out_image : nullptr;
  if (opts) { opts->out_image_written = 0; opts->xform_kept = 0; }
  int rc = validate_desc(d);
  if (rc) return rc;
  if (!dy || !w || !dx) { set_error("t2i_conv2d_bwd_data: null tensor"); return T2I_ERR_INVALID; }
  if (!tuning().no_thin) {
    if (head_conv_eligible(*d) && !bias && act == T2I_ACT_NONE)
      return check(head_bwd_data_launch(*d, dy, w, dx, (hipStream_t)stream), "t2i_conv2d_bwd_data(head)");
    if (tiny_conv_eligible(*d, true))
      return check(tiny_conv_launch(*d, true, dy, w, bias, dx, act, alpha, (hipStream_t)stream), "t2i_conv2d_bwd_data(tiny)");
    if (thin_deconv_eligible(*d) && aligned16(dy) && aligned16(w))
      return check(thin_deconv_launch(*d, dy, w, bias, dx, act, alpha, (hipStream_t)stream), "t2i_conv2d_bwd_data(thin)");
  }
  if (winograd_eligible(*d, true) && aligned16(dy) && aligned16(w) && aligned16(dx) && (!bias || aligned16(bias)))
    return winograd_conv(*d, true, dy, w, bias, dx, act, alpha, ws, ws_bytes, (hipStream_t)stream);
  if (winograd_k4s2_eligible(*d, true) && aligned16(dy) && aligned16(w) && aligned16(dx) && (!bias || aligned16(bias)))
    return winograd_k4s2_bwd_data(*d, dy, w, bias, dx, act, alpha, ws, ws_bytes, (hipStream_t)stream);
  if (h_eligible(*d, true) && aligned16(dy) && aligned16(w))
    return conv_h(MODE_BWD_DATA, d, dy, dy_h, w, bias, dx, dx_h, opts ? &opts->out_image_written : nullptr, act, alpha, ws, ws_bytes, (hipStream_t)stream,
                  "t2i_conv2d_bwd_data(bf16 operands)");
  IgemmParams p;
  fill_common(p, d);
  p.a = dy; p.b = w;
  p.a_bytes = (uint32_t)((size_t)d->B * d->Ho * d->Wo * d->Cout * 4);
  p.b_bytes = (uint32_t)((size_t)d->KH * d->KW * d->Cin * d->Cout * 4);
  p.K = fill_phases(p);
  p.M = d->B * p.hqwq; p.N = d->Cin;
  p.div_c.set(d->Cout);
  const bool vec = (d->Cout % 4 == 0) && aligned16(dy) && aligned16(w);
  const int var = !vec ? 0 : ((d->Cout % 32 == 0) ? 2 : 1);
  return run_gemm(MODE_BWD_DATA, p, (size_t)d->B * d->H * d->W * d->Cin, var, dx, bias, act, alpha, ws, ws_bytes,
                  (hipStream_t)stream, "t2i_conv2d_bwd_data");
}

static int conv2d_bwd_filter_impl(const t2i_conv_desc* d, const float* x, const float* dy, float* dw, int accumulate,
                                  t2i_conv_opts* opts, void* ws, size_t ws_bytes, t2i_stream_t stream);

int t2i_conv2d_bwd_filter(const t2i_conv_desc* d, const void* xv, const void* dyv, float* dw, int accumulate,
                          t2i_conv_opts* opts, void* ws, size_t ws_bytes, t2i_stream_t stream) {
  const bool xh = in_h(opts, 0), gh = in_h(opts, 1);
  if (!xh && !gh)
    return conv2d_bwd_filter_impl(d, reinterpret_cast<const float*>(xv), reinterpret_cast<const float*>(dyv), dw, accumulate, opts, ws, ws_bytes, stream);
  int rc = validate_desc(d);
  if (rc) return rc;
  if ((rc = storage_check(d, opts, "t2i_conv2d_bwd_filter"))) return rc;
  if (!xv || !dyv || !dw) { set_error("t2i_conv2d_bwd_filter: null tensor"); return T2I_ERR_INVALID; }
  opts->out_image_written = 0; opts->xform_kept = 0;
  if (!tuning().no_thin && !head_conv_eligible(*d) && !tiny_bwdw_eligible(*d) && stem_bwdf_eligible(*d) && aligned16(dw) && !xh && gh) {
    const size_t need = stem_bwdf_ws(*d);            // 3 -> 128 stem: fp32 image side x, bf16 activation gradient dy
    if (!ws || ws_bytes < need || !aligned16(ws)) { set_error("t2i_conv2d_bwd_filter: workspace %zu B < %zu B required", ws_bytes, need); return T2I_ERR_WORKSPACE; }
    return check(stem_bwdf_launch(*d, reinterpret_cast<const float*>(xv), dyv, dw, accumulate ? 1 : 0, ws, (hipStream_t)stream, true), "t2i_conv2d_bwd_filter(stem)");
  }
  const bool direct = !tuning().no_thin && (head_conv_eligible(*d) || tiny_bwdw_eligible(*d) || (stem_bwdf_eligible(*d) && aligned16(dw)));
  if (!direct && h_filter_eligible(*d) && aligned16(xv) && aligned16(dyv) && aligned16(dw))
    return conv_h_filter(d, xh ? nullptr : reinterpret_cast<const float*>(xv), gh ? nullptr : reinterpret_cast<const float*>(dyv),
                         xh ? xv : opts->a_image, gh ? dyv : opts->b_image, dw, accumulate ? 1 : 0, ws, ws_bytes, (hipStream_t)stream);
  const size_t nx = (size_t)d->B * d->H * d->W * d->Cin, ny = (size_t)d->B * d->Ho * d->Wo * d->Cout;
  Staging st{reinterpret_cast<char*>(ws), ws_bytes, 0};
  const float* x32 = reinterpret_cast<const float*>(xv);
  const float* dy32 = reinterpret_cast<const float*>(dyv);
  if (xh) {
    float* t = st.take(nx);
    if (!t) { set_error("t2i_conv2d_bwd_filter: workspace too small for the fp32 staging copy"); return T2I_ERR_WORKSPACE; }
    if (tuning().debug_plan) fprintf(stderr, "[t2i stage] cast_f32 t2i_conv2d_bwd_filter(stage): B=%d %dx%dx%d->%d k%d s%d\n", d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->SH);
    if ((rc = check(cast_f32_launch(xv, nx, t, (hipStream_t)stream), "t2i_conv2d_bwd_filter(stage)"))) return rc;
    x32 = t;
  }
  if (gh) {
    float* t = st.take(ny);
    if (!t) { set_error("t2i_conv2d_bwd_filter: workspace too small for the fp32 staging copy"); return T2I_ERR_WORKSPACE; }
    if (tuning().debug_plan) fprintf(stderr, "[t2i stage] cast_f32 t2i_conv2d_bwd_filter(stage): B=%d %dx%dx%d->%d k%d s%d\n", d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->SH);
    if ((rc = check(cast_f32_launch(dyv, ny, t, (hipStream_t)stream), "t2i_conv2d_bwd_filter(stage)"))) return rc;
    dy32 = t;
  }
  t2i_conv_opts o2 = *opts;
  o2.in_dtype = 0; o2.out_dtype = T2I_DT_F32; o2.a_image = nullptr; o2.b_image = nullptr;
  return conv2d_bwd_filter_impl(d, x32, dy32, dw, accumulate, &o2, st.base + st.off, ws_bytes - st.off, stream);
}

// One layer's backward pair on bf16 tensors (round 4): `first` = T2I_PAIR_BWD_DATA: dx = conv^T(g, w) (the input gradient of a
// conv) or T2I_PAIR_FWD: y = conv(g, w) (the input gradient of a transposed conv), together with the filter gradient dw (+)= fx (*) fdy.
// Exactly t2i_conv2d_bwd_data / t2i_conv2d_fwd (no bias, no activation) followed by t2i_conv2d_bwd_filter — same tiles, same
// arithmetic, same bits — except that, where both run on the bf16-operand DMA kernels, the two GEMMs share ONE launch
// (igemm_pair_kernel).  The two calls need their workspaces at the same time: ws1 / ws2 must not overlap.
int t2i_conv2d_bwd_pair(const t2i_conv_desc* d, int first, const void* g, const float* w, void* out1, t2i_conv_opts* opts1,
                        const void* fx, const void* fdy, float* dw, int accumulate, t2i_conv_opts* opts2,
                        void* ws1, size_t ws1_bytes, void* ws2, size_t ws2_bytes, t2i_stream_t stream) {
  if (first != T2I_PAIR_FWD && first != T2I_PAIR_BWD_DATA) { set_error("t2i_conv2d_bwd_pair: first must be T2I_PAIR_FWD or T2I_PAIR_BWD_DATA"); return T2I_ERR_INVALID; }
  int rc = validate_desc(d);
  if (rc) return rc;
  if (!opts1 || !opts2) { set_error("t2i_conv2d_bwd_pair: opts1 / opts2 are required (they carry the tensor dtypes)"); return T2I_ERR_INVALID; }
  const int mode = first == T2I_PAIR_FWD ? MODE_FWD : MODE_BWD_DATA;
  const bool all_h = in_h(opts1, 0) && out_h(opts1) && in_h(opts2, 0) && in_h(opts2, 1);
  const bool thin1 = !tuning().no_thin && (head_conv_eligible(*d) || tiny_conv_eligible(*d, mode == MODE_BWD_DATA) ||
                                           (mode == MODE_FWD ? stem_fwd_eligible(*d) : thin_deconv_eligible(*d)));
  const bool thin2 = !tuning().no_thin && (head_conv_eligible(*d) || tiny_bwdw_eligible(*d) || stem_bwdf_eligible(*d));
  const int Cout1 = mode == MODE_FWD ? d->Cout : d->Cin;
  // Measured (profiles/r04_bf16_pair_launch.txt): sharing the launch pays on the small maps (4x4, 8x8: each GEMM alone is a handful of
  // K-deep tiles per CU plus split-K slabs; planned for half the chip each they need half the slabs and hide each other's stalls:
  // -8..-23 %), and loses on the 16x16 / 32x32 maps, whose GEMMs already fill the chip two workgroups deep on their own.
  const bool small_map = (int64_t)d->B * d->H * d->W <= (int64_t)tuning().pair_max_px;
  const bool fuse = tuning().pair && small_map && all_h && d->math == T2I_MATH_BF16 && !thin1 && !thin2 && h_eligible(*d, mode == MODE_BWD_DATA) && (Cout1 % 4) == 0 &&
                    h_filter_eligible(*d) && g && w && out1 && fx && fdy && dw && aligned16(g) && aligned16(w) && aligned16(out1) && aligned16(fx) &&
                    aligned16(fdy) && aligned16(dw) && ws1 && ws2 &&
                    (reinterpret_cast<char*>(ws1) + ws1_bytes <= reinterpret_cast<char*>(ws2) || reinterpret_cast<char*>(ws2) + ws2_bytes <= reinterpret_cast<char*>(ws1));
  if (fuse) {
    if ((rc = storage_check(d, opts1, "t2i_conv2d_bwd_pair")) || (rc = storage_check(d, opts2, "t2i_conv2d_bwd_pair"))) return rc;
    opts1->out_image_written = 0; opts1->xform_kept = 0; opts2->out_image_written = 0; opts2->xform_kept = 0;
    PendingGemm ga, gb;
    const char* what = "t2i_conv2d_bwd_pair";
    // the filter gradient first, planned for its share of the chip: only igemm_hft_kernel's 128x128 case can share a launch
    int cus = tuning().pair_cus;
    if (cus < 32 || cus > 256) cus = 256;
    rc = conv_h_filter(d, nullptr, nullptr, fx, fdy, dw, accumulate ? 1 : 0, ws2, ws2_bytes, (hipStream_t)stream, &gb, cus);
    if (rc != T2I_OK) return rc;
    bool one = igemm_pair_fusable(gb.p, gb.wmt, gb.wnt);
    if (!one && cus != 256) {                      // two launches after all: each gets the whole chip, plan for that
      cus = 256;
      rc = conv_h_filter(d, nullptr, nullptr, fx, fdy, dw, accumulate ? 1 : 0, ws2, ws2_bytes, (hipStream_t)stream, &gb, cus);
      if (rc != T2I_OK) return rc;
      one = igemm_pair_fusable(gb.p, gb.wmt, gb.wnt);
    }
    rc = conv_h(mode, d, nullptr, g, w, nullptr, nullptr, out1, nullptr, T2I_ACT_NONE, 0.f, ws1, ws1_bytes, (hipStream_t)stream, what, &ga, one ? cus : 256);
    if (rc != T2I_OK) return rc;
    if (one && !(ga.wmt <= 2 && ga.wnt <= 2)) {
      // the shared launch dispatches the {1,2} x {1,2} wave tiles only; the planner may hand the first GEMM the 8-wave (4,2) tile
      // (T2I_TILE8_EFF / force_tile): then two launches, each planned for the whole chip
      one = false;
      rc = conv_h_filter(d, nullptr, nullptr, fx, fdy, dw, accumulate ? 1 : 0, ws2, ws2_bytes, (hipStream_t)stream, &gb, 256);
      if (rc != T2I_OK) return rc;
      rc = conv_h(mode, d, nullptr, g, w, nullptr, nullptr, out1, nullptr, T2I_ACT_NONE, 0.f, ws1, ws1_bytes, (hipStream_t)stream, what, &ga, 256);
      if (rc != T2I_OK) return rc;
    }
    if (one) {
      if (tuning().debug_plan) fprintf(stderr, "[t2i plan] pair: one launch, %d + %d workgroups\n",
                                       ga.p.tiles_m * ga.p.tiles_n * ga.p.splitk * (mode == MODE_BWD_DATA ? ga.p.nphase : 1), gb.p.tiles_m * gb.p.tiles_n * gb.p.splitk);
      rc = check(igemm_pair_launch(mode, ga.p, ga.wmt, ga.wnt, gb.p, (hipStream_t)stream), what);
      if (rc == T2I_OK) ++g_stat_pair_fused;
    } else {
      rc = check(igemm_h_launch(mode, ga.p, ga.wmt, ga.wnt, (hipStream_t)stream), what);
      if (rc == T2I_OK) rc = check(igemm_h_filter_launch(gb.p, gb.wmt, gb.wnt, (hipStream_t)stream), what);
    }
    if (rc != T2I_OK) return rc;
    if (tuning().pair_reduce && ga.p.splitk > 1 && gb.p.splitk > 1) {        // both split: their reductions share a launch too
      auto job = [](const PendingGemm& q) {
        ReduceJob j;
        j.slabs = q.slabs; j.bias = q.bias; j.out = q.out; j.out_h = q.out_h; j.out_elems = q.p.out_elems; j.alpha = q.alpha;
        j.splitk = q.p.splitk; j.N = q.p.N; j.act = q.act; j.accumulate = q.accumulate; j.blocks = 0;
        return j;
      };
      const ReduceJob ja = job(ga), jb = job(gb);
      if (splitk_reduce2_ok(ja, jb)) {
        rc = check(splitk_reduce2_launch(ja, jb, (hipStream_t)stream), what);
        if (rc == T2I_OK) {
          if (ga.out_h && ga.out_h_written) *ga.out_h_written = 1;
          if (gb.out_h && gb.out_h_written) *gb.out_h_written = 1;
        }
        return rc;
      }
    }
    if ((rc = finish_pending(ga, (hipStream_t)stream, what))) return rc;
    return finish_pending(gb, (hipStream_t)stream, what);
  }
  rc = mode == MODE_FWD ? t2i_conv2d_fwd(d, g, w, nullptr, out1, T2I_ACT_NONE, 0.f, opts1, ws1, ws1_bytes, stream)
                        : t2i_conv2d_bwd_data(d, g, w, nullptr, out1, T2I_ACT_NONE, 0.f, opts1, ws1, ws1_bytes, stream);
  if (rc != T2I_OK) return rc;
  return t2i_conv2d_bwd_filter(d, fx, fdy, dw, accumulate, opts2, ws2, ws2_bytes, stream);
}

static int conv2d_bwd_filter_impl(const t2i_conv_desc* d, const float* x, const float* dy, float* dw, int accumulate,
                                  t2i_conv_opts* opts, void* ws, size_t ws_bytes, t2i_stream_t stream) {
  if (opts) { opts->out_image_written = 0; opts->xform_kept = 0; }
  int rc = validate_desc(d);
  if (rc) return rc;
  if (!x || !dy || !dw) { set_error("t2i_conv2d_bwd_filter: null tensor"); return T2I_ERR_INVALID; }
  const float* vhave = (opts && opts->xform_mode == T2I_XFORM_HAVE && opts->xform && aligned16(opts->xform) && xform_bytes(*d) &&
                        opts->xform_bytes >= xform_bytes(*d)) ? reinterpret_cast<const float*>(opts->xform) : nullptr;
  // xform_plane_rows (ABI v9): `xform` is the transform of a LARGER batch (that many images) whose leading B images are this x
  int prows = 0;
  if (vhave && opts->xform_plane_rows != 0) {
    if (opts->xform_plane_rows < d->B) { set_error("t2i_conv2d_bwd_filter: xform_plane_rows < B"); return T2I_ERR_INVALID; }
    prows = opts->xform_plane_rows;
    if (prows > d->B) {
      t2i_conv_desc big = *d; big.B = prows;
      // only the 3x3 Winograd form strides plane by plane; the 4x4 stride-2 form cuts its planes into slabs: it transforms x anew
      if (!winograd_filter_eligible(*d) || validate_desc(&big) != T2I_OK || opts->xform_bytes < xform_bytes(big)) { vhave = nullptr; prows = 0; }
    }
  }
  // xform_valid_rows (ABI v9): the kept transform is current for that many leading images of the batch only; the rest is regenerated
  // from x INTO the caller's buffer (0 or >= B: all of it is current)
  const int vrows = (vhave && opts->xform_valid_rows > 0 && opts->xform_valid_rows < d->B) ? opts->xform_valid_rows : 0x7fffffff;
  if (vhave && opts->xform_valid_rows < 0) { set_error("t2i_conv2d_bwd_filter: xform_valid_rows < 0"); return T2I_ERR_INVALID; }
  if (!tuning().no_thin) {
    if (head_conv_eligible(*d))
      return check(head_bwd_filter_launch(*d, x, dy, dw, accumulate ? 1 : 0, (hipStream_t)stream), "t2i_conv2d_bwd_filter(head)");
    if (tiny_bwdw_eligible(*d)) {
      const size_t need = tiny_bwdw_ws(*d);
      if (!ws || ws_bytes < need || !aligned16(ws)) { set_error("t2i_conv2d_bwd_filter: workspace %zu B < %zu B required", ws_bytes, need); return T2I_ERR_WORKSPACE; }
      return check(tiny_bwdw_launch(*d, x, dy, dw, accumulate ? 1 : 0, ws, (hipStream_t)stream), "t2i_conv2d_bwd_filter(tiny)");
    }
    if (stem_bwdf_eligible(*d) && aligned16(dw)) {
      const size_t need = stem_bwdf_ws(*d);
      if (!ws || ws_bytes < need || !aligned16(ws)) { set_error("t2i_conv2d_bwd_filter: workspace %zu B < %zu B required", ws_bytes, need); return T2I_ERR_WORKSPACE; }
      return check(stem_bwdf_launch(*d, x, dy, dw, accumulate ? 1 : 0, ws, (hipStream_t)stream), "t2i_conv2d_bwd_filter(stem)");
    }
  }
  if (winograd_filter_eligible(*d) && aligned16(x) && aligned16(dy) && aligned16(dw))
    return winograd_filter_grad(*d, x, dy, dw, accumulate ? 1 : 0, ws, ws_bytes, (hipStream_t)stream, vhave, vrows, prows);
  if (winograd_k4s2_eligible(*d, false) && tuning().winograd_k4s2_bwdf && aligned16(x) && aligned16(dy) && aligned16(dw))
    return winograd_k4s2_filter_grad(*d, x, dy, dw, accumulate ? 1 : 0, ws, ws_bytes, (hipStream_t)stream, vhave, vrows);
  if (h_filter_eligible(*d) && aligned16(x) && aligned16(dy) && aligned16(dw))
    return conv_h_filter(d, x, dy, opts ? opts->a_image : nullptr, opts ? opts->b_image : nullptr, dw, accumulate ? 1 : 0, ws, ws_bytes, (hipStream_t)stream);
  IgemmParams p;
  fill_common(p, d);
  p.a = x; p.b = dy;
  p.a_bytes = (uint32_t)((size_t)d->B * d->H * d->W * d->Cin * 4);
  p.b_bytes = (uint32_t)((size_t)d->B * d->Ho * d->Wo * d->Cout * 4);
  p.M = d->KH * d->KW * d->Cin; p.N = d->Cout; p.K = d->B * d->Ho * d->Wo;
  p.div_c.set(d->Cin);
  const bool vec = (d->Cin % 4 == 0) && (d->Cout % 4 == 0) && aligned16(x) && aligned16(dy);
  int var = vec ? 1 : 0;
  if (vec && (32 % d->Wo) == 0) {        // one K-tile = 32 consecutive pixels = a whole number of output rows
    const int rows_adv = 32 / d->Wo;
    p.walk_db = rows_adv / d->Ho;
    p.walk_doh = rows_adv % d->Ho;
    var = 2;
  }
  return run_gemm(MODE_BWD_FILTER, p, (size_t)p.M * p.N, var, dw, nullptr, T2I_ACT_NONE, 0.f, ws, ws_bytes,
                  (hipStream_t)stream, "t2i_conv2d_bwd_filter", accumulate ? 1 : 0);
}

// bf16 tensors (dtype == T2I_DT_BF16) run on the vectorised paths only: C (or n) a multiple of 4, 16-byte aligned pointers; no twin
static int h_contract(int32_t dtype, bool aligned, int64_t quad, const void* twin, const char* what) {
  if (dtype != T2I_DT_F32 && dtype != T2I_DT_BF16) { set_error("%s: dtype must be T2I_DT_F32 or T2I_DT_BF16", what); return T2I_ERR_INVALID; }
  if (dtype == T2I_DT_BF16 && (!aligned || (quad & 3) != 0 || twin)) {
    set_error("%s: bf16 tensors need 16-byte aligned pointers, a channel / element count that is a multiple of 4, and no twin argument", what);
    return T2I_ERR_INVALID;
  }
  return T2I_OK;
}

size_t t2i_col_reduce_workspace_bytes(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return 0;
  return col_reduce_ws(rows, C);
}

int t2i_col_reduce(const void* a, const void* b, const float* center, int64_t rows, int32_t C, float* out0, float* out1,
                   int accumulate, void* ws, size_t ws_bytes, int32_t dtype, t2i_stream_t stream) {
  if (!a || !out0 || rows <= 0 || C <= 0 || (center && !b)) { set_error("t2i_col_reduce: bad argument"); return T2I_ERR_INVALID; }
  if (!ws || ws_bytes < col_reduce_ws(rows, C)) { set_error("t2i_col_reduce: workspace too small"); return T2I_ERR_WORKSPACE; }
  if (int rc = h_contract(dtype, aligned16(a) && aligned16(b) && aligned16(ws) && aligned16(center), C, nullptr, "t2i_col_reduce")) return rc;
  return check(col_reduce_launch(a, b, center, rows, C, out0, out1, accumulate ? 1 : 0, ws, (hipStream_t)stream, dtype == T2I_DT_BF16), "t2i_col_reduce");
}

int t2i_bn_stats(const float* x, int64_t rows, int32_t C, float* sum, float* m2, void* ws, size_t ws_bytes, t2i_stream_t stream) {
  if (!x || !sum || !m2 || rows <= 0 || C <= 0) { set_error("t2i_bn_stats: bad argument"); return T2I_ERR_INVALID; }
  if (!ws || ws_bytes < col_reduce_ws(rows, C)) { set_error("t2i_bn_stats: workspace too small"); return T2I_ERR_WORKSPACE; }
  return check(bn_stats_launch(x, rows, C, sum, m2, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws,
                               (hipStream_t)stream), "t2i_bn_stats");
}

int t2i_bn_train_fwd_stats(const void* x, const float* part_sum, const float* part_m2, int32_t chunks, int32_t tile_rows, int64_t rows,
                           int32_t C, const float* gamma, const float* beta, float eps, float decay, float* mean, float* rstd,
                           float* scale, float* shift, float* moving_mean, float* moving_var, void* ws, size_t ws_bytes,
                           int32_t dtype, t2i_stream_t stream) {
  if (!gamma || !beta || !mean || !rstd || !scale || !shift || rows <= 0 || C <= 0 || ((moving_mean == nullptr) != (moving_var == nullptr)) ||
      ((x == nullptr) == (part_sum == nullptr)) || (part_sum && (!part_m2 || chunks <= 0 || tile_rows <= 0))) {
    set_error("t2i_bn_train_fwd_stats: bad argument (exactly one of x / tile partials)");
    return T2I_ERR_INVALID;
  }
  if (x) {
    if (!ws || ws_bytes < col_reduce_ws(rows, C)) { set_error("t2i_bn_train_fwd_stats: workspace too small"); return T2I_ERR_WORKSPACE; }
    if (int rc = h_contract(dtype, aligned16(x) && aligned16(ws), C, nullptr, "t2i_bn_train_fwd_stats")) return rc;
    return check(bn_stats_launch(x, rows, C, nullptr, nullptr, gamma, beta, eps, decay, mean, rstd, scale, shift, moving_mean, moving_var, ws,
                                 (hipStream_t)stream, dtype == T2I_DT_BF16), "t2i_bn_train_fwd_stats");
  }
  return check(bn_stats_tiles_launch(part_sum, part_m2, chunks, tile_rows, rows, C, nullptr, nullptr, gamma, beta, eps, decay, mean, rstd, scale,
                                     shift, moving_mean, moving_var, (hipStream_t)stream), "t2i_bn_train_fwd_stats");
}

size_t t2i_bn_bwd_fused_workspace_bytes(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return 0;
  return col_reduce_ws(rows, C) + (size_t)3 * C * sizeof(float);
}

int t2i_bn_bwd_fused(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma, int64_t rows,
                     int32_t C, int act, float alpha, void* gmask, void* dx, void* dx_h, float* dgamma, float* dbeta, int accumulate, void* ws,
                     size_t ws_bytes, int32_t dtype, t2i_stream_t stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || rows <= 0 || C <= 0 || (C & 3) || (y && !gmask)) {
    set_error("t2i_bn_bwd_fused: bad argument (C % 4 == 0 required; gmask needed with an activation)");
    return T2I_ERR_INVALID;
  }
  if (!(aligned16(dy) && aligned16(x) && aligned16(dx) && aligned16(mean) && (!y || (aligned16(y) && aligned16(gmask))))) {
    set_error("t2i_bn_bwd_fused: tensors must be 16-byte aligned");
    return T2I_ERR_INVALID;
  }
  if (!ws || ws_bytes < t2i_bn_bwd_fused_workspace_bytes(rows, C) || !aligned16(ws)) { set_error("t2i_bn_bwd_fused: workspace too small"); return T2I_ERR_WORKSPACE; }
  if (!aligned16(dx_h)) { set_error("t2i_bn_bwd_fused: dx_h must be 16-byte aligned"); return T2I_ERR_INVALID; }
  if (int rc = h_contract(dtype, true, C, dx_h, "t2i_bn_bwd_fused")) return rc;
  const bool h = dtype == T2I_DT_BF16;       // bf16 storage: dy, y, x, gmask and dx are bf16 tensors
  return check(bn_bwd_fused_launch(dy, y, x, mean, rstd, gamma, rows, C, act, alpha, gmask, h ? nullptr : reinterpret_cast<float*>(dx), dgamma, dbeta,
                                   accumulate ? 1 : 0, ws, (hipStream_t)stream, h ? dx : dx_h, h), "t2i_bn_bwd_fused");
}

int t2i_bn_stats_tiles(const float* part_sum, const float* part_m2, int32_t chunks, int32_t tile_rows, int64_t rows, int32_t C,
                       float* sum, float* m2, t2i_stream_t stream) {
  if (!part_sum || !part_m2 || !sum || !m2 || chunks <= 0 || tile_rows <= 0 || rows <= 0 || C <= 0 ||
      (int64_t)chunks * tile_rows < rows || (int64_t)(chunks - 1) * tile_rows >= rows) {
    set_error("t2i_bn_stats_tiles: bad argument");
    return T2I_ERR_INVALID;
  }
  return check(bn_stats_tiles_launch(part_sum, part_m2, chunks, tile_rows, rows, C, sum, m2, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr,
                                     nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream), "t2i_bn_stats_tiles");
}

int t2i_col_reduce_partials(const float* part0, const float* part1, int32_t chunks, int32_t C, float* out0, float* out1,
                            int accumulate, t2i_stream_t stream) {
  if (!part0 || !out0 || chunks <= 0 || C <= 0 || ((part1 == nullptr) != (out1 == nullptr))) { set_error("t2i_col_reduce_partials: bad argument"); return T2I_ERR_INVALID; }
  return check(col_reduce_partials_launch(part0, part1, chunks, C, out0, out1, accumulate ? 1 : 0, (hipStream_t)stream),
               "t2i_col_reduce_partials");
}

int t2i_bn_finalize(const float* sum, const float* sumsq, int64_t n, int32_t C, const float* gamma, const float* beta,
                    float eps, float decay, float* mean, float* rstd, float* scale, float* shift, float* moving_mean,
                    float* moving_var, t2i_stream_t stream) {
  if (!sum || !sumsq || !gamma || !beta || !mean || !rstd || !scale || !shift || n <= 0 || C <= 0 ||
      ((moving_mean == nullptr) != (moving_var == nullptr))) {
    set_error("t2i_bn_finalize: bad argument");
    return T2I_ERR_INVALID;
  }
  return check(bn_finalize_launch(sum, sumsq, n, C, gamma, beta, eps, decay, mean, rstd, scale, shift, moving_mean,
                                  moving_var, (hipStream_t)stream), "t2i_bn_finalize");
}

int t2i_bn_apply(const void* x, const float* scale, const float* shift, int64_t rows, int32_t C, int act, float alpha,
                 void* y, void* y_h, int32_t dtype, t2i_stream_t stream) {
  if (!x || !scale || !shift || !y || rows <= 0 || C <= 0) { set_error("t2i_bn_apply: bad argument"); return T2I_ERR_INVALID; }
  const bool al = aligned16(x) && aligned16(y) && aligned16(scale) && aligned16(shift);
  if (y_h && !(al && (C & 3) == 0 && aligned16(y_h))) { set_error("t2i_bn_apply: y_h needs 16-byte aligned tensors and C %% 4 == 0"); return T2I_ERR_INVALID; }
  if (int rc = h_contract(dtype, al, C, y_h, "t2i_bn_apply")) return rc;
  const bool h = dtype == T2I_DT_BF16;
  return check(bn_apply_launch(x, scale, shift, rows, al ? C : -C, act, alpha, h ? nullptr : reinterpret_cast<float*>(y), (hipStream_t)stream,
                               h ? y : y_h, h), "t2i_bn_apply");
}

size_t t2i_bn_grouped_workspace_bytes(int64_t rows_per_group, int32_t C, int32_t groups) {
  if (rows_per_group <= 0 || C <= 0 || groups <= 0) return 0;
  return bn_grouped_ws(rows_per_group, C, groups);
}

int t2i_bn_train_fwd_grouped(const void* x, int64_t rows_per_group, int32_t C, int32_t groups, const float* gamma, const float* beta, float eps,
                             float decay, float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var, int act,
                             float alpha, void* y, void* y_h, const float* tile_sum, const float* tile_m2, int32_t tile_chunks, int32_t tile_rows,
                             int32_t moving_updates, int32_t moving_groups, void* ws, size_t ws_bytes, int32_t dtype, t2i_stream_t stream) {
  // the partials of a group are tile_chunks tiles of tile_rows rows, the last one possibly short: every tile must START inside the group
  // (an over-long tile_chunks would give the merge a tile of <= 0 rows: 1/n = inf, NaN statistics written into the moving averages)
  if (tile_sum && (!tile_m2 || tile_chunks <= 0 || tile_rows <= 0 || (int64_t)tile_chunks * tile_rows < rows_per_group ||
                   (int64_t)(tile_chunks - 1) * tile_rows >= rows_per_group ||
                   (groups > 1 && rows_per_group % tile_rows != 0) || !aligned16(tile_sum) || !aligned16(tile_m2))) {
    set_error("t2i_bn_train_fwd_grouped: bad tile partials (exactly ceil(rows_per_group / tile_rows) tiles of tile_rows rows per group; with groups > 1 a tile must not straddle groups)");
    return T2I_ERR_INVALID;
  }
  if (moving_updates < 1 || moving_updates > 8) {       // a device-side loop count: bounded (the trainers use 1 or 2)
    set_error("t2i_bn_train_fwd_grouped: moving_updates must be in [1, 8]");
    return T2I_ERR_INVALID;
  }
  if (moving_groups < 0 || moving_groups > groups) { set_error("t2i_bn_train_fwd_grouped: moving_groups must be in [0, groups]"); return T2I_ERR_INVALID; }
  if (rows_per_group > 0 && C > 0 && groups > 0) {       // element count against the 2^30 limit the conv descriptors enforce (32-bit offsets in the kernels)
    const int64_t lim = ((int64_t)1 << 30) - 16;
    if (rows_per_group > lim / C || rows_per_group * C > lim / groups) {
      set_error("t2i_bn_train_fwd_grouped: rows_per_group * groups * C exceeds 2^30 - 16 elements");
      return T2I_ERR_INVALID;
    }
  }
  if (!x || !gamma || !beta || !mean || !rstd || !scale || !shift || !y || rows_per_group <= 0 || C <= 0 || (C & 3) || groups <= 0 ||
      ((moving_mean == nullptr) != (moving_var == nullptr))) {
    set_error("t2i_bn_train_fwd_grouped: bad argument (C %% 4 == 0 required)");
    return T2I_ERR_INVALID;
  }
  if (!(aligned16(x) && aligned16(y) && aligned16(scale) && aligned16(shift) && aligned16(y_h))) {
    set_error("t2i_bn_train_fwd_grouped: tensors must be 16-byte aligned");
    return T2I_ERR_INVALID;
  }
  if (!ws || ws_bytes < bn_grouped_ws(rows_per_group, C, groups) || !aligned16(ws)) { set_error("t2i_bn_train_fwd_grouped: workspace too small"); return T2I_ERR_WORKSPACE; }
  if (int rc = h_contract(dtype, true, C, y_h, "t2i_bn_train_fwd_grouped")) return rc;
  const bool h = dtype == T2I_DT_BF16;
  return check(bn_fwd_grouped_launch(x, rows_per_group, C, groups, gamma, beta, eps, decay, mean, rstd, scale, shift, moving_mean, moving_var, act, alpha,
                                     h ? nullptr : reinterpret_cast<float*>(y), h ? y : y_h, ws, (hipStream_t)stream, h, tile_sum, tile_m2, tile_chunks, tile_rows, moving_updates, moving_groups),
               "t2i_bn_train_fwd_grouped");
}

int t2i_bn_bwd_grouped(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma, int64_t rows_per_group,
                       int32_t C, int32_t groups, int act, float alpha, void* gmask, void* dx, void* dx_h, float* dgamma, float* dbeta, int accumulate,
                       void* ws, size_t ws_bytes, int32_t dtype, t2i_stream_t stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || rows_per_group <= 0 || C <= 0 || (C & 3) || groups <= 0 || (y && !gmask)) {
    set_error("t2i_bn_bwd_grouped: bad argument (C %% 4 == 0 required; gmask needed with an activation)");
    return T2I_ERR_INVALID;
  }
  {
    const int64_t lim = ((int64_t)1 << 30) - 16;
    if (rows_per_group > lim / C || rows_per_group * C > lim / groups) {
      set_error("t2i_bn_bwd_grouped: rows_per_group * groups * C exceeds 2^30 - 16 elements");
      return T2I_ERR_INVALID;
    }
  }
  if (!(aligned16(dy) && aligned16(x) && aligned16(dx) && aligned16(mean) && aligned16(dx_h) && (!y || (aligned16(y) && aligned16(gmask))))) {
    set_error("t2i_bn_bwd_grouped: tensors must be 16-byte aligned");
    return T2I_ERR_INVALID;
  }
  if (!ws || ws_bytes < bn_grouped_ws(rows_per_group, C, groups) || !aligned16(ws)) { set_error("t2i_bn_bwd_grouped: workspace too small"); return T2I_ERR_WORKSPACE; }
  if (int rc = h_contract(dtype, true, C, dx_h, "t2i_bn_bwd_grouped")) return rc;
  const bool h = dtype == T2I_DT_BF16;
  return check(bn_bwd_grouped_launch(dy, y, x, mean, rstd, gamma, rows_per_group, C, groups, act, alpha, gmask, h ? nullptr : reinterpret_cast<float*>(dx),
                                     dgamma, dbeta, accumulate ? 1 : 0, ws, (hipStream_t)stream, h ? dx : dx_h, h), "t2i_bn_bwd_grouped");
}

int t2i_bn_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
               const float* sum_dy, const float* sum_dy_x, int64_t rows, int32_t C, float* dx, float* dgamma, float* dbeta,
               int accumulate, void* ws, size_t ws_bytes, t2i_stream_t stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !sum_dy || !sum_dy_x || !dx || !dgamma || !dbeta || rows <= 0 || C <= 0) {
    set_error("t2i_bn_bwd: bad argument");
    return T2I_ERR_INVALID;
  }
  if (!ws || ws_bytes < (size_t)3 * C * sizeof(float) || !aligned16(ws)) { set_error("t2i_bn_bwd: workspace too small"); return T2I_ERR_WORKSPACE; }
  const bool al = aligned16(dy) && aligned16(x) && aligned16(dx);
  return check(bn_bwd_launch(dy, x, mean, rstd, gamma, sum_dy, sum_dy_x, rows, al ? C : -C, dx, dgamma, dbeta,
                             reinterpret_cast<float*>(ws), accumulate ? 1 : 0, (hipStream_t)stream), "t2i_bn_bwd");
}

static int ew_call(int op, const void* a, const void* b, int64_t n, int act, float alpha, float beta, void* y, void* y_h, int32_t dtype,
                   t2i_stream_t stream, const char* what, bool need_b) {
  if (!a || !y || n <= 0 || (need_b && !b)) { set_error("%s: bad argument", what); return T2I_ERR_INVALID; }
  const bool al = aligned16(a) && aligned16(y) && (!b || aligned16(b));
  if (y_h && !(al && (n & 3) == 0 && aligned16(y_h))) { set_error("%s: the bf16 image needs 16-byte aligned tensors and n %% 4 == 0", what); return T2I_ERR_INVALID; }
  if (int rc = h_contract(dtype, al, n, y_h, what)) return rc;
  const bool h = dtype == T2I_DT_BF16;
  // unaligned views take the scalar tail path: tell the kernel there is no float4 body
  return check(ew_launch(op, a, b, al ? (size_t)n : ((size_t)n | (1ull << 63)), act, alpha, beta, h ? nullptr : reinterpret_cast<float*>(y),
                         (hipStream_t)stream, h ? y : y_h, h), what);
}

int t2i_act_fwd(const void* x, int64_t n, int act, float alpha, void* y, void* y_h, int32_t dtype, t2i_stream_t stream) {
  return ew_call(0, x, nullptr, n, act, alpha, 0.f, y, y_h, dtype, stream, "t2i_act_fwd", false);
}
int t2i_act_bwd(const void* dy, const void* y, int64_t n, int act, float alpha, void* dx, void* dx_h, int32_t dtype, t2i_stream_t stream) {
  return ew_call(1, dy, y, n, act, alpha, 0.f, dx, dx_h, dtype, stream, "t2i_act_bwd", true);
}
int t2i_act_bwd_colsum(const void* dy, const void* y, const void* x2, const float* center, int64_t rows, int32_t C, int act,
                       float alpha, void* dx, void* dx_h, float* colsum, float* colsum_x2, int accumulate, void* ws, size_t ws_bytes,
                       int32_t dtype, t2i_stream_t stream) {
  if (center && (!x2 || !aligned16(center))) { set_error("t2i_act_bwd_colsum: center needs x2 and 16-byte alignment"); return T2I_ERR_INVALID; }
  if ((x2 == nullptr) != (colsum_x2 == nullptr) || (x2 && !aligned16(x2))) { set_error("t2i_act_bwd_colsum: x2 / colsum_x2 must come together, 16-byte aligned"); return T2I_ERR_INVALID; }
  if (!dy || !y || !dx || !colsum || rows <= 0 || C <= 0 || (C & 3)) { set_error("t2i_act_bwd_colsum: bad argument (C % 4 == 0 required)"); return T2I_ERR_INVALID; }
  if (!(aligned16(dy) && aligned16(y) && aligned16(dx))) { set_error("t2i_act_bwd_colsum: tensors must be 16-byte aligned"); return T2I_ERR_INVALID; }
  if (!ws || ws_bytes < col_reduce_ws(rows, C) || !aligned16(ws)) { set_error("t2i_act_bwd_colsum: workspace too small"); return T2I_ERR_WORKSPACE; }
  if (!aligned16(dx_h)) { set_error("t2i_act_bwd_colsum: dx_h must be 16-byte aligned"); return T2I_ERR_INVALID; }
  if (int rc = h_contract(dtype, true, C, dx_h, "t2i_act_bwd_colsum")) return rc;
  const bool h = dtype == T2I_DT_BF16;
  return check(act_bwd_colsum_launch(dy, y, x2, center, rows, C, act, alpha, h ? nullptr : reinterpret_cast<float*>(dx), colsum, colsum_x2,
                                     accumulate ? 1 : 0, ws, (hipStream_t)stream, h ? dx : dx_h, h), "t2i_act_bwd_colsum");
}
int t2i_add_act(const void* a, const void* b, int64_t n, int act, float alpha, void* y, void* y_h, int32_t dtype, t2i_stream_t stream) {
  return ew_call(2, a, b, n, act, alpha, 0.f, y, y_h, dtype, stream, "t2i_add_act", true);
}
int t2i_axpby(const void* a, float alpha, const void* b, float beta, int64_t n, void* y, int32_t dtype, t2i_stream_t stream) {
  return ew_call(3, a, b, n, T2I_ACT_NONE, alpha, beta, y, nullptr, dtype, stream, "t2i_axpby", false);
}

int t2i_interp(const float* eps, const float* g, const float* x, int32_t B, int64_t per_sample, float* xhat,
               t2i_stream_t stream) {
  if (!eps || !g || !x || !xhat || B <= 0 || per_sample <= 0) { set_error("t2i_interp: bad argument"); return T2I_ERR_INVALID; }
  return check(interp_launch(eps, g, x, B, per_sample, xhat, (hipStream_t)stream), "t2i_interp");
}

static bool dt_ok(int32_t dtype, const char* what) {
  if (dtype == T2I_DT_F32 || dtype == T2I_DT_BF16) return true;
  set_error("%s: dtype must be T2I_DT_F32 or T2I_DT_BF16", what);
  return false;
}

int t2i_concat_tile_fwd(const void* feat, const void* emb, int32_t B, int32_t P, int32_t Cf, int32_t Ce, void* out,
                        int32_t dtype, t2i_stream_t stream) {
  if (!feat || !emb || !out || B <= 0 || P <= 0 || Cf <= 0 || Ce <= 0 || !dt_ok(dtype, "t2i_concat_tile_fwd")) { set_error("t2i_concat_tile_fwd: bad argument"); return T2I_ERR_INVALID; }
  return check(concat_tile_fwd_launch(feat, emb, B, P, Cf, Ce, out, (hipStream_t)stream, dtype == T2I_DT_BF16), "t2i_concat_tile_fwd");
}

int t2i_concat_tile_bwd(const void* dout, int32_t B, int32_t P, int32_t Cf, int32_t Ce, void* dfeat, void* demb,
                        int32_t dtype, t2i_stream_t stream) {
  if (!dout || !dfeat || !demb || B <= 0 || P <= 0 || Cf <= 0 || Ce <= 0 || !dt_ok(dtype, "t2i_concat_tile_bwd")) { set_error("t2i_concat_tile_bwd: bad argument"); return T2I_ERR_INVALID; }
  return check(concat_tile_bwd_launch(dout, B, P, Cf, Ce, dfeat, demb, (hipStream_t)stream, dtype == T2I_DT_BF16), "t2i_concat_tile_bwd");
}

int t2i_nchw_to_nhwc(const void* x, int32_t B, int32_t C, int32_t HW, void* y, int32_t dtype, t2i_stream_t stream) {
  if (!x || !y || B <= 0 || C <= 0 || HW <= 0 || !dt_ok(dtype, "t2i_nchw_to_nhwc")) { set_error("t2i_nchw_to_nhwc: bad argument"); return T2I_ERR_INVALID; }
  return check(transpose_launch(x, B, C, HW, y, (hipStream_t)stream, dtype == T2I_DT_BF16), "t2i_nchw_to_nhwc");   // [C,HW] -> [HW,C]
}

int t2i_nhwc_to_nchw(const void* x, int32_t B, int32_t C, int32_t HW, void* y, int32_t dtype, t2i_stream_t stream) {
  if (!x || !y || B <= 0 || C <= 0 || HW <= 0 || !dt_ok(dtype, "t2i_nhwc_to_nchw")) { set_error("t2i_nhwc_to_nchw: bad argument"); return T2I_ERR_INVALID; }
  return check(transpose_launch(x, B, HW, C, y, (hipStream_t)stream, dtype == T2I_DT_BF16), "t2i_nhwc_to_nchw");   // [HW,C] -> [C,HW]
}

int t2i_gp_slopes(const void* g, int32_t B, int64_t per_sample, float* slopes, int32_t dtype, t2i_stream_t stream) {
  if (!g || !slopes || B <= 0 || per_sample <= 0 || !dt_ok(dtype, "t2i_gp_slopes")) { set_error("t2i_gp_slopes: bad argument"); return T2I_ERR_INVALID; }
  if (dtype == T2I_DT_BF16 && (per_sample & 3) != 0) { set_error("t2i_gp_slopes: bf16 rows need a multiple of 4 elements"); return T2I_ERR_INVALID; }
  return check(gp_slopes_launch(g, B, per_sample, slopes, (hipStream_t)stream, dtype == T2I_DT_BF16), "t2i_gp_slopes");
}

int t2i_row_scale(const void* g, const float* coef, int32_t B, int64_t per_sample, void* out, int32_t dtype, t2i_stream_t stream) {
  if (!g || !coef || !out || B <= 0 || per_sample <= 0 || !dt_ok(dtype, "t2i_row_scale")) { set_error("t2i_row_scale: bad argument"); return T2I_ERR_INVALID; }
  return check(row_scale_launch(g, coef, B, per_sample, out, (hipStream_t)stream, dtype == T2I_DT_BF16), "t2i_row_scale");
}

int t2i_row_scale_div(const void* g, const float* num, const float* den, int32_t B, int64_t per_sample, void* out, int32_t dtype, t2i_stream_t stream) {
  if (!g || !num || !den || !out || B <= 0 || per_sample <= 0 || !dt_ok(dtype, "t2i_row_scale_div")) { set_error("t2i_row_scale_div: bad argument"); return T2I_ERR_INVALID; }
  return check(row_scale_launch(g, num, B, per_sample, out, (hipStream_t)stream, dtype == T2I_DT_BF16, den), "t2i_row_scale_div");
}

int t2i_crop_flip_normalize(const uint8_t* src, int64_t N, int32_t S, const int32_t* ids, const int32_t* row0,
                            const int32_t* col0, const int32_t* flip, int32_t B, int32_t out_size, float* out,
                            t2i_stream_t stream) {
  if (!src || !ids || !row0 || !col0 || !flip || !out || N <= 0 || S <= 0 || B <= 0 || out_size <= 0 || out_size > S) {
    set_error("t2i_crop_flip_normalize: bad argument");
    return T2I_ERR_INVALID;
  }
  return check(crop_flip_normalize_launch(src, S, ids, row0, col0, flip, B, out_size, out, (hipStream_t)stream),
               "t2i_crop_flip_normalize");
}

int t2i_gather_mean(const float* emb, int64_t N, int32_t En, int32_t D, const int32_t* ids, const int32_t* choice, int32_t B,
                    int32_t k, float* out, t2i_stream_t stream) {
  if (!emb || !ids || !choice || !out || N <= 0 || En <= 0 || D <= 0 || B <= 0 || k <= 0 || k > En) {
    set_error("t2i_gather_mean: bad argument");
    return T2I_ERR_INVALID;
  }
  return check(gather_mean_launch(emb, En, D, ids, choice, B, k, out, (hipStream_t)stream), "t2i_gather_mean");
}

int t2i_pool2_sum(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float scale, float* y, t2i_stream_t stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1)) { set_error("t2i_pool2_sum: bad argument (H, W must be even)"); return T2I_ERR_INVALID; }
  return check(resample2_launch(true, x, B, H / 2, W / 2, C, scale, y, (hipStream_t)stream), "t2i_pool2_sum");
}

int t2i_upscale2(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float scale, float* y, t2i_stream_t stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) { set_error("t2i_upscale2: bad argument"); return T2I_ERR_INVALID; }
  return check(resample2_launch(false, x, B, 2 * H, 2 * W, C, scale, y, (hipStream_t)stream), "t2i_upscale2");
}

size_t t2i_row_moments_workspace_bytes(int32_t B) { return B > 0 ? row_moments_ws(B) : 0; }

int t2i_row_moments(const float* a, const float* b, int32_t B, int64_t per_sample, float* s1, float* s2, void* ws,
                    size_t ws_bytes, t2i_stream_t stream) {
  if (!a || !s1 || !s2 || B <= 0 || per_sample <= 0) { set_error("t2i_row_moments: bad argument"); return T2I_ERR_INVALID; }
  if (!ws || ws_bytes < row_moments_ws(B)) { set_error("t2i_row_moments: workspace too small"); return T2I_ERR_WORKSPACE; }
  return check(row_moments_launch(a, b, B, per_sample, s1, s2, ws, (hipStream_t)stream), "t2i_row_moments");
}

int t2i_row_fma2(const float* a, const float* b, const float* alpha, const float* gamma, const float* delta, int32_t B,
                 int64_t per_sample, float* out, t2i_stream_t stream) {
  if (!a || !alpha || !out || B <= 0 || per_sample <= 0 || ((b == nullptr) != (gamma == nullptr))) { set_error("t2i_row_fma2: bad argument"); return T2I_ERR_INVALID; }
  return check(row_fma2_launch(a, b, alpha, gamma, delta, B, per_sample, out, (hipStream_t)stream), "t2i_row_fma2");
}

int t2i_wgan_d_head(const float* logits, const float* slopes1, const float* slopes2, const float* kt_dev, int32_t B,
                    float gp_coeff, float* seed_logits, float* seed_slopes1, float* seed_slopes2, float* scalars,
                    t2i_stream_t stream) {
  if (!logits || !slopes1 || !slopes2 || !seed_logits || !seed_slopes1 || !seed_slopes2 || !scalars || B <= 0) {
    set_error("t2i_wgan_d_head: bad argument");
    return T2I_ERR_INVALID;
  }
  return check(wgan_d_head_launch(logits, slopes1, slopes2, kt_dev, B, gp_coeff, seed_logits, seed_slopes1, seed_slopes2, scalars,
                                  (hipStream_t)stream), "t2i_wgan_d_head");
}

int t2i_sigmoid_ce_head(const float* l0, const float* l1, const float* l2, float y0, float y1, float y2, float w0, float w1, float w2, int32_t B,
                        float* seed0, float* seed1, float* seed2, float* prob0, float* prob1, float* prob2, float* losses, t2i_stream_t stream) {
  if (!l0 || !losses || B <= 0 || (!l1 && (seed1 || prob1)) || (!l2 && (seed2 || prob2))) {
    set_error("t2i_sigmoid_ce_head: bad argument (l0 and losses are required; outputs only for heads that have logits)");
    return T2I_ERR_INVALID;
  }
  const float* l[3] = {l0, l1, l2};
  const float y[3] = {y0, y1, y2}, w[3] = {w0, w1, w2};
  float* seed[3] = {seed0, seed1, seed2};
  float* prob[3] = {prob0, prob1, prob2};
  return check(sigmoid_ce_head_launch(l, y, w, seed, prob, B, losses, (hipStream_t)stream), "t2i_sigmoid_ce_head");
}

int t2i_ca_kl_fwd(const float* mean, const float* log_sigma, const float* eps, int64_t n, float* code, float* kl, t2i_stream_t stream) {
  if (!mean || !log_sigma || !eps || !code || !kl || n <= 0 || n > (1 << 24)) { set_error("t2i_ca_kl_fwd: bad argument"); return T2I_ERR_INVALID; }
  return check(ca_kl_fwd_launch(mean, log_sigma, eps, (int)n, code, kl, (hipStream_t)stream), "t2i_ca_kl_fwd");
}

int t2i_ca_kl_bwd(const float* mean, const float* log_sigma, const float* eps, const float* dcode, const float* dkl, int64_t n,
                  float* dmean, float* dlog_sigma, t2i_stream_t stream) {
  if (!mean || !log_sigma || !eps || !dmean || !dlog_sigma || n <= 0 || n > (1 << 24)) { set_error("t2i_ca_kl_bwd: bad argument"); return T2I_ERR_INVALID; }
  return check(ca_kl_bwd_launch(mean, log_sigma, eps, dcode, dkl, (int)n, dmean, dlog_sigma, (hipStream_t)stream), "t2i_ca_kl_bwd");
}

int t2i_lerp_dev(const float* a, const float* b, const float* t_dev, int32_t mode, int64_t n, float* out, t2i_stream_t stream) {
  if (!a || !t_dev || !out || n <= 0 || mode < 0 || mode > 2 || (mode == 0 && !b)) { set_error("t2i_lerp_dev: bad argument"); return T2I_ERR_INVALID; }
  return check(lerp_dev_launch(a, b, t_dev, mode, (size_t)n, out, (hipStream_t)stream), "t2i_lerp_dev");
}

int t2i_adam_tf(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, const float* lr_t_dev, float beta1,
                float beta2, float eps,
                float grad_scale, t2i_stream_t stream) {
  if (!w || !g || !v || n <= 0) { set_error("t2i_adam_tf: bad argument"); return T2I_ERR_INVALID; }
  if (!m && beta1 != 0.f) { set_error("t2i_adam_tf: m may be NULL only with beta1 == 0 (the first moment is then grad * grad_scale)"); return T2I_ERR_INVALID; }
  if (!(aligned16(w) && aligned16(g) && (!m || aligned16(m)) && aligned16(v))) { set_error("t2i_adam_tf: arena must be 16-byte aligned"); return T2I_ERR_INVALID; }
  filter_cache_invalidate(w, (size_t)n * 4);          // transformed filters of this arena are stale from here on
  const int rc = check(adam_tf_launch(w, g, m, v, n, lr_t, lr_t_dev, beta1, beta2, eps, grad_scale, (hipStream_t)stream), "t2i_adam_tf");
  if (rc != T2I_OK || !tuning().cache_refresh) return rc;
  return filter_cache_refresh(w, (size_t)n * 4, (hipStream_t)stream);   // ... and regenerated behind the update, all in one launch
}

long long t2i_stat(const char* key) {
  if (key && !strcmp(key, "pair_fused")) return g_stat_pair_fused.load();
  return -1;
}

int t2i_tuning_set(const char* key, double value) {
  if (!key) { set_error("t2i_tuning_set: null key"); return T2I_ERR_INVALID; }
  Tuning& t = tuning_mut();
  struct { const char* name; int* field; } ints[] = {
      {"force_tile", &t.force_tile}, {"force_splitk", &t.force_splitk}, {"debug_plan", &t.debug_plan}, {"group_n", &t.group_n},
      {"no_ut", &t.no_ut}, {"no_thin", &t.no_thin}, {"winograd", &t.winograd}, {"winograd_minc", &t.winograd_minc},
      {"winograd_maxhw", &t.winograd_maxhw}, {"winograd_k4s2", &t.winograd_k4s2}, {"winograd_k4s2_minc", &t.winograd_k4s2_minc},
      {"winograd_k4s2_bwd_minc", &t.winograd_k4s2_bwd_minc}, {"winograd_k4s2_bwdf", &t.winograd_k4s2_bwdf}, {"winograd_k4s2_minwork", &t.winograd_k4s2_minwork}, {"winograd_k4s2_minitems", &t.winograd_k4s2_minitems},
      {"adam_blocks", &t.adam_blocks}, {"max_chain", &t.max_chain}, {"bf16_operands", &t.bf16_operands},
      {"cache_refresh", &t.cache_refresh}, {"thin_parts", &t.thin_parts}, {"batch_lin", &t.batch_lin}, {"bgemm", &t.bgemm}, {"winograd_minwork", &t.winograd_minwork}, {"bf16_dma", &t.bf16_dma}, {"hft_boost", &t.hft_boost}, {"hft_ovh", &t.hft_ovh},
      {"bgemm_tile", &t.bgemm_tile}, {"bgemm_big_items", &t.bgemm_big_items}, {"vec_epi", &t.vec_epi}, {"pair", &t.pair}, {"pair_cus", &t.pair_cus}, {"h_stats", &t.h_stats}, {"pair_reduce", &t.pair_reduce}, {"bf16_waves", &t.bf16_waves}, {"bf16_pair_tiles", &t.bf16_pair_tiles}, {"colred_wgs", &t.colred_wgs}, {"colred_cap", &t.colred_cap}, {"bn_fuse", &t.bn_fuse}, {"wino_fuse", &t.wino_fuse}, {"wino_fuse_items", &t.wino_fuse_items}, {"wino_fuse_xf", &t.wino_fuse_xf}, {"tile8_eff", &t.tile8_eff}, {"dma_ovh", &t.dma_ovh}, {"dma_split_us", &t.dma_split_us}, {"pair_max_px", &t.pair_max_px}};
  for (auto& e : ints)
    if (!strcmp(key, e.name)) { *e.field = (int)value; return T2I_OK; }
  if (!strcmp(key, "split_cost")) { t.split_cost = value; return T2I_OK; }
  set_error("t2i_tuning_set: unknown key '%s'", key);
  return T2I_ERR_INVALID;
}

int t2i_kt_sgd(float* kt, const float* wdist_sums, float scale, float lr, t2i_stream_t stream) {
  if (!kt || !wdist_sums) { set_error("t2i_kt_sgd: bad argument"); return T2I_ERR_INVALID; }
  return check(kt_sgd_launch(kt, wdist_sums, scale, lr, (hipStream_t)stream), "t2i_kt_sgd");
}

int t2i_zero_ranges(float* base, const int64_t* ranges, int32_t n, t2i_stream_t stream) {
  if (!base || !ranges || n <= 0) { set_error("t2i_zero_ranges: bad argument"); return T2I_ERR_INVALID; }
  return check(zero_ranges_launch(base, reinterpret_cast<const long long*>(ranges), n, (hipStream_t)stream), "t2i_zero_ranges");
}

int t2i_trunc_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, float mean, float std, float lo, float hi, t2i_stream_t stream) {
  if (!out || n <= 0 || !(std > 0.f) || !(lo < hi)) { set_error("t2i_trunc_normal: bad argument (n > 0, std > 0, lo < hi)"); return T2I_ERR_INVALID; }
  return check(trunc_normal_launch(out, (size_t)n, seed, offset, mean, std, lo, hi, (hipStream_t)stream), "t2i_trunc_normal");
}

int t2i_conv2d_algo(const t2i_conv_desc* d, int32_t which) {
  if (validate_desc(d) || which < 0 || which > 2) return -1;
  const bool thin = !tuning().no_thin;
  if (which == 0) {
    if (thin && (head_conv_eligible(*d) || tiny_conv_eligible(*d, false) || stem_fwd_eligible(*d))) return T2I_ALGO_DIRECT_SMALL;
    if (h_eligible(*d, false)) return T2I_ALGO_IMPLICIT_GEMM_BF16_OPERANDS;
    if (winograd_eligible(*d, false)) return T2I_ALGO_WINOGRAD_F2X2_3X3;
    if (winograd_k4s2_eligible(*d, false)) return T2I_ALGO_WINOGRAD_F2X2_2X2;
  } else if (which == 1) {
    if (thin && (head_conv_eligible(*d) || tiny_conv_eligible(*d, true) || thin_deconv_eligible(*d))) return T2I_ALGO_DIRECT_SMALL;
    if (h_eligible(*d, true)) return T2I_ALGO_IMPLICIT_GEMM_BF16_OPERANDS;
    if (winograd_eligible(*d, true)) return T2I_ALGO_WINOGRAD_F2X2_3X3;
    if (winograd_k4s2_eligible(*d, true)) return T2I_ALGO_WINOGRAD_F2X2_2X2;
  } else {
    if (thin && (head_conv_eligible(*d) || tiny_bwdw_eligible(*d) || stem_bwdf_eligible(*d))) return T2I_ALGO_DIRECT_SMALL;
    if (winograd_filter_eligible(*d)) return T2I_ALGO_WINOGRAD_F2X2_3X3;
    if (winograd_k4s2_eligible(*d, false) && tuning().winograd_k4s2_bwdf) return T2I_ALGO_WINOGRAD_F2X2_2X2;
    if (h_filter_eligible(*d)) return T2I_ALGO_IMPLICIT_GEMM_BF16_OPERANDS;
  }
  return T2I_ALGO_IMPLICIT_GEMM;
}

int t2i_filter_cache_attach(void* buf, size_t bytes) {
  if ((buf == nullptr) != (bytes == 0) || !aligned16(buf)) { set_error("t2i_filter_cache_attach: need a 16-byte aligned buffer and its size, or (NULL, 0)"); return T2I_ERR_INVALID; }
  return filter_cache_attach(buf, bytes);
}

int t2i_filter_cache_enable(int on) { return filter_cache_enable(on); }

void t2i_filter_cache_invalidate(const void* p, size_t bytes) { filter_cache_invalidate(p, bytes); }

size_t t2i_filter_cache_bytes(void) { return filter_cache_bytes(); }

int t2i_filter_cache_refresh(const void* p, size_t bytes, t2i_stream_t stream) { return filter_cache_refresh(p, bytes, (hipStream_t)stream); }
int t2i_filter_cache_assume(const void* p, size_t bytes, t2i_stream_t stream) { return filter_cache_assume(p, bytes, (hipStream_t)stream); }

int t2i_cast_bf16(const float* x, int64_t n, void* out, t2i_stream_t stream) {
  if (!x || !out || n <= 0 || (n % 8) != 0 || !aligned16(x) || !aligned16(out)) {
    set_error("t2i_cast_bf16: need n %% 8 == 0 and 16-byte aligned buffers");
    return T2I_ERR_INVALID;
  }
  return check(cast_bf16_launch(x, (size_t)n, out, (hipStream_t)stream), "t2i_cast_bf16");
}

int t2i_cast_f32(const void* x_bf16, int64_t n, float* out, t2i_stream_t stream) {
  if (!x_bf16 || !out || n <= 0) { set_error("t2i_cast_f32: bad argument"); return T2I_ERR_INVALID; }
  return check(cast_f32_launch(x_bf16, (size_t)n, out, (hipStream_t)stream), "t2i_cast_f32");
}

uint64_t t2i_capture_id(t2i_stream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo((hipStream_t)stream, &st, &id) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return st == hipStreamCaptureStatusActive ? (uint64_t)id + 1 : 0;
}

size_t t2i_conv2d_input_transform_bytes(const t2i_conv_desc* d) {
  if (!d || validate_desc(d)) return 0;
  if (!((d->Cin % 4) == 0 && (d->Cout % 4) == 0)) return 0;
  return xform_bytes(*d);
}

}  // extern "C"
