/*
 * t2i_hip.h — C ABI of libt2i_hip.so: the MI355X (gfx950) kernels behind the reference's operator surface.
 *
 * The reference (crisbodnar/text-to-image) has no FFI/plugin layer: its operator API is the Python wrappers in
 * utils/ops.py, which hand the arithmetic to TensorFlow-1.4 kernels (Eigen / cuDNN).  This library is what sits
 * under those wrappers instead.  Each entry point cites the reference interface it serves.
 *
 * Conventions (all entry points):
 *   - extern "C"; return 0 on success, a negative T2I_ERR_* otherwise; never throw.  t2i_last_error() gives the
 *     thread-local message of the last failure.
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates or frees device memory.
 *     Scratch comes from the caller's workspace (size from the matching *_workspace_bytes query; 256-byte aligned); the
 *     optional transformed-filter cache lives in an arena the caller attaches (t2i_filter_cache_attach).
 *   - tensors are dense NHWC fp32 ("[B,H,W,C]", C fastest); conv filters are TF "HWIO" [KH,KW,Cin,Cout];
 *     dense kernels are [in,out].  No tensor may exceed 2^30-16 elements (4 GiB: buffer addressing).
 *   - every call is asynchronous on the given hipStream_t (void* here so the header needs no HIP include) and is
 *     safe to capture into a hipGraph: no allocation, no synchronisation.  Host-side state is limited to (a) the tuning
 *     switches, read from the T2I_* environment once at first use and changed afterwards only through t2i_tuning_set, and
 *     (b) the bookkeeping (not the storage) of the filter cache, mutex-guarded; the planner never consults the environment
 *     at call time, so equal descriptors take equal paths for the life of the process.  There is NO per-thread hand-over
 *     state: everything a call reads or writes besides its tensors travels in its own arguments (t2i_conv_opts for the
 *     conv family, an explicit image pointer for the elementwise producers) — ABI v5 removed the one-shot "arm the next
 *     call" entry points of v4 (t2i_conv2d_operand_images, t2i_output_image, t2i_conv2d_input_transform, *_written, *_kept).
 */
#ifndef T2I_HIP_H
#define T2I_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2I_OK 0
#define T2I_ERR_INVALID (-1)   /* bad descriptor / null pointer / unsupported geometry */
#define T2I_ERR_WORKSPACE (-2) /* workspace too small or misaligned */
#define T2I_ERR_LAUNCH (-3)    /* HIP launch failure (message holds hipGetErrorString) */

/* activation fused into an epilogue (reference utils/ops.py `act=` argument; model.py:110,131 lrelu 0.2) */
#define T2I_ACT_NONE 0
#define T2I_ACT_LRELU 1 /* max(x, alpha*x) */
#define T2I_ACT_RELU 2
#define T2I_ACT_TANH 3

typedef void* t2i_stream_t; /* a hipStream_t */

/* Element type of the ACTIVATION tensors of a call (v6).  T2I_DT_F32: dense fp32, the reference's dtype and the default everywhere.
 * T2I_DT_BF16: the tensors are bf16 in memory (BASELINE config 3 end to end: activations and activation gradients travel between
 * kernels as bf16; weights, biases, optimizer state, batch-norm statistics, every reduction result and all accumulators stay
 * fp32).  Entry points with a `dtype` argument read AND write all their activation tensors in that type (pointers are void* for
 * that reason); bf16 tensors need 16-byte aligned pointers and a channel / element count that is a multiple of 4, and there is no
 * separate bf16 twin then (the y_h arguments must be NULL).  Arithmetic is fp32 either way: a bf16 tensor is widened exactly on load
 * and rounded to nearest even on store. */
enum { T2I_DT_F32 = 0, T2I_DT_BF16 = 1 };

/* Geometry of one 2-D convolution, already resolved from the TF padding string (SAME/VALID -> pad_t/pad_l, Ho/Wo):
 * y[b,oh,ow,co] = sum_{kh,kw,ci} x[b, oh*SH-pad_t+kh, ow*SW-pad_l+kw, ci] * w[kh,kw,ci,co]. */
typedef struct t2i_conv_desc {
  int32_t B, H, W, Cin;  /* input  [B,H,W,Cin]   */
  int32_t Ho, Wo, Cout;  /* output [B,Ho,Wo,Cout] */
  int32_t KH, KW, SH, SW;
  int32_t pad_t, pad_l;  /* TF SAME puts the odd pixel bottom/right, so only top/left are needed */
  int32_t math;          /* T2I_MATH_F32 (0): exact fp32 matrix pipe.  T2I_MATH_BF16 (1): operands rounded to bf16 (RNE),
                          * v_mfma_*_bf16 with fp32 accumulation; the tensors at this interface stay fp32 (BASELINE
                          * config 3).  Where the gathered tensor has a multiple of 64 channels the forward conv and the
                          * input gradient first stage bf16 copies of their operands (activation: workspace; filter:
                          * workspace or the filter cache) and run a GEMM whose operands are bf16 in memory; elsewhere
                          * the rounding happens on the way into LDS.  Same arithmetic either way.  The direct kernels for
                          * thin layers (Cin or Cout <= 4, the logit head) compute in fp32 in both modes. */
} t2i_conv_desc;
enum { T2I_MATH_F32 = 0, T2I_MATH_BF16 = 1 };

/* ---- library ------------------------------------------------------------------------------------------------ */
int t2i_version(void);            /* ABI version, currently 9 (v9: t2i_conv_opts gained xform_valid_rows / xform_plane_rows, t2i_bn_train_fwd_grouped gained moving_groups, t2i_trunc_normal and t2i_zero_ranges added; v8: t2i_sigmoid_ce_head, t2i_bn_train_fwd_grouped, t2i_bn_bwd_grouped, t2i_bn_grouped_workspace_bytes added — no existing signature changed; v7: v7: t2i_conv2d_bwd_pair, t2i_row_scale_div, t2i_stat, t2i_filter_cache_assume added, t2i_adam_tf takes m == NULL at beta1 == 0 — no existing signature changed; v2: t2i_conv_desc.math; v3: caller-owned filter-cache arena,
                                   * t2i_tuning_set, t2i_kt_sgd; v4: t2i_filter_cache_refresh, bf16 operand images; v5: t2i_conv_opts
                                   * and explicit image arguments instead of thread-local one-shot hand-overs; v6: bf16 STORAGE —
                                   * activation tensors may be bf16 at this interface: t2i_dtype arguments, t2i_conv_opts.in_dtype /
                                   * out_dtype) */
const char* t2i_last_error(void); /* thread-local, never NULL */
long long t2i_stat(const char* key); /* process-wide counters for tests / diagnostics: "pair_fused" = t2i_conv2d_bwd_pair calls issued as one launch; -1: unknown key */
/* CU count, clock (kHz) and gcnArchName of `device` into caller buffers; used by bench.py to re-derive peaks. */
int t2i_device_info(int device, int32_t* cu_count, int32_t* clock_khz, char* arch, size_t arch_len);

/* Tuning / diagnostic switch `key` := value (tests, sweeps: tools/sweep_conv.py).  Keys and their T2I_* environment
 * defaults: force_tile (T2I_FORCE_TILE: 22, 21, 12, 11 = 128x128 ... 64x64; 0 = planner), force_splitk, debug_plan, group_n,
 * no_ut, no_thin, winograd, winograd_minc, winograd_maxhw, winograd_k4s2, winograd_k4s2_minc, winograd_k4s2_bwd_minc,
 * winograd_k4s2_bwdf, adam_blocks, max_chain (longest unsplit fp32 reduction chain, default 8192), split_cost, bf16_operands, cache_refresh, thin_parts.
 * Not a hot-path call; changes apply to launches planned afterwards (workspace queries included). */
int t2i_tuning_set(const char* key, double value);

/* ---- convolution family: reference utils/ops.py:58-63 (conv2d) and :66-71 (conv2d_transpose) ------------------ */
size_t t2i_conv2d_workspace_bytes(const t2i_conv_desc* d); /* upper bound for fwd / bwd_data / bwd_filter */

/* Optional side inputs / outputs of ONE conv call (pass NULL for none).  Plain data, owned by the caller, read at entry and
 * written (the two out-flags) before the call returns; the library keeps no pointer to it.  Results of the convolution are
 * identical with and without any of these.
 *   a_image / b_image   bf16 math: bf16 images (same shape and layout, made by t2i_cast_bf16 or written as a twin by the
 *                       producer) of the call's first / second ACTIVATION operand (fwd: x; bwd_data: dy; bwd_filter: x, dy).
 *                       An activation feeds up to three convs of a training step and a gradient two, so the caller that keeps
 *                       the image saves the repeated casts.  Paths that do not read bf16 operands from memory ignore them.
 *   out_image           bf16 math, fwd / fwd_stats / bwd_data: also write the bf16 image of the output tensor here (16-byte
 *                       aligned), in the epilogue's own pass; out_image_written tells whether the path taken did.
 *   xform, xform_bytes, xform_mode
 *                       fp32 Winograd: the forward conv of a layer and its filter gradient transform the same x (V = B^T x B
 *                       per tile, 4x resp. 2.25x the size of x).  T2I_XFORM_KEEP on t2i_conv2d_fwd / _fwd_stats leaves V in
 *                       `xform` (>= t2i_conv2d_input_transform_bytes(d) bytes; xform_kept tells whether the call took a path
 *                       that has one); T2I_XFORM_HAVE on t2i_conv2d_bwd_filter (same d, same unchanged x) reads V from there
 *                       instead of transforming x again; xform_valid_rows (below) restricts what is trusted to the leading images of the
 *                       batch — the stacked critic step (text-to-image_amd/stacked.py) replaces the x_hat rows of a layer's input by
 *                       the gradient penalty's tangent before its ONE filter-gradient launch over all 4B rows.
 *   in_dtype, out_dtype bf16 STORAGE: the activation tensors of the call are bf16 in memory.  The bf16-operand GEMMs and the
 *                       3 -> 128 stem read and write them directly (no cast, no fp32 copy anywhere); the remaining paths (thin /
 *                       head kernels, the generic kernel for channel counts that are not multiples of 64) run on fp32 staging copies
 *                       carved from the workspace — t2i_conv2d_workspace_bytes includes the room for them. */
enum { T2I_XFORM_NONE = 0, T2I_XFORM_KEEP = 1, T2I_XFORM_HAVE = 2 };
typedef struct t2i_conv_opts {
  const void* a_image;
  const void* b_image;
  void* out_image;
  void* xform;
  size_t xform_bytes;
  int32_t xform_mode;
  int32_t out_image_written; /* out */
  int32_t xform_kept;        /* out */
  int32_t in_dtype;          /* bf16 storage (needs math = T2I_MATH_BF16): bit 0 / bit 1 set = the call's first / second ACTIVATION operand
                              * (the pointer argument itself) is a bf16 tensor; a_image / b_image are then not consulted for it */
  int32_t out_dtype;         /* T2I_DT_BF16: the output pointer (y / dx) is a bf16 tensor and is the only thing written */
  int32_t xform_valid_rows;  /* v9, T2I_XFORM_HAVE: `xform` is current for the first xform_valid_rows images of the batch only (the caller
                              * changed the images behind them since the forward conv); t2i_conv2d_bwd_filter regenerates the transform of
                              * the remaining images from x, in place in `xform`.  0 (or >= B): all of it is current.  (v8: `reserved`, 0) */
  int32_t xform_plane_rows;  /* v9, T2I_XFORM_HAVE: `xform` was kept by the forward conv of a LARGER batch of that many images whose leading B
                              * images are this call's x (a stacked pass of which only the leading part is differentiated).  0: the batch is d->B.
                              * Honoured by the 3x3 Winograd form; other forms transform x anew. */
  int32_t reserved;
} t2i_conv_opts;
size_t t2i_conv2d_input_transform_bytes(const t2i_conv_desc* d);   /* > 0: fwd and bwd_filter of `d` both take a Winograd path */

/* y = act(conv(x, w) + bias).  bias may be NULL.  Serves ops.conv2d (utils/ops.py:58-63), ops.fc as a 1x1 conv on
 * [B,1,1,in] (utils/ops.py:84-87), and the double-backward term adj_gy = conv(ggx, w) of the gradient penalty. */
int t2i_conv2d_fwd(const t2i_conv_desc* d, const void* x, const float* w, const float* bias, void* y, int act,
                   float alpha, t2i_conv_opts* opts, void* ws, size_t ws_bytes, t2i_stream_t stream);

/* t2i_conv2d_fwd that also hands the batch norm behind it its statistics: if the launch takes the unsplit path,
 * *chunks = number of M-tiles, *tile_rows = their height and stats = [2][chunks][Cout]: per tile the column sums of y and
 * the second moment of y ABOUT THE TILE'S OWN MEAN (finish with t2i_bn_stats_tiles(stats, stats + chunks*Cout, chunks,
 * tile_rows, B*Ho*Wo, Cout, sum, m2, ...)); otherwise *chunks = 0 and the caller computes the statistics itself
 * (t2i_bn_stats).  stats must hold t2i_conv2d_stats_bytes(d). */
size_t t2i_conv2d_stats_bytes(const t2i_conv_desc* d);
int t2i_conv2d_fwd_stats(const t2i_conv_desc* d, const void* x, const float* w, const float* bias, void* y, int act,
                         float alpha, float* stats, size_t stats_bytes, int32_t* chunks, int32_t* tile_rows, t2i_conv_opts* opts,
                         void* ws, size_t ws_bytes, t2i_stream_t stream);

/* dx = conv^T(dy, w) (+ bias over Cin if non-NULL, then act).  This IS ops.conv2d_transpose (utils/ops.py:66-71):
 * TF stores the deconv filter as [KH,KW,Cout_deconv,Cin_deconv], i.e. the HWIO filter of the adjoint conv, so the
 * descriptor is that adjoint conv's (d->Cin = deconv output channels) and no re-layout is needed. */
int t2i_conv2d_bwd_data(const t2i_conv_desc* d, const void* dy, const float* w, const float* bias, void* dx,
                        int act, float alpha, t2i_conv_opts* opts, void* ws, size_t ws_bytes, t2i_stream_t stream);

/* dw = x (*) dy over all B*Ho*Wo positions (tf.gradients wrt `weights`, reference models/wgancls/model.py:94-106);
 * accumulate != 0: dw += x (*) dy in the epilogue, i.e. the gradient is summed straight into the optimizer's arena. */
int t2i_conv2d_bwd_filter(const t2i_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate,
                          t2i_conv_opts* opts, void* ws, size_t ws_bytes, t2i_stream_t stream);

/* One layer's backward on bf16 tensors in (possibly) ONE launch (v7).  In tf.gradients' walk of the reference graph
 * (models/wgancls/model.py:94-106) every conv2d (utils/ops.py:58-63) contributes an input gradient conv^T(g, w) AND a filter
 * gradient x (*) g, every conv2d_transpose (utils/ops.py:66-71) contributes conv(g, w) AND a filter gradient g (*) saved dy: two
 * independent GEMMs on the same incoming gradient.  first = T2I_PAIR_BWD_DATA: out1 = conv^T(g, w);  T2I_PAIR_FWD: out1 = conv(g, w)
 * (no bias, no activation);  then dw = fx (*) fdy (accumulate != 0: dw += ...).  The results are the SAME SUMS as those of
 * t2i_conv2d_bwd_data / t2i_conv2d_fwd followed by t2i_conv2d_bwd_filter with the same opts (same operand rounding, same fp32
 * accumulation inside a K range) but not necessarily the same bits: a shared launch plans each GEMM for its share of the chip
 * (tuning pair_cus, default half), which may choose another split-K grouping, i.e. another order of the fp32 partial sums
 * (tests/test_storage_gpu.py bounds the difference at the fp32 GEMM tolerance).  Hence bf16-storage results depend on whether the
 * pair path is taken (kernels.pair_calls, and autograd takes it only on one stream): runs are bit-reproducible for a FIXED
 * setting, the side-stream / pair toggles are not bit-neutral in bf16.  Where both GEMMs run on the
 * bf16-operand kernels they share one launch (a B = 64 layer leaves each of them one workgroup per CU; two streams of a captured
 * graph do not overlap on this stack).  opts1 (in_dtype bit 0: g, out_dtype: out1) and opts2 (bit 0: fx, bit 1: fdy) are
 * required; ws1 / ws2 (each t2i_conv2d_workspace_bytes(d)) are in use at the same time and must not overlap. */
enum { T2I_PAIR_FWD = 0, T2I_PAIR_BWD_DATA = 1 };
int t2i_conv2d_bwd_pair(const t2i_conv_desc* d, int first, const void* g, const float* w, void* out1, t2i_conv_opts* opts1,
                        const void* fx, const void* fdy, float* dw, int accumulate, t2i_conv_opts* opts2,
                        void* ws1, size_t ws1_bytes, void* ws2, size_t ws2_bytes, t2i_stream_t stream);

/* ---- column reductions over a [rows, C] view ------------------------------------------------------------------ */
size_t t2i_col_reduce_workspace_bytes(int64_t rows, int32_t C);
/* out0[c] = sum_r a[r,c];  out1[c] = sum_r a[r,c]*(b[r,c] - center[c])  (b == NULL -> a*a; center == NULL -> 0;
 * out1 == NULL -> skipped).  Bias gradients (db = colsum(dy)) and the batch-norm backward sums (sum dy, sum dy*(x - mean):
 * centred inside the reduction — sum(dy*x) - mean*sum(dy) would cancel). */
int t2i_col_reduce(const void* a, const void* b, const float* center, int64_t rows, int32_t C, float* out0, float* out1,
                   int accumulate, void* ws, size_t ws_bytes, int32_t dtype /* of a and b */,
                   t2i_stream_t stream); /* accumulate != 0: out += (gradient arena) */

/* ---- batch norm, training mode: reference utils/ops.py:7-29 (tf.contrib.layers.batch_norm fused, scale=True) --- */
/* Second stage alone: out0[c] = sum_k part0[k*C + c] (k < chunks, fixed order), same for part1/out1 when given. */
int t2i_col_reduce_partials(const float* part0, const float* part1, int32_t chunks, int32_t C, float* out0, float* out1,
                            int accumulate, t2i_stream_t stream);
/* Batch statistics of x [rows, C], numerically stable: sum[c] = sum_r x[r,c] and m2[c] = sum_r (x[r,c] - mean[c])^2, from
 * per-chunk moments about a sample of the chunk merged with Chan's update — NOT from sum(x^2) - sum(x)^2/n, which loses the
 * variance to cancellation when |mean| >> std (a rank-2 batch norm over a batch of 2; deep batch-normed stacks at batch 2).
 * Workspace as t2i_col_reduce.  t2i_bn_stats_tiles: the same from the per-tile partials of t2i_conv2d_fwd_stats. */
int t2i_bn_stats(const float* x, int64_t rows, int32_t C, float* sum, float* m2, void* ws, size_t ws_bytes, t2i_stream_t stream);
int t2i_bn_stats_tiles(const float* part_sum, const float* part_m2, int32_t chunks, int32_t tile_rows, int64_t rows, int32_t C,
                       float* sum, float* m2, t2i_stream_t stream);
/* Statistics AND the finalize step in one chain (stage 1 + one stage-2 launch): exactly one of x (the tensor, [rows, C];
 * workspace as t2i_col_reduce) or the tile partials of t2i_conv2d_fwd_stats (part_sum / part_m2 / chunks / tile_rows).
 * Outputs as t2i_bn_finalize.  What the training-mode batch norm of utils/ops.py:7-29 launches in front of t2i_bn_apply. */
int t2i_bn_train_fwd_stats(const void* x, const float* part_sum, const float* part_m2, int32_t chunks, int32_t tile_rows, int64_t rows,
                           int32_t C, const float* gamma, const float* beta, float eps, float decay, float* mean, float* rstd,
                           float* scale, float* shift, float* moving_mean, float* moving_var, void* ws, size_t ws_bytes,
                           int32_t dtype /* of x */, t2i_stream_t stream);
/* From sum and the centred second moment m2 over n rows: mean, rstd = 1/sqrt(m2/n + eps) (biased variance);
 * scale = gamma*rstd, shift = beta-mean*scale; and, if moving_mean != NULL,
 * moving = decay*moving + (1-decay)*{mean, var_biased*n/(n-1)} in place. */
int t2i_bn_finalize(const float* sum, const float* m2, int64_t n, int32_t C, const float* gamma,
                    const float* beta, float eps, float decay, float* mean, float* rstd, float* scale, float* shift,
                    float* moving_mean, float* moving_var, t2i_stream_t stream);
/* y = act(x*scale[c] + shift[c])  (normalise + affine + activation in one pass; also eval-mode BN).
 * y_h (here and in t2i_bn_bwd_fused, t2i_act_fwd, t2i_act_bwd, t2i_act_bwd_colsum, t2i_add_act): NULL, or a buffer that receives
 * the bf16 image (round to nearest even) of the output tensor in the same pass — the "twin" a bf16-math conv reading that
 * tensor next takes as its operand image (t2i_conv_opts.a_image), so that no cast launch is needed.  Requires the vectorised
 * path: every tensor and y_h 16-byte aligned and C % 4 == 0 (n % 4 == 0); otherwise the call fails with T2I_ERR_INVALID
 * instead of silently not writing it. */
int t2i_bn_apply(const void* x, const float* scale, const float* shift, int64_t rows, int32_t C, int act,
                 float alpha, void* y, void* y_h, int32_t dtype, t2i_stream_t stream);
/* dx = gamma*rstd*(dy - sum_dy/n - xhat*sum_dy_xhat/n), xhat = (x-mean)*rstd, sum_dy_xhat = rstd * sum_dy_x with
 * sum_dy_x = sum dy*(x - mean) (t2i_col_reduce / t2i_act_bwd_colsum with center = mean).  Also emits dgamma, dbeta. */
int t2i_bn_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
               const float* sum_dy, const float* sum_dy_x, int64_t rows, int32_t C, float* dx, float* dgamma,
               float* dbeta, int accumulate /* dgamma/dbeta += */, void* ws /* >= 3*C floats */, size_t ws_bytes,
               t2i_stream_t stream);

/* The whole training-mode batch-norm backward in three launches: [g = dy*act'(y) and the reductions sum g, sum g*(x - mean),
 * stage 1] -> [stage 2 + dgamma/dbeta + the coefficients of dx] -> [dx = k_dy*g + k_x*x + k_0].  y == NULL: no activation
 * behind the batch norm (g = dy, gmask unused).  gmask: caller's [rows, C] buffer for g.  C % 4 == 0, 16-byte alignment. */
size_t t2i_bn_bwd_fused_workspace_bytes(int64_t rows, int32_t C);
int t2i_bn_bwd_fused(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma, int64_t rows,
                     int32_t C, int act, float alpha, void* gmask, void* dx, void* dx_h, float* dgamma, float* dbeta, int accumulate, void* ws,
                     size_t ws_bytes, int32_t dtype /* of dy, y, x, gmask, dx */, t2i_stream_t stream);

/* Batch norm of a STACKED batch (v8).  The reference evaluates its batch-normalised critic three times per sess.run with shared variables
 * (models/gancls/model.py:48-51: fake / match / mismatch images; models/stackgan/stageI/model.py:44-48): convolutions, activations and the text
 * projection are per-sample, so the three passes can run as ONE batch of groups * B samples — except batch norm, which must keep each pass's own
 * statistics.  x is [groups * rows_per_group, C], group g = rows [g * rows_per_group, (g + 1) * rows_per_group) (contiguous: the batch axis is the
 * slowest).  Forward: statistics per group (mean / rstd / scale / shift are [groups][C]), moving averages updated once per group in group order
 * (what `groups` sequential passes do), y = act(x * scale_g + shift_g) — three launches for all groups.  Backward: dx with the group's own
 * statistics, dgamma / dbeta summed over the groups (accumulate != 0: += into the arena slots) — three launches.  groups = 1 is the ordinary
 * training-mode batch norm.  C % 4 == 0, 16-byte aligned tensors; ws >= t2i_bn_grouped_workspace_bytes(...) for either call.
 * tile_sum / tile_m2 (optional): the per-tile partials the producing conv's epilogue left (t2i_conv2d_fwd_stats: tile_chunks tiles of tile_rows rows
 * per group) — the statistics' first pass over x is then skipped.  With at most 64 partial rows per column and group the second stage runs in the
 * prologue of the normalisation / dx kernel (two launches, one with tile partials; T2I_BN_FUSE=0: always three).
 * moving_groups (v9; 0 = all): only the first moving_groups groups move the moving averages — a stacked pass whose later groups are evaluations
 * outside UPDATE_OPS (the generator's critic-step evaluation stacked behind its generator-step evaluation).
 * moving_updates (>= 1): how many sequential exponential-average steps the moving statistics take with this batch's statistics per group — 2 when
 * ONE evaluation of a network stands for two identical evaluations of the reference graph (gancls: the generator in the D run and in the G run of
 * one iteration, models/gancls/trainer.py:115-134: same feed, no noise, weights unchanged in between; both runs sit under UPDATE_OPS). */
size_t t2i_bn_grouped_workspace_bytes(int64_t rows_per_group, int32_t C, int32_t groups);
int t2i_bn_train_fwd_grouped(const void* x, int64_t rows_per_group, int32_t C, int32_t groups, const float* gamma, const float* beta, float eps,
                             float decay, float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var, int act,
                             float alpha, void* y, void* y_h, const float* tile_sum, const float* tile_m2, int32_t tile_chunks, int32_t tile_rows,
                             int32_t moving_updates, int32_t moving_groups, void* ws, size_t ws_bytes, int32_t dtype, t2i_stream_t stream);
int t2i_bn_bwd_grouped(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma, int64_t rows_per_group,
                       int32_t C, int32_t groups, int act, float alpha, void* gmask, void* dx, void* dx_h, float* dgamma, float* dbeta, int accumulate,
                       void* ws, size_t ws_bytes, int32_t dtype, t2i_stream_t stream);

/* ---- elementwise ----------------------------------------------------------------------------------------------- */
/* y = act(x) */
int t2i_act_fwd(const void* x, int64_t n, int act, float alpha, void* y, void* y_h, int32_t dtype, t2i_stream_t stream);
/* dx = dy * act'(.) with the derivative taken from the OUTPUT y (lrelu/relu are sign preserving, tanh' = 1-y^2). */
int t2i_act_bwd(const void* dy, const void* y, int64_t n, int act, float alpha, void* dx, void* dx_h, int32_t dtype, t2i_stream_t stream);
/* Fused activation backward + column sums: dx = dy * act'(y), colsum[c] = sum_r dx[r,c] and, if x2 != NULL,
 * colsum_x2[c] = sum_r dx[r,c]*(x2[r,c] - center[c]) (center NULL = 0), in ONE pass over a [rows, C] view (C % 4 == 0,
 * 16-byte aligned).  Serves the bias gradient of a conv layer and the two reductions of the batch-norm backward (x2 = the
 * layer input, center = its batch mean).  accumulate: sums are added to the outputs.  Workspace as t2i_col_reduce. */
int t2i_act_bwd_colsum(const void* dy, const void* y, const void* x2, const float* center, int64_t rows, int32_t C, int act,
                       float alpha, void* dx, void* dx_h, float* colsum, float* colsum_x2, int accumulate, void* ws, size_t ws_bytes,
                       int32_t dtype, t2i_stream_t stream);
/* y = act(a + b): residual joins (reference models/wgancls/model.py:145-146, 190-191, 206-207). */
int t2i_add_act(const void* a, const void* b, int64_t n, int act, float alpha, void* y, void* y_h, int32_t dtype, t2i_stream_t stream);
/* y = alpha*a + beta*b (b may be NULL). */
int t2i_axpby(const void* a, float alpha, const void* b, float beta, int64_t n, void* y, int32_t dtype, t2i_stream_t stream);
/* x_hat[b,:] = eps[b]*g[b,:] + (1-eps[b])*x[b,:]   (reference models/wgancls/model.py:53). */
int t2i_interp(const float* eps, const float* g, const float* x, int32_t B, int64_t per_sample, float* xhat,
               t2i_stream_t stream);
/* out[b,p,:] = concat(feat[b,p,:Cf], emb[b,:Ce]) for p < P: tile the compressed text embedding over the 4x4 map
 * and concatenate on channels (reference models/wgancls/model.py:153-155).  bwd splits the gradient back. */
int t2i_concat_tile_fwd(const void* feat, const void* emb, int32_t B, int32_t P, int32_t Cf, int32_t Ce,
                        void* out, int32_t dtype, t2i_stream_t stream);
int t2i_concat_tile_bwd(const void* dout, int32_t B, int32_t P, int32_t Cf, int32_t Ce, void* dfeat, void* demb,
                        int32_t dtype, t2i_stream_t stream);
/* NCHW <-> NHWC physical transposes: reference utils/ops.py:129-134 (to_nchw / to_nhwc) and the dense_2 reshape. */
int t2i_nchw_to_nhwc(const void* x, int32_t B, int32_t C, int32_t HW, void* y, int32_t dtype, t2i_stream_t stream);
int t2i_nhwc_to_nchw(const void* x, int32_t B, int32_t C, int32_t HW, void* y, int32_t dtype, t2i_stream_t stream);

/* ---- gradient penalty: reference models/wgancls/model.py:62-70 -------------------------------------------------- */
/* slopes[b] = sqrt(sum_j g[b,j]^2); one wavefront-shuffle reduction per sample. */
int t2i_gp_slopes(const void* g, int32_t B, int64_t per_sample, float* slopes, int32_t dtype, t2i_stream_t stream);
/* out[b,:] = coef[b] * g[b,:]  (per-sample scaling: backward of the slope norm, and its own double backward). */
int t2i_row_scale(const void* g, const float* coef, int32_t B, int64_t per_sample, void* out, int32_t dtype, t2i_stream_t stream);
/* out[b,:] = (den[b] > 0 ? num[b] / max(den[b], 1e-30) : 0) * g[b,:]: the slope norm's backward with its coefficient d / ||g_b|| formed
 * in the kernel (v7; tf.gradients of tf.sqrt(tf.reduce_sum(tf.square(.))), model.py:64,69). */
int t2i_row_scale_div(const void* g, const float* num, const float* den, int32_t B, int64_t per_sample, void* out, int32_t dtype, t2i_stream_t stream);

/* ---- loss heads: reference models/wgancls/model.py:72-92 (critic losses) and :117-127 (conditioning augmentation) --- */
/* From the critic's 3B logits (fake | real | mismatch, each B) and the two per-sample slope vectors: the loss scalars
 *   scalars[0..11] = D_loss, D_loss_real, D_loss_fake, D_loss_mismatch, wdist, wdist2, real_gp, real_gp2, reg_loss,
 *                    balance_loss, d balance_loss / d kt, kt
 * with D_loss = -wdist - kt*wdist2 + gp_coeff*(real_gp + real_gp2), real_gp = mean(max(0, slope-1)^2), and the seeds
 * of the backward pass dD_loss/dlogits [3B], dD_loss/dslopes1 [B], dD_loss/dslopes2 [B].  kt_dev: device scalar, or NULL
 * for kt = 1 (models/pggan/pggan.py:104).  One workgroup; replaces ~40 elementwise launches. */
int t2i_wgan_d_head(const float* logits, const float* slopes1, const float* slopes2, const float* kt_dev, int32_t B,
                    float gp_coeff, float* seed_logits, float* seed_slopes1, float* seed_slopes2, float* scalars,
                    t2i_stream_t stream);
/* Sigmoid cross-entropy heads: reference models/gancls/trainer.py:20-34 (tf.nn.sigmoid_cross_entropy_with_logits on the critic's fake /
 * match / mismatch logits with constant labels 0 / 0.9 / 0, weighted 1-alpha / 1 / alpha; G_loss with label 1) and
 * models/stackgan/stageI/trainer.py:53-77.  For each head k < 3 with logits l_k[B] (l1, l2 may be NULL):
 *   losses[1+k] = mean_i [max(l,0) - l y_k + log1p(exp(-|l|))],  losses[0] = sum_k w_k losses[1+k],
 *   seed_k[i] = w_k (sigmoid(l) - y_k) / B  (= d losses[0] / d l_k[i]),  prob_k[i] = sigmoid(l)   (seed_k / prob_k may be NULL).
 * One workgroup; replaces ~25 elementwise / reduction launches per set of heads. */
int t2i_sigmoid_ce_head(const float* l0, const float* l1, const float* l2, float y0, float y1, float y2, float w0, float w1, float w2, int32_t B,
                        float* seed0, float* seed1, float* seed2, float* prob0, float* prob1, float* prob2, float* losses, t2i_stream_t stream);
/* code = mean + exp(log_sigma)*eps (eps: the truncated-normal draw) and kl[0] = mean(-ls + .5(-1 + exp(2 ls) + mean^2))
 * over the n = B*D elements; bwd: dmean = dcode + dkl/n * mean, dlog_sigma = dcode*eps*exp(ls) + dkl/n * (exp(2 ls) - 1)
 * (dcode and/or dkl may be NULL = zero). */
int t2i_ca_kl_fwd(const float* mean, const float* log_sigma, const float* eps, int64_t n, float* code, float* kl, t2i_stream_t stream);
int t2i_ca_kl_bwd(const float* mean, const float* log_sigma, const float* eps, const float* dcode, const float* dkl, int64_t n,
                  float* dmean, float* dlog_sigma, t2i_stream_t stream);

/* ---- optimizer: tf.train.AdamOptimizer as used at reference models/wgancls/model.py:94-106 ---------------------- */
/* m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g*g; w -= lr_t*m/(sqrt(v)+eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the
 * caller (epsilon outside the bias correction).  One launch over a flat parameter arena. grad_scale multiplies g
 * first (1/world_size for summed data-parallel gradients).  If lr_t_dev != NULL the step size is read from that device
 * scalar instead of lr_t, so a hipGraph that captured the launch can be replayed with the next step's bias correction.
 * m may be NULL when beta1 == 0 (v7): m_t = g_t * grad_scale whatever m_{t-1} was, so it is neither read nor written. */
int t2i_adam_tf(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, const float* lr_t_dev, float beta1,
                float beta2, float eps, float grad_scale, t2i_stream_t stream);

/* kt <- kt - lr * 2 (kt*wd2 - wd) * wd2, the SGD step on balance_loss = (kt*wdist2 - wdist)^2 (reference
 * models/wgancls/model.py:85,100: GradientDescentOptimizer(0.001) minimising balance_loss over kt).  wdist_sums = device
 * [2] holding wdist and wdist2 SUMMED over the data-parallel ranks (each rank's value being its batch mean) and
 * scale = 1/ranks, so that wd, wd2 are the global-batch means (the loss is quadratic in them: averaging per-rank
 * gradients would not be its gradient).  Single rank: the two means and scale = 1. */
int t2i_kt_sgd(float* kt, const float* wdist_sums, float scale, float lr, t2i_stream_t stream);
/* v9.  base[start_r + i] = 0 for every range r < n of the DEVICE table ranges[n][2] = (start, length) in elements: the small slots of a gradient arena
 * whose large filter slots take their first contribution of a step as a plain store (accumulate = 0) and therefore need no zero-fill. */
int t2i_zero_ranges(float* base, const int64_t* ranges, int32_t n, t2i_stream_t stream);
/* v9.  out[i] = mean + std * t_i, t_i ~ N(0, 1) truncated to [lo, hi] (tf.truncated_normal: reference models/wgancls/model.py:119, the
 * conditioning-augmentation noise redrawn on every run), by CDF inversion of Philox4x32-10 uniforms keyed by (seed, offset + i / 4): the
 * draw is a pure function of (seed, offset, i).  The caller advances offset by (n + 3) / 4 per call.  One launch (the tensor library's
 * trunc_normal_ is eight). */
int t2i_trunc_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, float mean, float std, float lo, float hi, t2i_stream_t stream);

/* ---- PGGAN operators: reference utils/ops.py:74-81 (layer_norm), :100-101 (pool), :109-111 (upscale) ------------ */
/* y[b,h,w,:] = scale * sum of the 2x2 window of x [B,H,W,C] (H, W even) -> [B,H/2,W/2,C].  scale = 1/4 is
 * ops.pool(x, 2) (AVG, SAME, even extents); scale = 1 is the backward of t2i_upscale2. */
int t2i_pool2_sum(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float scale, float* y, t2i_stream_t stream);
/* y[b,2h+i,2w+j,:] = scale * x[b,h,w,:]: ops.upscale(x, 2) (nearest neighbour) for scale = 1; scale = 1/4 is the
 * backward of the average pool. */
int t2i_upscale2(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float scale, float* y, t2i_stream_t stream);
/* s1[b] = sum_i a[b,i]; s2[b] = sum_i a[b,i]*b[b,i] (b = a when NULL): the per-sample moments of layer norm. */
size_t t2i_row_moments_workspace_bytes(int32_t B);
int t2i_row_moments(const float* a, const float* b, int32_t B, int64_t per_sample, float* s1, float* s2, void* ws,
                    size_t ws_bytes, t2i_stream_t stream);
/* out[b,i] = a[b,i]*alpha[b] + b[b,i]*gamma[b] + delta[b]  (b/gamma and delta optional): layer-norm normalise and its
 * backward. */
int t2i_row_fma2(const float* a, const float* b, const float* alpha, const float* gamma, const float* delta, int32_t B,
                 int64_t per_sample, float* out, t2i_stream_t stream);

/* Fade-in mix with the weight t in device memory (reference models/pggan/pggan.py:267,314 reads the `alpha_tra` variable):
 * mode 0: out = (1-t)*a + t*b;  mode 1: out = t*a;  mode 2: out = (1-t)*a  (modes 1, 2 = the backward of mode 0). */
int t2i_lerp_dev(const float* a, const float* b, const float* t_dev, int32_t mode, int64_t n, float* out, t2i_stream_t stream);

/* Which algorithm the three conv entry points pick for this descriptor (which: 0 = t2i_conv2d_fwd, 1 = t2i_conv2d_bwd_data,
 * 2 = t2i_conv2d_bwd_filter), assuming 16-byte aligned tensors.  For tests, benchmarks and roofline accounting: the
 * Winograd paths execute 1/2.25 (F(2x2,3x3)) resp. 9/16 (F(2x2,2x2), 4x4 stride 2) of the direct convolution's
 * multiply-adds.  Returns -1 for an invalid descriptor. */
enum {
  T2I_ALGO_IMPLICIT_GEMM = 0,        /* igemm_kernel on the direct convolution */
  T2I_ALGO_WINOGRAD_F2X2_3X3 = 1,    /* 3x3 stride 1: transforms + 16 batched GEMMs */
  T2I_ALGO_WINOGRAD_F2X2_2X2 = 2,    /* 4x4 stride 2: space-to-depth / per-phase F(2x2,2x2), 9 or 36 batched GEMMs */
  T2I_ALGO_DIRECT_SMALL = 3,         /* thin / tiny / head kernels of the 3-channel and 1-output layers */
  T2I_ALGO_IMPLICIT_GEMM_BF16_OPERANDS = 4 /* math = BF16, gathered channels % 64 == 0 (fwd, bwd_data): the gathered tensor and
                                      * the filter are first copied to bf16 (workspace / filter cache), then igemm_h_kernel */
};
int t2i_conv2d_algo(const t2i_conv_desc* d, int32_t which);

/* ---- transformed-filter cache (optional) -------------------------------------------------------------------------
 * The Winograd paths of the three conv entry points transform the filter (U = G g G^T) on every call.  A training step
 * uses each critic filter in up to six convs between two optimizer updates; with the cache on, the transform is kept in
 * a slot of a CALLER-OWNED arena, keyed by (filter pointer, transform kind, Cin, Cout), and reused until the filter changes.
 * t2i_filter_cache_attach(buf, bytes) hands the library that arena (device memory, 16-byte aligned, on the device the
 * convs run on); slots are carved from it in order and never moved; when it is full, further filters are simply
 * transformed per call.  attach(NULL, 0) detaches (all entries dropped); the caller frees the arena only after that and
 * after every graph captured with the cache on has been destroyed.  The wgancls step at the reference's widths needs 0.65 GB.
 * CONTRACT when enabled: filter memory may be modified only by t2i_adam_tf (which drops the entries of the arena it
 * updates) or must be followed by t2i_filter_cache_invalidate(ptr, bytes) (ptr NULL = everything) — this includes
 * initialisation, checkpoint loads, broadcasts, and replaying a captured graph that contains t2i_adam_tf from a process
 * that also issues eager convs.  Launches captured into a hipGraph reuse only transforms filled in the same capture, so a
 * graph always contains every transform it depends on; an image filled lazily by one captured stream is not handed to another
 * stream of the same capture (no dependency between them) — only images regenerated by t2i_filter_cache_refresh, which must be
 * issued before the capture's streams fork, are shared across its streams.  Results are bit-identical with and without the cache.  Off by
 * default; t2i_filter_cache_enable returns the previous state. */
int t2i_filter_cache_attach(void* buf, size_t bytes);
int t2i_filter_cache_enable(int on);
void t2i_filter_cache_invalidate(const void* ptr, size_t bytes);
size_t t2i_filter_cache_bytes(void);            /* bytes of the attached arena handed out so far */
/* Regenerates, in ONE launch per 80 entries, every cached image whose filter lies inside [ptr, ptr + bytes) (ptr NULL: all —
 * only if every filter the cache has ever seen is still allocated: entries outlive their filters) and is
 * stale in the launch context of `stream` (eager, or the capture active on it), and marks it filled for that context.
 * Call it behind t2i_adam_tf for the arena it updated (or set the tuning key cache_refresh = 1 and t2i_adam_tf does), and at
 * the head of a capture with (NULL, 0): an iteration then holds one batched regeneration per optimizer step instead of one
 * small fill per (filter, kind) at its first use — ~60 launches.  The bytes moved are the same as with lazy fills as long as
 * an image is not regenerated twice between two updates of its filter. */
int t2i_filter_cache_refresh(const void* ptr, size_t bytes, t2i_stream_t stream);
/* v7: marks every cached image whose filter lies inside [ptr, ptr + bytes) as filled for the launch context of `stream` WITHOUT
 * regenerating it — the caller's promise that, whenever the work issued (or captured) after this call runs, the images in the
 * cache arena are those of the current filters.  For a captured iteration that regenerates an arena's images behind its own
 * t2i_adam_tf (mid-graph t2i_filter_cache_refresh): every replay then finds them as the previous replay left them, and the
 * regeneration at the head of the graph (one read of every filter per iteration) is redundant.  The caller must regenerate
 * them eagerly (t2i_filter_cache_refresh outside the capture, same stream as the replay) before the first replay and after any
 * other writer touched the filters (the writers that must call t2i_filter_cache_invalidate).  Launches nothing. */
int t2i_filter_cache_assume(const void* ptr, size_t bytes, t2i_stream_t stream);

/* ---- bf16 operand images (T2I_MATH_BF16) --------------------------------------------------------------------------
 * out[i] = bf16(x[i]), round to nearest even; n % 8 == 0, both buffers 16-byte aligned.  The image of an activation tensor
 * is handed to the convs that read it through t2i_conv_opts.a_image / b_image. */
int t2i_cast_bf16(const float* x, int64_t n, void* out, t2i_stream_t stream);
/* out[i] = float(x[i]) for a bf16 tensor x (exact widening; any n and alignment): a bf16 activation handed to fp32 code. */
int t2i_cast_f32(const void* x_bf16, int64_t n, float* out, t2i_stream_t stream);
/* 0 for an eager stream, else a number unique to the capture active on `stream`: a caller that keeps bf16 images (or any
 * derived buffer) across calls must not let a capture reuse one made outside it — the graph would not contain its producer. */
uint64_t t2i_capture_id(t2i_stream_t stream);

/* ---- data pipeline: reference preprocess/dataset.py ------------------------------ */
/* out[b] = crop/flip/normalise of stored image ids[b] (reference Dataset.next_batch + transform, dataset.py:83-96,150):
 * src [N,S,S,3] uint8 resident on the device; out [B,out_size,out_size,3] float32 with
 *   out[b,r,c,:] = u8 * (2/255) - 1  at  src[ids[b], row0[b]+r, col0[b] + (flip[b] ? out_size-1-c : c), :]
 * in float32 without fused multiply-add, i.e. bit-identical to NumPy's `images.astype(float32) * (2./255) - 1.`.
 * ids/row0/col0/flip are device int32[B]; the caller guarantees ids < N and row0/col0 + out_size <= S (the reference
 * draws them as floor((S - out_size) * U[0,1))). */
int t2i_crop_flip_normalize(const uint8_t* src, int64_t N, int32_t S, const int32_t* ids, const int32_t* row0,
                            const int32_t* col0, const int32_t* flip, int32_t B, int32_t out_size, float* out,
                            t2i_stream_t stream);
/* out[b,:] = mean_j emb[ids[b], choice[b,j], :]  (reference Dataset.sample_embeddings, dataset.py:98-120: mean of `k`
 * of the image's caption embeddings).  emb [N,En,D] float32, choice device int32[B,k]; sequential fp32 sum in choice
 * order then division by k — bit-identical to np.mean(e[choice], axis=0). */
int t2i_gather_mean(const float* emb, int64_t N, int32_t En, int32_t D, const int32_t* ids, const int32_t* choice, int32_t B,
                    int32_t k, float* out, t2i_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* T2I_HIP_H */
