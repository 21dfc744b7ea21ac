"""torch.autograd.Functions whose forward AND backward are libt2i_hip.so kernels.

The critic's gradient penalty (reference models/wgancls/model.py:62-70) differentiates through a gradient, so every
Function on the critic path has a backward that is itself composed of Functions (conv <-> conv^T <-> filter-gradient
form a closed family: the double backward needs no new kernel type).  Generator-only ops (batch norm, tanh, slope
norm) are first-order (``once_differentiable``), exactly what the reference's two optimizers need.

``input_grads_only()`` marks the first-order pass of the gradient penalty (tf.gradients(y, [x]) at model.py:63,68):
there only d/d(input) is wanted, so filter / bias gradients are not launched.
"""
import contextlib
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import kernels as K

_INPUTS_ONLY = [False]

# Gradient sinks: weight storage address -> view of the optimizer's gradient arena.  In the final (first-order) backward
# a filter gradient whose weight has a sink is ACCUMULATED by the GEMM epilogue straight into the arena and autograd gets
# `None` for it: no temporary dw tensor, no separate `grad += dw` pass (0.55 ms/iteration of elementwise adds before).
# Registered by optim.Arena.enable_sinks().  NOTIFY[0], if set (dp.DataParallel), is called with the parameter's address
# after each contribution has been issued: sunk gradients never reach AccumulateGrad, so the data-parallel bucket
# overlap counts these notifications instead of post-accumulate hooks.
SINKS = {}          # address -> (weakref to the parameter leaf, its slot of the gradient arena as a flat view)
NOTIFY = [None]


def register_sink(param, slot, first_touch=None):
    """first_touch: callable() -> bool (optim.Arena): True exactly once per step for a slot whose first contribution may be a plain store."""
    SINKS[param.data_ptr()] = (weakref.ref(param), slot, first_touch)


def sink_accumulate(ptr):
    """Must the contribution about to be written into the sink of the parameter at `ptr` be ADDED to what the slot holds?  False exactly for
    the first contribution of a step into a slot the arena does not zero (optim.Arena.zero_grad): that one is written as a plain store."""
    e = SINKS.get(ptr)
    if e is None or e[2] is None:
        return True
    return not e[2]()


def sink_at(ptr):
    """The arena slot registered for the parameter stored at `ptr`, or None.  An entry whose parameter has been freed is
    dropped: the allocator may hand its address to an unrelated tensor (a stale entry would swallow that tensor's
    gradient into a dead arena)."""
    e = SINKS.get(ptr)
    if e is None:
        return None
    if e[0]() is None:
        del SINKS[ptr]
        return None
    return e[1]


def _notify(t):
    cb = NOTIFY[0]
    if cb is not None and t is not None:
        cb(t.data_ptr())


def _sink_of(t):
    """The gradient-arena slot of parameter `t` if sinks are on and this is the final (first-order) backward."""
    if t is None or torch.is_grad_enabled():
        return None
    return sink_at(t.data_ptr())


class _Side:
    """Filter-gradient stream.  A sunk filter gradient feeds nothing but the optimizer, so it is off the backward's
    critical path (which is the bwd-data chain): launched on a second HIP stream it runs CONCURRENTLY with the next
    layers' bwd-data kernels and fills their tail waves (and vice versa).  All sunk filter gradients share this one
    stream, so accumulations into a slot keep their program order and the sums stay bit-identical to the one-stream
    schedule.  Inputs are kept alive until side_join() because the caching allocator only orders reuse on one stream."""
    stream = None
    keep = []


SIDE = _Side()


def enable_side_stream(on=True):
    SIDE.stream = torch.cuda.Stream(priority=int(os.environ.get('T2I_SIDE_PRIO', '0'))) if on else None
    SIDE.keep = []


def side_join():
    """Make the current stream wait for every filter gradient issued so far (call before the optimizer reads the arena)."""
    if SIDE.stream is not None:
        torch.cuda.current_stream().wait_stream(SIDE.stream)
        SIDE.keep.clear()


def sunk_launch(launch, keep):
    """Run `launch()` — a sunk filter gradient: it feeds nothing but the optimizer — on the filter-gradient stream if there is one
    (see _Side), else on the current stream.  keep: the tensors it reads, held until side_join()."""
    if SIDE.stream is not None:
        SIDE.stream.wait_stream(torch.cuda.current_stream())      # its operands and the zeroed arena are ready
        K.WS_LANE[0] = 1                                          # its own split-K workspace
        try:
            with torch.cuda.stream(SIDE.stream):
                launch()
        finally:
            K.WS_LANE[0] = 0
        SIDE.keep.append(keep)
    else:
        launch()


def _filter_grad(x, gpre, geom, w, xform=None, xform_plane_rows=0):
    """dw for weight `w`: into its sink if it has one and this is the final backward, else as a differentiable Function.
    xform: the Winograd input transform of `x` the forward conv left behind (kernels.LAST_XFORM), or None; xform_plane_rows: that
    transform belongs to a larger, stacked batch of this many images whose leading images are `x` (stacked.py)."""
    if not torch.is_grad_enabled():
        sink = sink_at(w.data_ptr())
        if sink is not None:
            xs, gs = _c(x), _c(gpre)
            acc = sink_accumulate(w.data_ptr())
            if xs is not x:
                xform = None
            if SIDE.stream is not None:
                SIDE.stream.wait_stream(torch.cuda.current_stream())      # x, gpre and the zeroed arena are ready
                K.WS_LANE[0] = 1                                          # its own split-K workspace
                try:
                    with torch.cuda.stream(SIDE.stream):
                        K.conv_bwd_filter(xs, gs, geom[0], geom[1], out=sink, xform=xform, xform_plane_rows=xform_plane_rows, accumulate=acc)
                finally:
                    K.WS_LANE[0] = 0
                SIDE.keep.append((xs, gs, xform))
            else:
                K.conv_bwd_filter(xs, gs, geom[0], geom[1], out=sink, xform=xform, xform_plane_rows=xform_plane_rows, accumulate=acc)
            _notify(w)
            return None
    if os.environ.get('T2I_DP_DEBUG') == '1' and NOTIFY[0] is not None:
        import sys
        sys.stderr.write('[ag] differentiable filter gradient: grad_enabled=%s sink=%s inputs_only=%s w.requires_grad=%s\n' % (
            torch.is_grad_enabled(), sink_at(w.data_ptr()) is not None, _INPUTS_ONLY[0], w.requires_grad))
    return ConvBwdFilterFn.apply(x, gpre, geom)


def _pair_ok(g, other, w):
    """Final (first-order) backward, sunk filter gradient, bf16 tensors on one stream: the layer's two backward GEMMs may share a
    launch (kernels.conv_bwd_pair).  Everything else keeps the two differentiable Functions."""
    return (not torch.is_grad_enabled() and SIDE.stream is None and K.pair_calls() and g.dtype == torch.bfloat16 and
            other.dtype == torch.bfloat16 and g.is_cuda and sink_at(w.data_ptr()) is not None)


@contextlib.contextmanager
def input_grads_only():
    prev = _INPUTS_ONLY[0]
    _INPUTS_ONLY[0] = True
    try:
        yield
    finally:
        _INPUTS_ONLY[0] = prev


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class ActBwdFn(Function):
    """gy * act'(.) with the derivative read from the activation OUTPUT y.  Linear in gy; piecewise-constant in y for
    lrelu/relu (zero second derivative — only the mask propagates into the double backward)."""

    @staticmethod
    def forward(ctx, gy, y, act, alpha):
        ctx.save_for_backward(y)
        ctx.act, ctx.alpha = act, alpha
        ctx.set_materialize_grads(False)
        return K.act_bwd(_c(gy), y, act, alpha)

    @staticmethod
    def backward(ctx, gg):
        if gg is None:
            return None, None, None, None
        (y,) = ctx.saved_tensors
        return ActBwdFn.apply(gg, y, ctx.act, ctx.alpha), None, None, None


def _act_bwd(gy, y, act, alpha):
    return ActBwdFn.apply(gy, y, act, alpha) if act != K.ACT_NONE else _c(gy)


class ColSumFn(Function):
    """[rows, C] -> [C]  (bias gradients)."""

    @staticmethod
    def forward(ctx, a):
        ctx.shape, ctx.dtype = a.shape, a.dtype
        return K.col_reduce(_c(a))[0]

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).expand(ctx.shape)


class Conv2dFn(Function):
    """y = act(conv(x, w) + b): reference utils/ops.py:58-63.  geom = (ConvDesc, workspace_bytes)."""

    @staticmethod
    def forward(ctx, x, w, b, geom, act, alpha, want_stats=False, out_dtype=None):
        """out_dtype: None = the storage mode's rule (kernels._act_dtype); a gradient piece passes the dtype of the tensor it is the
        gradient of, so that autograd never has to cast (bf16 storage)."""
        d, ws = geom
        x = _c(x)
        ctx.set_materialize_grads(False)   # an undefined upstream gradient must not become a zero-filled conv launch
        # want_stats: a batch norm consumes this output next; the GEMM epilogue leaves it the per-tile column sums
        # the filter gradient of this layer transforms the same x (fp32 Winograd): keep the transform if a gradient will be asked for
        ctx.geom_b = K.bwd_geom(geom)       # the backward GEMMs' descriptor (kernels.math_scope(bwd_math=...)); usually geom itself
        keep = bool(ctx.needs_input_grad[1]) and not _INPUTS_ONLY[0] and ctx.geom_b is geom
        y = (K.conv_fwd_stats(x, w, b, d, ws, act, alpha, keep_xform=keep, out_dtype=out_dtype) if want_stats
             else K.conv_fwd(x, w, b, d, ws, act, alpha, keep_xform=keep, out_dtype=out_dtype))
        ctx.xform = K.LAST_XFORM[0] if keep else None
        K.LAST_XFORM[0] = None
        ctx.save_for_backward(x, w, y if act != K.ACT_NONE else None)
        ctx.geom, ctx.act, ctx.alpha, ctx.has_bias = geom, act, alpha, b is not None
        ctx.bias_ref = b            # only its address is used (gradient sink lookup)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None, None, None, None, None, None, None, None
        x, w, y = ctx.saved_tensors
        gx, gw, gb = conv2d_backward(x, w, y, gy, ctx.geom_b, ctx.act, ctx.alpha, ctx.has_bias, ctx.bias_ref, ctx.needs_input_grad[:3], ctx.xform)
        ctx.xform = None
        return gx, gw, gb, None, None, None, None, None


def conv2d_backward(x, w, y, gy, geom_b, act, alpha, has_bias, bias_ref, need, xform=None, xform_plane_rows=0):
    """Backward of y = act(conv(x, w) + b) for upstream gradient gy -> (gx, gw, gb); need = (x, w, b) wanted.  Shared by Conv2dFn and
    the per-part cases of stacked.SConv2dFn (a slice of a stacked pass is an ordinary pass of its own)."""
    params = not _INPUTS_ONLY[0]
    want_b = has_bias and need[2] and params
    gb = None
    bsink = _sink_of(bias_ref) if want_b else None
    if (want_b and act != K.ACT_NONE and not torch.is_grad_enabled() and gy.shape[-1] % 4 == 0):
        # final (first-order) backward: activation backward and bias gradient in ONE pass over the tensor
        gpre, gb = K.act_bwd_colsum(_c(gy), y, act, alpha, out=bsink)
        if bsink is not None:
            gb = None                       # already summed into the optimizer's arena
            _notify(bias_ref)
    else:
        gpre = _act_bwd(gy, y, act, alpha)
        if want_b and bsink is not None:
            K.col_reduce(_c(gpre), out=bsink)
            _notify(bias_ref)
        elif want_b:
            gb = ColSumFn.apply(gpre)
    if need[0] and need[1] and params and _pair_ok(gpre, x, w):
        # final backward on bf16 tensors: the input gradient and the (sunk) filter gradient in one launch
        gx = K.conv_bwd_pair(K.PAIR_BWD_DATA, _c(gpre), w, _c(x), _c(gpre), geom_b[0], geom_b[1], sink_at(w.data_ptr()), out_dtype=x.dtype,
                             accumulate=sink_accumulate(w.data_ptr()))
        _notify(w)
        return gx, None, gb
    gx = ConvBwdDataFn.apply(gpre, w, None, geom_b, K.ACT_NONE, 0.0, x.dtype) if need[0] else None
    gw = _filter_grad(x, gpre, geom_b, w, xform, xform_plane_rows) if (need[1] and params) else None
    return gx, gw, gb


class ConvBwdDataFn(Function):
    """dx = act(conv^T(dy, w) + b).  As a backward piece: b=None, act=NONE.  As a forward op this is
    tf conv2d_transpose (reference utils/ops.py:66-71) — same kernel, TF deconv filters are already HWIO of the adjoint."""

    @staticmethod
    def forward(ctx, dy, w, b, geom, act, alpha, out_dtype=None):
        d, ws = geom
        dy = _c(dy)
        ctx.set_materialize_grads(False)
        out = K.conv_bwd_data(dy, w, b, d, ws, act, alpha, out_dtype=out_dtype)
        ctx.save_for_backward(dy, w, out if act != K.ACT_NONE else None)
        ctx.geom, ctx.act, ctx.alpha, ctx.has_bias = geom, act, alpha, b is not None
        ctx.geom_b = K.bwd_geom(geom)
        ctx.bias_ref = b
        return out

    @staticmethod
    def backward(ctx, gg):
        if gg is None:
            return None, None, None, None, None, None, None
        dy, w, out = ctx.saved_tensors
        g_dy, g_w, g_b = bwd_data_backward(dy, w, out, gg, ctx.geom_b, ctx.act, ctx.alpha, ctx.has_bias, ctx.bias_ref, ctx.needs_input_grad[:3])
        return g_dy, g_w, g_b, None, None, None, None


def bwd_data_backward(dy, w, out, gg, geom_b, act, alpha, has_bias, bias_ref, need):
    """Backward of out = act(conv^T(dy, w) + b) for upstream gradient gg -> (g_dy, g_w, g_b).  Shared by ConvBwdDataFn and
    stacked.SBwdDataFn (whose double backward runs on the x_hat rows only)."""
    gpre = _act_bwd(gg, out, act, alpha)
    params = not _INPUTS_ONLY[0]
    if need[0] and need[1] and params and _pair_ok(gpre, dy, w):
        g_dy = K.conv_bwd_pair(K.PAIR_FWD, _c(gpre), w, _c(gpre), _c(dy), geom_b[0], geom_b[1], sink_at(w.data_ptr()), out_dtype=dy.dtype,
                               accumulate=sink_accumulate(w.data_ptr()))
        _notify(w)
        g_w = None
    else:
        g_dy = Conv2dFn.apply(gpre, w, None, geom_b, K.ACT_NONE, 0.0, False, dy.dtype) if need[0] else None
        g_w = _filter_grad(gpre, dy, geom_b, w) if (need[1] and params) else None
    g_b = None
    if has_bias and need[2] and params:
        bsink = _sink_of(bias_ref)
        if bsink is not None:
            K.col_reduce(_c(gpre), out=bsink)
            _notify(bias_ref)
        else:
            g_b = ColSumFn.apply(gpre)
    return g_dy, g_w, g_b


class ConvBwdFilterFn(Function):
    """dw = x (*) dy."""

    @staticmethod
    def forward(ctx, x, dy, geom):
        d, ws = geom
        x, dy = _c(x), _c(dy)
        ctx.save_for_backward(x, dy)
        ctx.geom = geom
        return K.conv_bwd_filter(x, dy, d, ws)

    @staticmethod
    def backward(ctx, ggw):
        x, dy = ctx.saved_tensors
        ggw = _c(ggw)
        g_x = ConvBwdDataFn.apply(dy, ggw, None, ctx.geom, K.ACT_NONE, 0.0, x.dtype) if ctx.needs_input_grad[0] else None
        g_dy = Conv2dFn.apply(x, ggw, None, ctx.geom, K.ACT_NONE, 0.0, False, dy.dtype) if ctx.needs_input_grad[1] else None
        return g_x, g_dy, None


class AddActFn(Function):
    """y = act(a + b): the residual joins (reference models/wgancls/model.py:145-146,190-191,206-207)."""

    @staticmethod
    def forward(ctx, a, b, act, alpha):
        ctx.set_materialize_grads(False)
        y = K.add_act(_c(a), _c(b), act, alpha)
        ctx.save_for_backward(y if act != K.ACT_NONE else None)
        ctx.act, ctx.alpha = act, alpha
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None, None, None, None
        (y,) = ctx.saved_tensors
        g = _act_bwd(gy, y, ctx.act, ctx.alpha)
        return g, g, None, None


class ConcatTileFn(Function):
    """[B,H,W,Cf] ++ tile([B,Ce]) -> [B,H,W,Cf+Ce]  (reference models/wgancls/model.py:153-155)."""

    @staticmethod
    def forward(ctx, feat, emb):
        ctx.cf, ctx.ce = feat.shape[-1], emb.shape[-1]
        ctx.set_materialize_grads(False)
        return K.concat_tile_fwd(_c(feat), _c(emb))

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        return ConcatTileBwdFn.apply(g, ctx.cf, ctx.ce)


class ConcatTileBwdFn(Function):
    @staticmethod
    def forward(ctx, g, cf, ce):
        return K.concat_tile_bwd(_c(g), cf, ce)

    @staticmethod
    def backward(ctx, gg_feat, gg_emb):
        return ConcatTileFn.apply(gg_feat, gg_emb), None, None


class NchwToNhwcFn(Function):
    """physical [B,C,H,W] -> physical [B,H,W,C] (reference utils/ops.py:132-134 to_nhwc)."""

    @staticmethod
    def forward(ctx, x):
        return K.nchw_to_nhwc(_c(x))

    @staticmethod
    def backward(ctx, g):
        return NhwcToNchwFn.apply(g)


class NhwcToNchwFn(Function):
    @staticmethod
    def forward(ctx, x):
        return K.nhwc_to_nchw(_c(x))

    @staticmethod
    def backward(ctx, g):
        return NchwToNhwcFn.apply(g)


class ActFn(Function):
    """y = act(x) as a standalone op (first-order for tanh, any order for lrelu/relu)."""

    @staticmethod
    def forward(ctx, x, act, alpha):
        y = K.act_fwd(_c(x), act, alpha)
        ctx.save_for_backward(y)
        ctx.act, ctx.alpha = act, alpha
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return ActBwdFn.apply(gy, y, ctx.act, ctx.alpha), None, None


_BN_ONE_ENTRY = [os.environ.get('T2I_BN_ONE_ENTRY', '1') != '0']     # 0: the separate statistics / normalise calls of rounds 1-4


class BatchNormTrainFn(Function):
    """Training-mode fused batch norm + activation: reference utils/ops.py:7-29.  Normalises with the biased batch
    variance; if moving_mean/var are given they are updated in place with the unbiased one (TF UPDATE_OPS semantics are
    decided by the caller).  Generator only => first order."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, decay, act, alpha, moving_updates=1):
        x = _c(x)
        C = x.shape[-1]
        n = x.numel() // C
        # statistics (the producing conv's epilogue partials when ops.conv2d(..., stats=True) left any, else a pass over x;
        # numerically stable either way) and the finalize step in one chain of launches
        # round 5: one entry point — [first stage unless the conv left tile partials] -> [second stage + finalize] -> [normalise], the
        # middle launch folded into the last one's prologue for the small tensors (t2i_bn_train_fwd_grouped with groups = 1)
        if _BN_ONE_ENTRY[0] and C % 4 == 0 and x.data_ptr() % 16 == 0 and gamma.data_ptr() % 4 == 0:
            y, mean, rstd = K.bn_train_fwd_grouped(x, gamma, beta, eps, decay, 1, act, alpha, moving_mean, moving_var, moving_updates)
            mean, rstd = mean[0], rstd[0]
        else:
            for _ in range(moving_updates - 1 if moving_mean is not None else 0):      # (rare path: the statistics pass again per extra update)
                K.bn_train_stats(x, gamma, beta, eps, decay, moving_mean, moving_var)
            mean, rstd, scale, shift = K.bn_train_stats(x, gamma, beta, eps, decay, moving_mean, moving_var)
            y = K.bn_apply(x, scale, shift, act, alpha)
        ctx.save_for_backward(x, gamma, mean, rstd, y if act != K.ACT_NONE else None)
        ctx.act, ctx.alpha = act, alpha
        ctx.gamma_ref, ctx.beta_ref = gamma, beta
        ctx.mark_non_differentiable(mean, rstd)
        ctx.set_materialize_grads(False)     # else the engine zero-fills a gradient for mean and rstd on every backward
        return y, mean, rstd

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gm, _gr):
        if gy is None:
            return (None,) * 10
        x, gamma, mean, rstd, y = ctx.saved_tensors
        gy = _c(gy)
        fused = gy.shape[-1] % 4 == 0 and all(t.data_ptr() % 16 == 0 for t in (gy, x, mean))
        if not fused:
            if ctx.act != K.ACT_NONE:
                gy = K.act_bwd(gy, y, ctx.act, ctx.alpha)
            sum_dy, sum_dy_x = K.col_reduce(gy, x, True, center=mean)     # sum dy, sum dy * (x - mean)
        # gamma / beta that do not require a gradient (a critic with batch norm run under store.frozen() in the generator
        # step: detached views that share the real parameters' addresses) must neither reach the sinks nor be announced
        want_g, want_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        gsink = _sink_of(ctx.gamma_ref) if want_g else None
        bsink = _sink_of(ctx.beta_ref) if want_b else None
        sunk = gsink is not None and bsink is not None
        if fused and _BN_ONE_ENTRY[0]:      # [act backward + both reductions] -> [second stage + coefficients + dx]: two launches when the partials are few
            dx, dgamma, dbeta = K.bn_bwd_grouped(gy, y if ctx.act != K.ACT_NONE else None, x, mean, rstd, gamma, 1, ctx.act, ctx.alpha,
                                                 dgamma_out=gsink if sunk else None, dbeta_out=bsink if sunk else None)
        elif fused:          # [act backward + both reductions] -> [second stage + coefficients] -> [dx]: three launches
            dx, dgamma, dbeta = K.bn_bwd_fused(gy, y if ctx.act != K.ACT_NONE else None, x, mean, rstd, gamma, ctx.act, ctx.alpha,
                                               dgamma_out=gsink if sunk else None, dbeta_out=bsink if sunk else None)
        else:
            dx, dgamma, dbeta = K.bn_bwd(gy, x, mean, rstd, gamma, sum_dy, sum_dy_x, dgamma_out=gsink if sunk else None,
                                         dbeta_out=bsink if sunk else None)
        if sunk:
            _notify(ctx.gamma_ref); _notify(ctx.beta_ref)
            return dx, None, None, None, None, None, None, None, None, None
        return dx, (dgamma if want_g else None), (dbeta if want_b else None), None, None, None, None, None, None, None


class BatchNormTrainGroupedFn(Function):
    """Training-mode batch norm of a BATCHED pass whose `groups` equal slices along the batch axis are separate passes of the
    reference graph (models/gancls/model.py:48-51: the critic on fake / match / mismatch images, each with its own batch statistics):
    statistics, normalisation and backward per slice, while every convolution around it runs once on the whole batch.  Moving averages
    move once per slice, in slice order (what three sequential passes did); dgamma / dbeta are summed over the slices.  Three launches
    forward, three backward for all slices (t2i_bn_train_fwd_grouped / t2i_bn_bwd_grouped; C % 4 == 0 and 16-byte alignment — otherwise
    the slices go through the ordinary kernels one by one, forward AND backward: tests/test_kernels_gpu.py::
    test_grouped_batch_norm_odd_channels).  First order only."""

    @staticmethod
    def _fast(x, groups):
        b = x.shape[0] // groups
        return x.shape[-1] % 4 == 0 and x.data_ptr() % 16 == 0 and (x[:b].numel() * x.element_size()) % 16 == 0

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, decay, act, alpha, groups):
        x = _c(x)
        Bt = x.shape[0]
        assert Bt % groups == 0, (Bt, groups)
        b = Bt // groups
        ctx.fast = BatchNormTrainGroupedFn._fast(x, groups)
        if ctx.fast:
            y, mean, rstd = K.bn_train_fwd_grouped(x, gamma, beta, eps, decay, groups, act, alpha, moving_mean, moving_var)
            stats = [mean, rstd]
        else:
            stats, scales, shifts = [], [], []
            for g in range(groups):
                mean, rstd, scale, shift = K.bn_train_stats(x[g * b:(g + 1) * b], gamma, beta, eps, decay, moving_mean, moving_var)
                stats += [mean, rstd]
                scales.append(scale); shifts.append(shift)
            y = K.bn_apply_groups(x, scales, shifts, act, alpha)
        ctx.save_for_backward(x, gamma, y if act != K.ACT_NONE else None, *stats)
        ctx.act, ctx.alpha, ctx.groups = act, alpha, groups
        ctx.gamma_ref, ctx.beta_ref = gamma, beta
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 10
        x, gamma, y = ctx.saved_tensors[:3]
        stats = ctx.saved_tensors[3:]
        gy = _c(gy)
        groups = ctx.groups
        b = x.shape[0] // groups
        want_g, want_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        gsink = _sink_of(ctx.gamma_ref) if want_g else None
        bsink = _sink_of(ctx.beta_ref) if want_b else None
        sunk = gsink is not None and bsink is not None
        if ctx.fast and gy.data_ptr() % 16 == 0:
            dx, dgamma, dbeta = K.bn_bwd_grouped(gy, y if ctx.act != K.ACT_NONE else None, x, stats[0], stats[1], gamma, groups, ctx.act, ctx.alpha,
                                                 dgamma_out=gsink if sunk else None, dbeta_out=bsink if sunk else None)
        else:
            if ctx.fast:                 # per-group views of the [groups, C] statistics
                stats = [t[g] for g in range(groups) for t in (stats[0], stats[1])]
            vec = gy.shape[-1] % 4 == 0 and all(t.data_ptr() % 16 == 0 for t in (gy, x)) and (x[:b].numel() * x.element_size()) % 16 == 0
            dx = torch.empty_like(x)
            dgamma = dbeta = None
            for g in range(groups):
                sl = slice(g * b, (g + 1) * b)
                if vec:
                    _, dg, db = K.bn_bwd_fused(gy[sl], y[sl] if ctx.act != K.ACT_NONE else None, x[sl], stats[2 * g], stats[2 * g + 1], gamma, ctx.act,
                                               ctx.alpha, dgamma_out=gsink if sunk else None, dbeta_out=bsink if sunk else None, out=dx[sl])
                else:
                    # a channel count that is not a multiple of 4 (an odd DF_DIM), or unaligned slices: the scalar kernels, slice by slice
                    # — activation backward, the two column reductions about the slice's mean, then the batch-norm backward itself
                    gs = K.act_bwd(_c(gy[sl]), _c(y[sl]), ctx.act, ctx.alpha) if ctx.act != K.ACT_NONE else _c(gy[sl])
                    xs = _c(x[sl])
                    sum_dy, sum_dy_x = K.col_reduce(gs, xs, True, center=stats[2 * g])
                    dxs, dg, db = K.bn_bwd(gs, xs, stats[2 * g], stats[2 * g + 1], gamma, sum_dy, sum_dy_x,
                                           dgamma_out=gsink if sunk else None, dbeta_out=bsink if sunk else None)
                    dx[sl].copy_(dxs)
                if not sunk:
                    dgamma = dg if dgamma is None else dgamma + dg
                    dbeta = db if dbeta is None else dbeta + db
        if sunk:
            _notify(ctx.gamma_ref); _notify(ctx.beta_ref)
            return (dx,) + (None,) * 9
        return (dx, dgamma if want_g else None, dbeta if want_b else None) + (None,) * 7


class GpSlopesFn(Function):
    """slopes[b] = ||g[b]||_2 (reference models/wgancls/model.py:64,69).  Its backward feeds the double backward of the
    critic; it is itself only differentiated once."""

    @staticmethod
    def forward(ctx, g):
        g = _c(g)
        s = K.gp_slopes(g)
        ctx.save_for_backward(g, s)
        return s

    @staticmethod
    @once_differentiable
    def backward(ctx, ds):
        g, s = ctx.saved_tensors
        # coef_b = where(s > 0, ds / clamp_min(s, 1e-30), 0), formed inside the kernel (five tensor-library launches per term before)
        return K.row_scale_div(g, _c(ds), s)


# ---- PGGAN operators (reference utils/ops.py:74-81,100-101,109-111) ------------------------------------------------------
class Pool2Fn(Function):
    """scale * 2x2 window sum, stride 2 (scale = 1/4: tf.nn.pool AVG SAME on even extents).  Linear; its adjoint is the
    nearest-neighbour replication with the same scale, so the pair is closed under differentiation of any order (the
    critic of PGGAN is differentiated twice by the gradient penalty)."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        ctx.set_materialize_grads(False)
        return K.pool2_sum(_c(x), scale)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        return Upscale2Fn.apply(g, ctx.scale), None


class Upscale2Fn(Function):
    """scale * nearest-neighbour x2 (scale = 1: ops.upscale)."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        ctx.set_materialize_grads(False)
        return K.upscale2(_c(x), scale)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        return Pool2Fn.apply(g, ctx.scale), None


class AxpbyFn(Function):
    """alpha*a + beta*b with host scalars: the fade-in mix of a new resolution (reference models/pggan/pggan.py:267,314)."""

    @staticmethod
    def forward(ctx, a, alpha, b, beta):
        ctx.alpha, ctx.beta, ctx.has_b = alpha, beta, b is not None
        ctx.set_materialize_grads(False)
        return K.axpby(_c(a), alpha, _c(b) if b is not None else None, beta)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        ga = AxpbyFn.apply(g, ctx.alpha, None, 0.0) if ctx.needs_input_grad[0] else None
        gb = AxpbyFn.apply(g, ctx.beta, None, 0.0) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        return ga, None, gb, None


class LayerNormFn(Function):
    """tf.contrib.layers.layer_norm(x, begin_norm_axis=1, begin_params_axis=-1) + activation (reference utils/ops.py:74-81):
    each sample is normalised over all of its elements (biased variance, eps = 1e-12), then scaled and shifted per
    last-axis channel.  Generator only in the reference (models/pggan/pggan.py:289-309) => first order."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act, alpha):
        x = _c(x)
        B = x.shape[0]
        n = x.numel() // B
        s1, s2 = K.row_moments(x)
        mean = s1 / n
        var = torch.clamp(s2 / n - mean * mean, min=0.0)
        rstd = torch.rsqrt(var + eps)
        xhat = K.row_fma2(x, rstd, delta=-mean * rstd)
        y = K.bn_apply(xhat, gamma, beta, act, alpha)               # per-channel affine + activation
        ctx.save_for_backward(xhat, rstd, gamma, y if act != K.ACT_NONE else None)
        ctx.act, ctx.alpha, ctx.n = act, alpha, n
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xhat, rstd, gamma, y = ctx.saved_tensors
        gy = _c(gy)
        if ctx.act != K.ACT_NONE:
            gy = K.act_bwd(gy, y, ctx.act, ctx.alpha)
        dbeta, dgamma = K.col_reduce(gy, xhat, True)                  # sum gy, sum gy * xhat over rows of [*, C]
        g = K.bn_apply(gy, gamma, torch.zeros_like(gamma), K.ACT_NONE, 0.0)
        t1, t2 = K.row_moments(g, xhat)
        dx = K.row_fma2(g, rstd, xhat, -rstd * t2 / ctx.n, -rstd * t1 / ctx.n)
        return dx, dgamma, dbeta, None, None, None


class CaSampleKlFn(Function):
    """(code, kl) = (mean + exp(log_sigma)*eps, KL(N(mean, sigma) || N(0,1)) averaged over all elements): the conditioning
    augmentation of reference models/wgancls/model.py:117-127 as one forward and one backward launch."""

    @staticmethod
    def forward(ctx, mean, log_sigma, eps):
        mean, log_sigma, eps = _c(mean), _c(log_sigma), _c(eps)
        ctx.save_for_backward(mean, log_sigma, eps)
        ctx.set_materialize_grads(False)
        code, kl = K.ca_kl_fwd(mean, log_sigma, eps)
        return code, kl

    @staticmethod
    @once_differentiable
    def backward(ctx, dcode, dkl):
        mean, log_sigma, eps = ctx.saved_tensors
        if dcode is None and dkl is None:
            return None, None, None
        dmean, dls = K.ca_kl_bwd(mean, log_sigma, eps, _c(dcode) if dcode is not None else None,
                                 _c(dkl).reshape(1) if dkl is not None else None)
        return dmean, dls, None


class LerpDevFn(Function):
    """mode 0: (1-t)*a + t*b;  1: t*a;  2: (1-t)*a, t in device memory (graph-replayable fade-in).  Closed under
    differentiation: the gradients are modes 2 and 1 of the incoming gradient."""

    @staticmethod
    def forward(ctx, a, b, t_dev, mode):
        ctx.t_dev, ctx.mode = t_dev, mode
        ctx.set_materialize_grads(False)
        return K.lerp_dev(_c(a), _c(b) if b is not None else None, t_dev, mode)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        if ctx.mode == 0:
            ga = LerpDevFn.apply(g, None, ctx.t_dev, 2) if ctx.needs_input_grad[0] else None
            gb = LerpDevFn.apply(g, None, ctx.t_dev, 1) if ctx.needs_input_grad[1] else None
            return ga, gb, None, None
        return (LerpDevFn.apply(g, None, ctx.t_dev, ctx.mode) if ctx.needs_input_grad[0] else None), None, None, None
