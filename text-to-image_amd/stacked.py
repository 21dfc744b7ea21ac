"""The critic's four passes of one step as ONE stacked pass (round 6): [ D(G) | D(x) | D(x_mismatch) ] ++ [ D(x_hat) ].

reference models/wgancls/model.py:48-55,62-70: the critic runs on G, x and x_mismatch (their logits enter D_loss) and on the
interpolate x_hat (whose logits only enter through the gradient penalty ||d D(x_hat) / d x_hat||).  Nothing orders the four
passes; the critic has no batch norm, so its samples are independent.  Rounds 1-5 ran [G | x | x_mis] as one pass of 3B images
and x_hat as a second pass of B — every layer twice forward and twice in the first-order input-gradient chain.  Here every
kernel of the forward pass and of the first-order chain runs ONCE on 4B images:

  forward              y4 = conv(x4)                                   one launch per layer (4B)
  first-order chain    gx4 = conv^T(gpre4), gpre4 = gy4 * act'(y4)      one launch each per layer (4B); the upstream gradient is
                       [ dD_loss/dlogits (3B rows) | ones (B rows) ]: the loss gradient of the three scored passes and the
                       gradient-penalty's  d sum(D(x_hat)) / d(.)  travel down the layers side by side
  filter gradients     from the first 3B rows only (D(x_hat) is not in the loss), into the optimizer's arena
  double backward      differentiates the chain of the LAST B rows only (the penalty) — B-sized, as before

A stacked tensor is two torch tensors, `main` (rows [0, R)) and `hat` (rows [R, 4B)), that are adjacent slices of one buffer;
the Functions below take and return both, launch on the whole buffer, and give autograd two outputs — so the graph the double
backward walks is the x_hat slice's alone and nothing is ever zero-padded to 4B.  If the two parts of an argument are not
adjacent in memory (never the case on the path above) they are concatenated: correct, one copy slower.

The data-parallel schedules run the same stacked critic step, uncut (WGanCls._d_cut_ranges).

The same machinery stacks other evaluations that share weights: the generator's two evaluations of a wgancls
iteration (SBatchNormFn: per-evaluation statistics, moving_groups), and the three critic evaluations of the generator step of the
sigmoid-CE trainers (gancls, StackGAN: fake with the gradient | match | mismatch)."""
import torch
from torch.autograd import Function

from . import autograd as A
from . import kernels as K

_FIRST = [0]          # inside the stacked first-order pass: the number of main rows R; else 0


class first_order_pass(object):
    """with first_order_pass(R): torch.autograd.grad(...) — the backward that runs inside is the stacked first-order chain: rows
    [0, R) carry the loss gradient (their filter / bias gradients go to the sinks, nothing of theirs is differentiated again), the
    rows behind carry the gradient penalty's chain (input gradients only, differentiable)."""

    def __init__(self, rows):
        self.rows = int(rows)

    def __enter__(self):
        self.prev, _FIRST[0] = _FIRST[0], self.rows

    def __exit__(self, *a):
        _FIRST[0] = self.prev


class Stacked(object):
    """(main, hat): adjacent slices along axis 0 of one buffer, seen by utils/ops.py as one tensor of main + hat rows."""
    __slots__ = ('main', 'hat')

    def __init__(self, main, hat):
        assert main.shape[1:] == hat.shape[1:] and main.dtype == hat.dtype, (tuple(main.shape), tuple(hat.shape))
        self.main, self.hat = main, hat

    @property
    def shape(self):
        return torch.Size((self.main.shape[0] + self.hat.shape[0],) + tuple(self.main.shape[1:]))

    @property
    def device(self):
        return self.main.device

    @property
    def dtype(self):
        return self.main.dtype

    def dim(self):
        return self.main.dim()

    def permute(self, *dims):
        return Stacked(self.main.permute(*dims), self.hat.permute(*dims))

    def detach(self):
        return Stacked(self.main.detach(), self.hat.detach())

    def reshape_parts(self, *tail):
        """each part reshaped to (its rows,) + tail"""
        return Stacked(self.main.reshape((self.main.shape[0],) + tuple(tail)), self.hat.reshape((self.hat.shape[0],) + tuple(tail)))


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def full(a, b):
    """The whole buffer behind the parts (a detached view; no copy) — or their concatenation if they are not adjacent."""
    ba = a._base
    if (ba is not None and ba is b._base and not ba.requires_grad and ba.is_contiguous() and a.is_contiguous() and b.is_contiguous() and
            ba.dtype == a.dtype == b.dtype and ba.dim() == a.dim() and ba.shape[1:] == a.shape[1:] == b.shape[1:] and
            ba.shape[0] == a.shape[0] + b.shape[0] and ba.data_ptr() == a.data_ptr() and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size()):
        return ba            # the producing Function's own output tensor: the object that carries the bf16 twin its kernel wrote (kernels._twin_keep)
    a, b = _c(a.detach()), _c(b.detach())
    if (a.dtype == b.dtype and a.shape[1:] == b.shape[1:] and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and
            b.data_ptr() == a.data_ptr() + a.numel() * a.element_size()):
        return a.as_strided((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride())
    return torch.cat([a, b], 0)


def _both(gm, gh):
    return gm is not None and gh is not None and _FIRST[0] > 0


class SActBwdFn(Function):
    """(gy * act'(y)) on a stacked pair.  Differentiated again on the hat rows only (autograd.ActBwdFn there).
    colsum_into (a bias slot of the gradient arena, or None): also ACCUMULATE the column sums of the MAIN rows' result into it — the
    bias gradient of a conv layer, formed in the pass that writes those rows (t2i_act_bwd_colsum) while the hat rows take the plain
    kernel; without it the whole 4B buffer is one launch."""

    @staticmethod
    def forward(ctx, gym, gyh, y4, act, alpha, colsum_into=None):
        R = gym.shape[0]
        gy4 = full(gym, gyh)
        if colsum_into is not None:
            out4 = torch.empty_like(gy4)
            K.act_bwd_colsum(gy4[:R], y4[:R], act, alpha, out=colsum_into, dx_out=out4[:R])
            K.act_bwd(gy4[R:], y4[R:], act, alpha, out=out4[R:])
        else:
            out4 = K.act_bwd(gy4, y4, act, alpha)
        ctx.save_for_backward(y4)
        ctx.R, ctx.act, ctx.alpha = R, act, alpha
        ctx.set_materialize_grads(False)
        return out4[:R], out4[R:]

    @staticmethod
    def backward(ctx, ggm, ggh):
        (y4,) = ctx.saved_tensors
        R = ctx.R
        gm = A.ActBwdFn.apply(ggm, y4[:R], ctx.act, ctx.alpha) if ggm is not None else None
        gh = None
        if ggh is not None:
            yh = y4[R:]
            if _TANGENT_IN_PLACE[0] and _DEFER[0] and not torch.is_grad_enabled() and ctx.act in (K.ACT_LRELU, K.ACT_RELU) and \
                    ggh.dtype == yh.dtype and K._twin_for(yh, ggh) is None:
                # The double backward of the gradient penalty (no third order): the result is the tangent that the layer ABOVE writes over
                # the x_hat rows of its input for its one filter-gradient launch (SBwdDataFn.backward) — and its input IS this y4.  The
                # mask read here was those rows' last use as activations, and the kernel is elementwise, so it writes the tangent where it
                # is wanted: one copy launch less per layer, same bits.
                gh = K.act_bwd(_c(ggh), yh, ctx.act, ctx.alpha, out=yh)
                K._drop_image(y4)
            else:
                gh = A.ActBwdFn.apply(ggh, yh, ctx.act, ctx.alpha)
        return gm, gh, None, None, None, None


def _sact_bwd(gym, gyh, y4, act, alpha, colsum_into=None):
    return SActBwdFn.apply(gym, gyh, y4, act, alpha, colsum_into) if act != K.ACT_NONE else (gym, gyh)


# Filter gradients of the stacked step that wait for the double backward (SBwdDataFn): [x4, gp4, w, geometry, kept transform, done].
# flush_deferred() — called by the model after the double backward — launches whatever the double backward did not reach, from the
# scored rows alone, so that no layer can lose its loss gradient to a pruned graph.
_DEFERRED = []
_DEFER = [True]

# The deferred filter gradients on a stream of their own (T2I_STACK_SIDE=1): inside the double backward the critical path is the
# tangent chain — per layer a B-row conv, an activation backward and a copy, kernels that leave most of the chip idle — while the 4B-row
# filter gradients feed nothing but the optimizer.  Issued on a second stream they fill the chip beside the chain instead of
# alternating with it.  join() makes the calling stream wait for them (before Adam reads the arena).
import os as _os
_SIDE = {'on': _os.environ.get('T2I_STACK_SIDE', '0') == '1', 'stream': None, 'keep': []}
# SActBwdFn.backward writes the penalty's tangent straight over the x_hat rows of the activation it masks with (= the next layer's input,
# where SBwdDataFn.backward wants it); 0: a fresh tensor + the copy, as before — same bits, nine launches more
_TANGENT_IN_PLACE = [_os.environ.get('T2I_TANGENT_IN_PLACE', '1') != '0']


def side_filter_gradients(on):
    prev, _SIDE['on'] = _SIDE['on'], bool(on)
    return prev


def prepare_side(device):
    """Create the stream and its workspace lane (outside any capture; the lane is sized like the main lane is NOW)."""
    if _SIDE['on']:
        if _SIDE['stream'] is None:
            _SIDE['stream'] = torch.cuda.Stream(device=device)
        K.stream_lane(_SIDE['stream'], device)


def join():
    s = _SIDE['stream']
    if s is not None and _SIDE['keep']:
        torch.cuda.current_stream().wait_stream(s)
        del _SIDE['keep'][:]


def defer_filter_gradients(on):
    prev, _DEFER[0] = _DEFER[0], bool(on)
    return prev


def flush_deferred():
    recs, _DEFERRED[:] = list(_DEFERRED), []
    for rec in recs:
        if not rec['done']:
            R = rec['R']
            with torch.no_grad():
                A._filter_grad(rec['x4'][:R], rec['gp4'][:R], K.rebatch(rec['geom4'], R), rec['w'], None)
            rec['done'] = True


class SBwdDataFn(Function):
    """conv^T(gpre, w) on a stacked pair in one launch: the first-order input-gradient chain.  Its backward — the double backward of
    the gradient penalty — runs on the hat rows only: exactly autograd.ConvBwdDataFn's, on their batch.

    defer (a record of _DEFERRED, or None): the layer's filter gradient has NOT been launched by the first-order pass.  The loss term
    sum_{scored rows} x (*) gpre and the penalty term  tangent (*) gpre_hat  (tangent = this backward's incoming gradient, the
    derivative of the penalty with respect to the layer's input-gradient, B rows) are both "rows of an input (*) rows of gpre4": the
    tangent is written over the x_hat rows of the layer's input — dead by now: their only later reader was the activation mask of the
    layer below, whose double backward has already run — and ONE filter-gradient launch over all 4B rows produces the sum.  The
    forward's kept Winograd transform is reused for the scored rows; the library regenerates the x_hat rows' tiles (xform_valid_rows)."""

    @staticmethod
    def forward(ctx, gpm, gph, w, geom4, out_dtype, defer):
        R = gpm.shape[0]
        gp4 = full(gpm, gph)
        d, ws = geom4
        out4 = K.conv_bwd_data(gp4, w, None, d, ws, K.ACT_NONE, 0.0, out_dtype=out_dtype)
        ctx.save_for_backward(gp4, w)
        ctx.R, ctx.geom4, ctx.defer = R, geom4, defer
        if defer is not None:
            defer['gp4'] = gp4
        ctx.set_materialize_grads(False)
        return out4[:R], out4[R:]

    @staticmethod
    def backward(ctx, ggm, ggh):
        gp4, w = ctx.saved_tensors
        R = ctx.R
        if ggm is not None:
            raise NotImplementedError('the main rows of a stacked first-order chain are not differentiated again')
        if ggh is None:
            return None, None, None, None, None, None
        gh = K.rebatch(ctx.geom4, gp4.shape[0] - R)
        rec = ctx.defer
        if rec is None or rec['done'] or A._INPUTS_ONLY[0] or torch.is_grad_enabled():
            g_dy, g_w, _ = A.bwd_data_backward(gp4[R:], w, None, ggh, gh, K.ACT_NONE, 0.0, False, None,
                                               (ctx.needs_input_grad[1], ctx.needs_input_grad[2], False))
            return None, g_dy, g_w, None, None, None
        g_dy = A.Conv2dFn.apply(ggh, w, None, gh, K.ACT_NONE, 0.0, False, gp4.dtype) if ctx.needs_input_grad[1] else None
        x4, d4 = rec['x4'], ctx.geom4[0]
        tang = _c(ggh)
        if tang.data_ptr() != x4[R:].data_ptr():              # (already there: SActBwdFn.backward of the layer below wrote it in place)
            K.axpby(tang, 1.0, out=x4[R:])                    # the tangent over the (dead) x_hat rows of the layer's input
        K._drop_image(x4)
        xf, sink, ws4 = rec['xform'], rec['sink'], ctx.geom4[1]
        acc = A.sink_accumulate(w.data_ptr())
        launch = lambda: K.conv_bwd_filter(x4, gp4, d4, ws4, out=sink, xform=xf, xform_valid_rows=R if xf is not None else 0, accumulate=acc)
        side = _SIDE['stream'] if (_SIDE['on'] and A.SIDE.stream is None) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())       # the tangent is in place; everything else it reads was final long ago
            with torch.cuda.stream(side):
                launch()
            _SIDE['keep'].append((x4, gp4, xf, tang))
        else:
            A.sunk_launch(launch, (x4, gp4, xf))
        A._notify(w)
        rec['done'] = True
        rec['x4'] = rec['gp4'] = rec['xform'] = None
        return None, g_dy, None, None, None, None


class SConv2dFn(Function):
    """y = act(conv(x, w) + b) on a stacked pair in one launch (also the dense layers: 1x1 on [B,1,1,C])."""

    @staticmethod
    def forward(ctx, xm, xh, w, b, geom4, act, alpha, want_stats=False):
        R = xm.shape[0]
        x4 = full(xm, xh)
        d, ws = geom4
        assert d.B == x4.shape[0], (d.B, tuple(x4.shape))
        ctx.set_materialize_grads(False)
        ctx.geom_b4 = K.bwd_geom(geom4)        # the backward GEMMs' descriptor, decided inside the network's math scope (as Conv2dFn does)
        # the layer's filter gradient transforms the same x (fp32 Winograd): keep the transform — for the stacked critic step's deferred
        # launch over all rows, or for the leading part's own filter gradient (the generator pair: xform_plane_rows)
        keep = bool(ctx.needs_input_grad[2]) and not A._INPUTS_ONLY[0] and ctx.geom_b4 is geom4 and (
            (bool(ctx.needs_input_grad[0]) and _DEFER[0]) or not ctx.needs_input_grad[1])
        y4 = (K.conv_fwd_stats if want_stats else K.conv_fwd)(x4, w, b, d, ws, act, alpha, keep_xform=keep)
        ctx.xform = K.LAST_XFORM[0] if keep else None
        K.LAST_XFORM[0] = None
        ctx.save_for_backward(x4, w, y4 if act != K.ACT_NONE else None)
        ctx.R, ctx.act, ctx.alpha, ctx.has_bias, ctx.bias_ref = R, act, alpha, b is not None, b
        return y4[:R], y4[R:]

    @staticmethod
    def backward(ctx, gym, gyh):
        if gym is None and gyh is None:
            return (None,) * 8
        x4, w, y4 = ctx.saved_tensors
        R, B4 = ctx.R, x4.shape[0]
        need_xm, need_xh, need_w, need_b = ctx.needs_input_grad[:4]
        geo_m, geo_h = K.rebatch(ctx.geom_b4, R), K.rebatch(ctx.geom_b4, B4 - R)
        if not _both(gym, gyh):
            # one part only (the two-pass order of rounds 1-5, or a caller outside the stacked pass): each slice is an ordinary pass
            gxm = gxh = gw = gb = None
            for g, sl, geom, nx in ((gym, slice(0, R), geo_m, need_xm), (gyh, slice(R, B4), geo_h, need_xh)):
                if g is None:
                    continue
                # the leading part may read the kept transform of the whole stacked batch (its tiles lead every plane)
                xf = ctx.xform if sl.start == 0 else None
                gx, gw1, gb1 = A.conv2d_backward(x4[sl], w, y4[sl] if y4 is not None else None, g, geom, ctx.act, ctx.alpha,
                                                 ctx.has_bias, ctx.bias_ref, (nx, need_w, need_b), xf, B4 if xf is not None else 0)
                if sl.start == 0:
                    gxm = gx
                else:
                    gxh = gx
                gw = gw1 if gw is None else (gw if gw1 is None else gw + gw1)
                gb = gb1 if gb is None else (gb if gb1 is None else gb + gb1)
            ctx.xform = None
            return gxm, gxh, gw, gb, None, None, None, None
        # ---- the stacked first-order pass: one activation backward, one input-gradient launch for all 4B rows
        bsink = A.sink_at(ctx.bias_ref.data_ptr()) if (ctx.has_bias and need_b) else None
        fuse_b = bsink is not None and ctx.act != K.ACT_NONE and gym.shape[-1] % 4 == 0      # bias gradient in the activation backward's pass
        gpm, gph = _sact_bwd(gym, gyh, y4, ctx.act, ctx.alpha, bsink if fuse_b else None)
        if fuse_b:
            A._notify(ctx.bias_ref)
        gxm = gxh = None
        rec = None
        if need_xm:
            sink = A.sink_at(w.data_ptr()) if (need_w and _DEFER[0]) else None
            if sink is not None and gpm.dtype == x4.dtype:
                # the filter gradient waits for the double backward: ONE launch over all 4B rows there (SBwdDataFn)
                rec = {'x4': x4, 'gp4': None, 'w': w, 'geom4': ctx.geom_b4, 'xform': ctx.xform, 'sink': sink, 'R': R, 'done': False}
                _DEFERRED.append(rec)
            gxm, gxh = SBwdDataFn.apply(gpm, gph, w, ctx.geom_b4, x4.dtype, rec)
        elif need_xh:          # the first layer: its main rows are images, nobody wants their gradient
            gxh = A.ConvBwdDataFn.apply(gph, w, None, geo_h, K.ACT_NONE, 0.0, x4.dtype)
        # ---- parameters: from the main rows only, and never differentiated again (D(x_hat) is not in the loss; the penalty reaches
        # the filters through the double backward of the chain above)
        gw = gb = None
        with torch.no_grad():
            gpm_d = _c(gpm.detach())
            if ctx.has_bias and need_b and not fuse_b:
                if bsink is not None:
                    K.col_reduce(gpm_d, out=bsink)
                    A._notify(ctx.bias_ref)
                else:
                    gb = K.col_reduce(gpm_d)[0]
            if need_w and rec is None:
                gw = A._filter_grad(x4[:R], gpm_d, geo_m, w, None)
        ctx.xform = None
        return gxm, gxh, gw, gb, None, None, None, None


class SAddActFn(Function):
    """act(a + b) on stacked pairs (the critic's residual join)."""

    @staticmethod
    def forward(ctx, am, ah, bm, bh, act, alpha):
        R = am.shape[0]
        ctx.set_materialize_grads(False)
        y4 = K.add_act(full(am, ah), full(bm, bh), act, alpha)
        ctx.save_for_backward(y4 if act != K.ACT_NONE else None)
        ctx.R, ctx.act, ctx.alpha = R, act, alpha
        return y4[:R], y4[R:]

    @staticmethod
    def backward(ctx, gym, gyh):
        (y4,) = ctx.saved_tensors
        R = ctx.R
        if _both(gym, gyh):
            gm, gh = _sact_bwd(gym, gyh, y4, ctx.act, ctx.alpha)
        else:
            gm = A._act_bwd(gym, y4[:R] if y4 is not None else None, ctx.act, ctx.alpha) if gym is not None else None
            gh = A._act_bwd(gyh, y4[R:] if y4 is not None else None, ctx.act, ctx.alpha) if gyh is not None else None
        return gm, gh, gm, gh, None, None


class SConcatTileFn(Function):
    """[B,H,W,Cf] ++ tile([B,Ce]) on stacked pairs."""

    @staticmethod
    def forward(ctx, fm, fh, em, eh):
        R = fm.shape[0]
        ctx.cf, ctx.ce, ctx.R = fm.shape[-1], em.shape[-1], R
        ctx.set_materialize_grads(False)
        out4 = K.concat_tile_fwd(full(fm, fh), full(em, eh))
        return out4[:R], out4[R:]

    @staticmethod
    def backward(ctx, gm, gh):
        if _both(gm, gh):
            return SConcatTileBwdFn.apply(gm, gh, ctx.cf, ctx.ce)
        dfm = dem = dfh = deh = None
        if gm is not None:
            dfm, dem = A.ConcatTileBwdFn.apply(gm, ctx.cf, ctx.ce)
        if gh is not None:
            dfh, deh = A.ConcatTileBwdFn.apply(gh, ctx.cf, ctx.ce)
        return dfm, dfh, dem, deh


class SConcatTileBwdFn(Function):
    @staticmethod
    def forward(ctx, gm, gh, cf, ce):
        R = gm.shape[0]
        ctx.set_materialize_grads(False)
        dfeat4, demb4 = K.concat_tile_bwd(full(gm, gh), cf, ce)
        return dfeat4[:R], dfeat4[R:], demb4[:R], demb4[R:]

    @staticmethod
    def backward(ctx, ggfm, ggfh, ggem, ggeh):
        if ggfm is not None or ggem is not None:
            raise NotImplementedError('the main rows of a stacked first-order chain are not differentiated again')
        if ggfh is None and ggeh is None:
            return None, None, None, None
        return None, A.ConcatTileFn.apply(ggfh, ggeh), None, None


class SDeconvFn(Function):
    """act(conv^T(x, w) + b) on a stacked pair in one launch: tf conv2d_transpose as a forward op (the generator's upsampling layers).
    Backward per part: autograd.ConvBwdDataFn's, on that part's batch."""

    @staticmethod
    def forward(ctx, xm, xh, w, b, geom4, act, alpha):
        R = xm.shape[0]
        x4 = full(xm, xh)
        d, ws = geom4
        ctx.set_materialize_grads(False)
        out4 = K.conv_bwd_data(x4, w, b, d, ws, act, alpha)
        ctx.save_for_backward(x4, w, out4 if act != K.ACT_NONE else None)
        ctx.R, ctx.act, ctx.alpha, ctx.has_bias, ctx.bias_ref = R, act, alpha, b is not None, b
        ctx.geom_b4 = K.bwd_geom(geom4)
        return out4[:R], out4[R:]

    @staticmethod
    def backward(ctx, ggm, ggh):
        if ggm is None and ggh is None:
            return (None,) * 7
        x4, w, out4 = ctx.saved_tensors
        R, B4 = ctx.R, x4.shape[0]
        need_xm, need_xh, need_w, need_b = ctx.needs_input_grad[:4]
        gxm = gxh = gw = gb = None
        for g, sl, nx in ((ggm, slice(0, R), need_xm), (ggh, slice(R, B4), need_xh)):
            if g is None:
                continue
            geom = K.rebatch(ctx.geom_b4, sl.stop - sl.start)
            gx, gw1, gb1 = A.bwd_data_backward(x4[sl], w, out4[sl] if out4 is not None else None, g, geom, ctx.act, ctx.alpha, ctx.has_bias,
                                               ctx.bias_ref, (nx, need_w, need_b))
            if sl.start == 0:
                gxm = gx
            else:
                gxh = gx
            gw = gw1 if gw is None else (gw if gw1 is None else gw + gw1)
            gb = gb1 if gb is None else (gb if gb1 is None else gb + gb1)
        return gxm, gxh, gw, gb, None, None, None


_MOVING = [0]      # how many leading groups of a stacked batch norm move the moving averages (0 = all): see moving_groups()


class moving_groups(object):
    """with moving_groups(1): the stacked batch norms inside let only their first group move the moving averages (the paired generator:
    only the generator step's evaluation runs under UPDATE_OPS).  Default 0: every group does, in stacking order."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.prev, _MOVING[0] = _MOVING[0], self.n

    def __exit__(self, *a):
        _MOVING[0] = self.prev


class SBatchNormFn(Function):
    """Training-mode batch norm + activation of a stacked pair that consists of `groups` evaluations of the reference graph of equal size
    (main = the leading ones): each evaluation is normalised with ITS OWN batch statistics — one grouped launch chain for all
    (t2i_bn_train_fwd_grouped).  moving_groups = k > 0: only the first k evaluations move the moving averages (the paired generator:
    the generator-step evaluation runs under UPDATE_OPS, the critic step's does not: reference models/wgancls/model.py:98,102).
    First order; backward per part, on that part's groups."""

    @staticmethod
    def forward(ctx, xm, xh, gamma, beta, mm, mv, eps, decay, act, alpha, moving_updates, moving_grp, groups):
        R = xm.shape[0]
        x4 = full(xm, xh)
        assert x4.shape[0] % groups == 0 and R % (x4.shape[0] // groups) == 0, (tuple(x4.shape), R, groups)
        y4, mean, rstd = K.bn_train_fwd_grouped(x4, gamma, beta, eps, decay, groups, act, alpha, mm, mv, moving_updates, moving_grp)
        ctx.save_for_backward(x4, gamma, mean, rstd, y4 if act != K.ACT_NONE else None)
        ctx.R, ctx.act, ctx.alpha, ctx.gm = R, act, alpha, R // (x4.shape[0] // groups)
        ctx.groups = groups
        ctx.gamma_ref, ctx.beta_ref = gamma, beta
        ctx.set_materialize_grads(False)
        return y4[:R], y4[R:]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gym, gyh):
        if gym is None and gyh is None:
            return (None,) * 13
        x4, gamma, mean, rstd, y4 = ctx.saved_tensors
        R, gm = ctx.R, ctx.gm
        want_g, want_b = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        gsink = A._sink_of(ctx.gamma_ref) if want_g else None
        bsink = A._sink_of(ctx.beta_ref) if want_b else None
        sunk = gsink is not None and bsink is not None
        outs = [None, None]
        dgamma = dbeta = None
        for i, (g, sl, gs) in enumerate(((gym, slice(0, R), slice(0, gm)), (gyh, slice(R, x4.shape[0]), slice(gm, ctx.groups)))):
            if g is None:
                continue
            dx, dg, db = K.bn_bwd_grouped(_c(g), y4[sl] if y4 is not None else None, x4[sl], mean[gs], rstd[gs], gamma, gs.stop - gs.start, ctx.act,
                                          ctx.alpha, dgamma_out=gsink if sunk else None, dbeta_out=bsink if sunk else None)
            outs[i] = dx
            if not sunk:
                dgamma = dg if dgamma is None else dgamma + dg
                dbeta = db if dbeta is None else dbeta + db
        if sunk:
            A._notify(ctx.gamma_ref); A._notify(ctx.beta_ref)
            dgamma = dbeta = None
        return (outs[0], outs[1], dgamma if want_g else None, dbeta if want_b else None) + (None,) * 9


def batch_norm_ok(x, groups=2):
    """the grouped batch-norm kernels take this stacked tensor (C % 4 == 0, whole groups per part, 16-byte aligned groups)"""
    m, h = x.main, x.hat
    rows = m.shape[0] + h.shape[0]
    if rows % groups or m.shape[0] % (rows // groups) or not (m.is_contiguous() and h.is_contiguous()):
        return False
    per = (rows // groups) * (m[0].numel() if m.shape[0] else 0) * m.element_size()
    return m.shape[-1] % 4 == 0 and m.data_ptr() % 16 == 0 and per % 16 == 0


class STransposeFn(Function):
    """physical [B,C,H,W] -> physical [B,H,W,C] on a stacked pair (autograd.NchwToNhwcFn; ops.reshape_to_map)."""

    @staticmethod
    def forward(ctx, xm, xh):
        R = xm.shape[0]
        ctx.R = R
        ctx.set_materialize_grads(False)
        y4 = K.nchw_to_nhwc(full(xm, xh))
        return y4[:R], y4[R:]

    @staticmethod
    def backward(ctx, gm, gh):
        return (A.NhwcToNchwFn.apply(gm) if gm is not None else None), (A.NhwcToNchwFn.apply(gh) if gh is not None else None)


class SForkFn(Function):
    """A stacked tensor that feeds two consumers (the critic's trunk: the bottleneck branch and the residual join).  Forward: the same
    memory twice.  Backward: the two incoming stacked gradients summed by ONE launch on the whole buffer (autograd's own accumulation
    would add the main and the hat parts separately into unrelated allocations, which the layer below then has to concatenate)."""

    @staticmethod
    def forward(ctx, xm, xh):
        ctx.set_materialize_grads(False)
        return xm.view_as(xm), xh.view_as(xh), xm.view_as(xm), xh.view_as(xh)

    @staticmethod
    def backward(ctx, am, ah, bm, bh):
        if am is not None and ah is not None and bm is not None and bh is not None:
            return SAddActFn.apply(am, ah, bm, bh, K.ACT_NONE, 0.0)
        pick = lambda a, b: a if b is None else (b if a is None else a + b)
        return pick(am, bm), pick(ah, bh)


def fork(x):
    am, ah, bm, bh = SForkFn.apply(x.main, x.hat)
    return Stacked(am, ah), Stacked(bm, bh)


# ---- what utils/ops.py calls when it is handed a Stacked ----------------------------------------------------------------------
def conv2d(xp, w, b, geom4, act, alpha, want_stats=False):
    ym, yh = SConv2dFn.apply(xp.main, xp.hat, w, b, geom4, act, alpha, want_stats)
    return Stacked(ym, yh)


def conv2d_transpose(xp, w, b, geom4, act, alpha):
    ym, yh = SDeconvFn.apply(xp.main, xp.hat, w, b, geom4, act, alpha)
    return Stacked(ym, yh)


def batch_norm(xp, gamma, beta, mm, mv, eps, decay, act, alpha, moving_updates=1, groups=2):
    ym, yh = SBatchNormFn.apply(xp.main, xp.hat, gamma, beta, mm, mv, eps, decay, act, alpha, moving_updates, _MOVING[0], groups)
    return Stacked(ym, yh)


def nchw_to_nhwc(x):
    ym, yh = STransposeFn.apply(x.main, x.hat)
    return Stacked(ym, yh)


def add_act(a, b, act, alpha):
    ym, yh = SAddActFn.apply(a.main, a.hat, b.main, b.hat, act, alpha)
    return Stacked(ym, yh)


def concat_tile(feat, emb):
    ym, yh = SConcatTileFn.apply(feat.main, feat.hat, emb.main, emb.hat)
    return Stacked(ym, yh)
