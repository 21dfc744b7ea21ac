"""The reference's operator surface (reference utils/ops.py:1-148), eager and MI355X-native.

Same names, parameters, defaults and error behaviour as the reference wrappers; below them sit libt2i_hip.so kernels
instead of tf.contrib.layers.  Differences that follow from "eager torch instead of a TF-1 graph":
  * parameters are created/looked up in a TF-style variable scope (``scope.py``) with TF's automatic names
    (``Conv``, ``Conv_1``, ``Conv2d_transpose``, ``dense``, ``BatchNorm``), so ``reuse=True`` and
    ``trainable_variables('d_net')`` mean what they mean in the reference (models/wgancls/model.py:59-60,134,167);
  * ``act`` is an ``Activation`` (``lrelu_act(0.2)``, ``relu``, ``tanh``) which the conv / BN epilogues fuse; any other
    callable is applied after the op, unfused;
  * data is physically NHWC always.  ``df=NCHW`` tensors are *logical* NCHW views (a permute of NHWC storage), so
    ``to_nchw`` / ``to_nhwc`` cost nothing and the result is layout-independent (SURVEY.md §7 last bullet).
north_star's ``deconv2d / linear / bn`` do not exist in the reference; they are exported as aliases.
"""
import torch

from .. import autograd as A
from .. import kernels as K
from .. import scope as S
from .. import stacked as ST

NHWC = 'NHWC'
NCHW = 'NCHW'


class Activation(object):
    """A fusable activation (what the reference passes as a python callable: tf.nn.relu, a leaky_relu lambda, tf.nn.tanh)."""

    def __init__(self, kind, alpha=0.0):
        self.kind, self.alpha = kind, float(alpha)

    def __call__(self, x):
        shape = x.shape
        return A.ActFn.apply(x.contiguous(), self.kind, self.alpha).view(shape)

    def __repr__(self):
        return 'Activation(%s, %g)' % ({K.ACT_LRELU: 'lrelu', K.ACT_RELU: 'relu', K.ACT_TANH: 'tanh'}[self.kind], self.alpha)


relu = Activation(K.ACT_RELU)
tanh = Activation(K.ACT_TANH)


def lrelu_act(alpha=0.2):
    """reference utils/ops.py:90-91"""
    return Activation(K.ACT_LRELU, alpha)


def _split_act(act):
    """-> (fused kind, alpha, post-callable)"""
    if act is None:
        return K.ACT_NONE, 0.0, None
    if isinstance(act, Activation):
        return act.kind, act.alpha, None
    if callable(act):
        return K.ACT_NONE, 0.0, act
    raise TypeError('act must be None, an ops.Activation or a callable')


def _check_df(df):
    if df not in (NHWC, NCHW):
        raise ValueError('Invalid data format %s' % df)


def _phys(x, df):
    """logical tensor -> physically-NHWC 4-D tensor"""
    _check_df(df)
    if x.dim() != 4:
        raise ValueError('expected a rank-4 tensor, got shape %s' % (tuple(x.shape),))
    return x.permute(0, 2, 3, 1) if df == NCHW else x


def _logical(y, df):
    return y.permute(0, 3, 1, 2) if df == NCHW else y


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def conv2d(x, f, ks=(4, 4), s=(2, 2), padding='SAME', act=None, init=None, name=None, df=NHWC, stats=False):
    """reference utils/ops.py:58-63.  Variables: <scope>/Conv[_k]/{weights [kh,kw,Cin,f] (He), biases [f] (zeros)}.
    stats=True (not in the reference; a hint, results are unchanged): a batch_norm consumes this output next, so the GEMM
    epilogue also emits the per-tile column sums the normalisation needs and batch_norm skips its own pass over the tensor."""
    st = S.default_store()
    xp = _phys(x, df)
    B, H, W, Cin = xp.shape
    (kh, kw), (sh, sw) = _pair(ks), _pair(s)
    kind, alpha, post = _split_act(act)
    with st.variable_scope(name or st.unique_op_name('Conv'), reuse=st.reuse()):
        w = st.get_variable('weights', (kh, kw, Cin, f), init or S.he_init(kh * kw * Cin))
        b = st.get_variable('biases', (f,), S.constant_init(0.0))
    geom = K.conv_desc(B, H, W, Cin, f, kh, kw, sh, sw, padding)
    if isinstance(xp, ST.Stacked):        # a stacked pass (stacked.py): one launch for both parts
        y = ST.conv2d(xp, w, b, geom, kind, alpha, bool(stats) and kind == K.ACT_NONE and post is None)
    else:
        y = A.Conv2dFn.apply(xp, w, b, geom, kind, alpha, bool(stats) and kind == K.ACT_NONE and post is None)
    y = _logical(y, df)
    return post(y) if post else y


def conv2d_transpose(x, f, ks=(4, 4), s=(2, 2), padding='SAME', act=None, init=None, name=None, df=NHWC):
    """reference utils/ops.py:66-71.  Variables: <scope>/Conv2d_transpose[_k]/{weights [kh,kw,f,Cin], biases [f]}.
    He fan_in follows TF's variance_scaling on that layout: kh*kw*shape[-2] = kh*kw*f."""
    st = S.default_store()
    xp = _phys(x, df)
    B, H, W, Cin = xp.shape
    (kh, kw), (sh, sw) = _pair(ks), _pair(s)
    kind, alpha, post = _split_act(act)
    with st.variable_scope(name or st.unique_op_name('Conv2d_transpose'), reuse=st.reuse()):
        w = st.get_variable('weights', (kh, kw, f, Cin), init or S.he_init(kh * kw * f))
        b = st.get_variable('biases', (f,), S.constant_init(0.0))
    geom = K.deconv_desc(B, H, W, Cin, f, kh, kw, sh, sw, padding)
    if isinstance(xp, ST.Stacked):
        y = ST.conv2d_transpose(xp, w, b, geom, kind, alpha)
    else:
        y = A.ConvBwdDataFn.apply(xp, w, b, geom, kind, alpha)
    y = _logical(y, df)
    return post(y) if post else y


def fc(x, units, act=None, init=None, bias=True, name=None):
    """reference utils/ops.py:84-87 (tf.layers.dense).  Variables: <scope>/dense[_k]/{kernel [in,units], bias [units]}.
    Runs as a 1x1 convolution on a [B,1,1,in] view (the same MFMA GEMM kernel)."""
    st = S.default_store()
    if x.dim() != 2:
        raise ValueError('fc expects a rank-2 tensor, got shape %s' % (tuple(x.shape),))
    B, I = x.shape
    kind, alpha, post = _split_act(act)
    with st.variable_scope(name or st.unique_op_name('dense'), reuse=st.reuse()):
        w = st.get_variable('kernel', (I, units), init or S.he_init(I))
        b = st.get_variable('bias', (units,), S.constant_init(0.0)) if bias else None
    geom = K.conv_desc(B, 1, 1, I, units, 1, 1, 1, 1, 'VALID')
    if isinstance(x, ST.Stacked):
        Bm, Bh = x.main.shape[0], x.hat.shape[0]
        y = ST.conv2d(ST.Stacked(x.main.reshape(Bm, 1, 1, I), x.hat.reshape(Bh, 1, 1, I)), w.view(1, 1, I, units), b, geom, kind, alpha)
        y = ST.Stacked(y.main.view(Bm, units), y.hat.view(Bh, units))
    else:
        y = A.Conv2dFn.apply(x.reshape(B, 1, 1, I), w.view(1, 1, I, units), b, geom, kind, alpha).view(B, units)
    return post(y) if post else y


# TF collects the moving-average assignments of batch_norm in GraphKeys.UPDATE_OPS and runs them only under ops that
# depend on them (reference models/wgancls/model.py:98,102: G_optim yes, D_optim no).  Eager equivalent: the moving
# statistics are updated by the BN kernel itself iff the caller is inside `update_ops()`.
_UPDATE_OPS = [False]          # False, or how many times the moving averages take each batch norm's statistics (True == 1)


class update_ops(object):
    """times = 2: one evaluation stands for two identical evaluations of the reference graph, each of which would run the update op
    (models/gancls/trainer.py: the generator in the D run and in the G run of one iteration)."""

    def __init__(self, times=1):
        self.times = int(times)

    def __enter__(self):
        self.prev = _UPDATE_OPS[0]
        _UPDATE_OPS[0] = self.times

    def __exit__(self, *a):
        _UPDATE_OPS[0] = self.prev


def batch_norm(x, train, init=None, act=None, name=None, eps=1e-5, decay=0.9, df=NHWC, groups=1):
    """reference utils/ops.py:7-29 (tf.contrib.layers.batch_norm, fused, scale=True).  Rank-4 (per channel) or rank-2
    (per feature).  Variables: <scope>/BatchNorm[_k]/{beta, gamma, moving_mean, moving_variance}.
    groups > 1 (not in the reference; training mode): x is a batched pass whose `groups` equal slices along the batch axis are separate
    passes of the reference graph — each slice is normalised with its own batch statistics (autograd.BatchNormTrainGroupedFn)."""
    st = S.default_store()
    _check_df(df)
    if x.dim() == 4:
        xp = _phys(x, df)
    elif x.dim() == 2:
        xp = x
    else:
        raise ValueError('batch_norm expects rank 2 or 4, got shape %s' % (tuple(x.shape),))
    C = xp.shape[-1]
    kind, alpha, post = _split_act(act)
    init = init or {}
    with st.variable_scope(name or st.unique_op_name('BatchNorm'), reuse=st.reuse()):
        beta = st.get_variable('beta', (C,), init.get('beta', S.constant_init(0.0)))
        gamma = st.get_variable('gamma', (C,), init.get('gamma', S.constant_init(1.0)))
        mm = st.get_variable('moving_mean', (C,), S.constant_init(0.0), trainable=False)
        mv = st.get_variable('moving_variance', (C,), S.constant_init(1.0), trainable=False)
    if train and isinstance(xp, ST.Stacked):
        # two evaluations of the reference graph stacked along the batch axis (the generator pair): per-part statistics; only the leading
        # part — the evaluation that runs under UPDATE_OPS — moves the moving averages
        total = int(groups) if groups > 1 else 2          # groups = 1: each part is one evaluation (the paired generator)
        upd = _UPDATE_OPS[0]
        if not ST.batch_norm_ok(xp, total):
            raise NotImplementedError('stacked batch norm needs C % 4 == 0, whole evaluations per part and 16-byte aligned evaluations')
        y = ST.batch_norm(xp, gamma, beta, mm if upd else None, mv if upd else None, eps, decay, kind, alpha, max(int(upd), 1), total)
    elif train and groups > 1:
        upd = _UPDATE_OPS[0]
        if int(upd) > 1:
            raise NotImplementedError('update_ops(times > 1) with a stacked batch')
        y = A.BatchNormTrainGroupedFn.apply(xp, gamma, beta, mm if upd else None, mv if upd else None, eps, decay, kind, alpha, int(groups))
    elif train:
        upd = _UPDATE_OPS[0]
        y, _, _ = A.BatchNormTrainFn.apply(xp, gamma, beta, mm if upd else None, mv if upd else None, eps, decay, kind, alpha, max(int(upd), 1))
    else:
        # inference: y = act(x*scale + shift) with the moving statistics; [C]-sized host-side vector math
        with torch.no_grad():
            scale = gamma / torch.sqrt(mv + eps)
            shift = beta - mm * scale
        y = K.bn_apply(xp.contiguous(), scale.contiguous(), shift.contiguous(), kind, alpha)
    if x.dim() == 4:
        y = _logical(y, df)
    return post(y) if post else y


def layer_norm(x, act=None, scope=None, df=NHWC):
    """reference utils/ops.py:74-81 (tf.contrib.layers.layer_norm, begin_params_axis = channel axis).  Rank-4 NHWC or
    rank-2.  Variables: <scope>/LayerNorm[_k]/{beta [C] zeros, gamma [C] ones}."""
    st = S.default_store()
    _check_df(df)
    if x.dim() == 4:
        xp = _phys(x, df)
    elif x.dim() == 2:
        xp = x
    else:
        raise ValueError('layer_norm expects rank 2 or 4, got shape %s' % (tuple(x.shape),))
    C = xp.shape[-1]
    kind, alpha, post = _split_act(act)
    with st.variable_scope(scope or st.unique_op_name('LayerNorm'), reuse=st.reuse()):
        beta = st.get_variable('beta', (C,), S.constant_init(0.0))
        gamma = st.get_variable('gamma', (C,), S.constant_init(1.0))
    y = A.LayerNormFn.apply(xp, gamma, beta, 1e-12, kind, alpha)
    if x.dim() == 4:
        y = _logical(y, df)
    return post(y) if post else y


def pool(x, s=2, p_type='AVG', df=NHWC):
    """reference utils/ops.py:100-101 (tf.nn.pool, window = stride = s, SAME).  The reference only calls pool(x, 2) with
    the default average type on power-of-two maps: that case is built."""
    if s != 2 or p_type != 'AVG':
        raise NotImplementedError('pool: only the 2x2 average pool the reference uses is built (got s=%r, %r)' % (s, p_type))
    _check_df(df)
    return _logical(A.Pool2Fn.apply(_phys(x, df), 0.25), df)


def upscale(x, s=2):
    """reference utils/ops.py:109-111: nearest-neighbour resize to (h*s, w*s), NHWC."""
    if s != 2:
        raise NotImplementedError('upscale: only the factor 2 the reference uses is built (got %r)' % (s,))
    return A.Upscale2Fn.apply(x, 1.0)


def lerp(a, b, t):
    """(1 - t)*a + t*b with a host scalar t: the fade-in of a new resolution (tf.multiply / tf.add on the `alpha_tra`
    variable, reference models/pggan/pggan.py:267,314)."""
    if isinstance(t, torch.Tensor):          # weight in device memory: the step can be captured into a hipGraph
        return A.LerpDevFn.apply(a, b, t, 0)
    return A.AxpbyFn.apply(a, 1.0 - float(t), b, float(t))


def to_nchw(x):
    """reference utils/ops.py:129-130.  Logical transpose only: storage stays NHWC."""
    return x.permute(0, 3, 1, 2)


def to_nhwc(x):
    """reference utils/ops.py:133-134"""
    return x.permute(0, 2, 3, 1)


def reshape_to_map(x, C, H, W, df=NHWC):
    """tf.reshape of a rank-2 activation to a feature map (reference models/wgancls/model.py:178-181).  With df=NCHW
    feature index = c*H*W + h*W + w, so the storage is re-tiled to NHWC by one transpose kernel and returned as a
    logical-NCHW view; with df=NHWC the reshape is free."""
    _check_df(df)
    B = x.shape[0]
    if isinstance(x, ST.Stacked):
        if df == NHWC:
            return x.reshape_parts(H, W, C)
        return ST.nchw_to_nhwc(x.reshape_parts(C, H, W)).permute(0, 3, 1, 2)
    if df == NHWC:
        return x.reshape(B, H, W, C)
    return A.NchwToNhwcFn.apply(x.reshape(B, C, H, W)).permute(0, 3, 1, 2)


def add(a, b, act=None, df=NHWC):
    """tf.add followed by an activation (the residual joins, reference models/wgancls/model.py:145-146,190-191)."""
    kind, alpha, post = _split_act(act)
    if isinstance(a, ST.Stacked):
        y = _logical(ST.add_act(_phys(a, df), _phys(b, df), kind, alpha), df)
    elif a.dim() == 4:
        y = _logical(A.AddActFn.apply(_phys(a, df), _phys(b, df), kind, alpha), df)
    else:
        y = A.AddActFn.apply(a, b, kind, alpha)
    return post(y) if post else y


def fork(x, df=NHWC):
    """-> (x, x) for a tensor with two consumers.  Plain tensors: the tensor itself twice (autograd sums the two gradients).  A stacked
    pass (stacked.py): two aliases whose incoming stacked gradients are summed by one launch on the whole buffer."""
    if not isinstance(x, ST.Stacked):
        return x, x
    a, b = ST.fork(_phys(x, df))
    return _logical(a, df), _logical(b, df)


def concat_tile(feat, emb, df=NHWC):
    """expand_dims x2 -> tile over the spatial map -> concat on channels (reference models/wgancls/model.py:153-155)."""
    if isinstance(feat, ST.Stacked):
        return _logical(ST.concat_tile(_phys(feat, df), emb), df)
    return _logical(A.ConcatTileFn.apply(_phys(feat, df), emb), df)


def get_conv_shape(tensor):
    return get_ints_from_shape(tensor)


def get_ints_from_shape(tensor):
    return [int(n) for n in tensor.shape]


def df_to_channel(df):
    """reference utils/ops.py:137-142"""
    if df == NHWC:
        return 'channels_last'
    if df == NCHW:
        return 'channels_first'
    raise RuntimeError('Invalid data format %s' % df)


# aliases named by BASELINE.json:north_star
deconv2d = conv2d_transpose
linear = fc
bn = batch_norm
