"""The three `tf.layers` calls the gancls model makes directly instead of going through utils/ops.py (reference
models/gancls/model.py:58-109,116-189): same keyword names, same variable names TF gives them (`conv2d[_k]/{kernel,bias}`,
`conv2d_transpose[_k]/{kernel,bias}`, `dense[_k]/{kernel,bias}`) and the same default initializer (glorot_uniform when
`kernel_initializer` is None), on the same libt2i_hip.so kernels as utils/ops.py.  NHWC only (tf.layers' default
`channels_last`, which is what gancls uses)."""
import math

import torch

from .. import autograd as A
from .. import kernels as K
from .. import scope as S
from .. import stacked as ST
from .ops import _split_act


def glorot_uniform_init(fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))

    def init(shape, gen):
        return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * lim
    return init


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def conv2d(inputs, filters, kernel_size, strides=(1, 1), padding='valid', activation=None, kernel_initializer=None,
           name=None):
    st = S.default_store()
    B, H, W, Cin = inputs.shape
    (kh, kw), (sh, sw) = _pair(kernel_size), _pair(strides)      # the reference passes float 4.0 here (model.py:55,106)
    kind, alpha, post = _split_act(activation)
    with st.variable_scope(name or st.unique_op_name('conv2d'), reuse=st.reuse()):
        w = st.get_variable('kernel', (kh, kw, Cin, filters),
                            kernel_initializer or glorot_uniform_init(kh * kw * Cin, kh * kw * filters))
        b = st.get_variable('bias', (filters,), S.constant_init(0.0))
    geom = K.conv_desc(B, H, W, Cin, filters, kh, kw, sh, sw, padding)
    y = ST.conv2d(inputs, w, b, geom, kind, alpha) if isinstance(inputs, ST.Stacked) else A.Conv2dFn.apply(inputs, w, b, geom, kind, alpha)
    return post(y) if post else y


def conv2d_transpose(inputs, filters, kernel_size, strides=(1, 1), padding='valid', activation=None,
                     kernel_initializer=None, name=None):
    st = S.default_store()
    B, H, W, Cin = inputs.shape
    (kh, kw), (sh, sw) = _pair(kernel_size), _pair(strides)
    kind, alpha, post = _split_act(activation)
    with st.variable_scope(name or st.unique_op_name('conv2d_transpose'), reuse=st.reuse()):
        w = st.get_variable('kernel', (kh, kw, filters, Cin),
                            kernel_initializer or glorot_uniform_init(kh * kw * Cin, kh * kw * filters))
        b = st.get_variable('bias', (filters,), S.constant_init(0.0))
    y = A.ConvBwdDataFn.apply(inputs, w, b, K.deconv_desc(B, H, W, Cin, filters, kh, kw, sh, sw, padding), kind, alpha)
    return post(y) if post else y


def dense(inputs, units, activation=None, kernel_initializer=None, name=None):
    st = S.default_store()
    B, I = inputs.shape
    kind, alpha, post = _split_act(activation)
    with st.variable_scope(name or st.unique_op_name('dense'), reuse=st.reuse()):
        w = st.get_variable('kernel', (I, units), kernel_initializer or glorot_uniform_init(I, units))
        b = st.get_variable('bias', (units,), S.constant_init(0.0))
    geom = K.conv_desc(B, 1, 1, I, units, 1, 1, 1, 1, 'VALID')
    if isinstance(inputs, ST.Stacked):        # a stacked pass (stacked.py)
        y = ST.conv2d(inputs.reshape_parts(1, 1, I), w.view(1, 1, I, units), b, geom, kind, alpha).reshape_parts(units)
    else:
        y = A.Conv2dFn.apply(inputs.reshape(B, 1, 1, I), w.view(1, 1, I, units), b, geom, kind, alpha).view(B, units)
    return post(y) if post else y
