"""Oracle (a): NumPy float64 restatement of every operator on the wgancls/gancls hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the
checker.  The product path (``text-to-image_amd``) never imports this package.

PARITY UNPINNED: the arithmetic of the reference lives in TensorFlow 1.4 (``README.md:46`` of the
reference; not vendored, not installable here) and the reference has no tests / golden vectors
(SURVEY.md §4, §8c).  This file restates the *documented* TF-1.4 semantics that the reference's
operator wrappers select, citing the wrapper that selects them:

  conv2d            reference utils/ops.py:58-63   -> tf.contrib.layers.conv2d (HWIO weights, SAME/VALID)
  conv2d_transpose  reference utils/ops.py:66-71   -> tf.contrib.layers.conv2d_transpose ([kh,kw,Cout,Cin])
  fc                reference utils/ops.py:84-87   -> tf.layers.dense
  batch_norm        reference utils/ops.py:7-29    -> tf.contrib.layers.batch_norm(fused=True, scale=True)
  lrelu             reference utils/ops.py:90-91 / models/wgancls/model.py:110,131
  Adam / SGD        reference models/wgancls/model.py:94-106 -> tf.train.AdamOptimizer (TF epsilon placement)
  gradient penalty  reference models/wgancls/model.py:62-70

Everything is written as direct tap loops over explicit index arithmetic (no library convolution),
in float64, NHWC, so that it is obviously-correct rather than fast.  ``conv2d_scalar`` is a fully
scalar 7-deep loop used to pin the tap-loop version on the tiniest cases.
"""
import math

import numpy as np

F64 = np.float64


# ----------------------------------------------------------------------------------------------
# padding arithmetic (TF "SAME"/"VALID"; reference utils/ops.py:58-71 pass the string through)
# ----------------------------------------------------------------------------------------------
def same_pad(in_size, k, s):
    """TF SAME: out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0); extra goes bottom/right."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def out_geometry(H, W, KH, KW, SH, SW, padding):
    """-> (Ho, Wo, pad_top, pad_left) for a case-insensitive TF padding string."""
    p = padding.upper()
    if p == 'SAME':
        Ho, pt, _ = same_pad(H, KH, SH)
        Wo, pl, _ = same_pad(W, KW, SW)
        return Ho, Wo, pt, pl
    if p == 'VALID':
        return (H - KH) // SH + 1, (W - KW) // SW + 1, 0, 0
    raise ValueError('Invalid padding %s' % padding)


# ----------------------------------------------------------------------------------------------
# convolution family
# ----------------------------------------------------------------------------------------------
def conv2d_scalar(x, w, b, stride, padding):
    """Fully scalar cross-correlation, used only to pin conv2d() on tiny inputs."""
    x = np.asarray(x, F64); w = np.asarray(w, F64)
    B, H, W, Ci = x.shape
    KH, KW, _, Co = w.shape
    SH, SW = stride
    Ho, Wo, pt, pl = out_geometry(H, W, KH, KW, SH, SW, padding)
    y = np.zeros((B, Ho, Wo, Co), F64)
    for n in range(B):
        for oh in range(Ho):
            for ow in range(Wo):
                for co in range(Co):
                    acc = 0.0 if b is None else float(b[co])
                    for kh in range(KH):
                        ih = oh * SH - pt + kh
                        if ih < 0 or ih >= H:
                            continue
                        for kw in range(KW):
                            iw = ow * SW - pl + kw
                            if iw < 0 or iw >= W:
                                continue
                            for ci in range(Ci):
                                acc += x[n, ih, iw, ci] * w[kh, kw, ci, co]
                    y[n, oh, ow, co] = acc
    return y


def _taps(H, W, KH, KW, SH, SW, Ho, Wo, pt, pl):
    """Yield (kh, kw, oh0, oh1, ow0, ow1): for tap (kh,kw), the output range whose input pixel
    ih = oh*SH - pt + kh, iw = ow*SW - pl + kw lies inside the image."""
    for kh in range(KH):
        oh0 = max(0, -(-(pt - kh) // SH))
        oh1 = min(Ho, (H - 1 + pt - kh) // SH + 1)
        if oh1 <= oh0:
            continue
        for kw in range(KW):
            ow0 = max(0, -(-(pl - kw) // SW))
            ow1 = min(Wo, (W - 1 + pl - kw) // SW + 1)
            if ow1 <= ow0:
                continue
            yield kh, kw, oh0, oh1, ow0, ow1


def conv2d(x, w, b=None, stride=(2, 2), padding='SAME'):
    """y[n,oh,ow,co] = b[co] + sum_{kh,kw,ci} x[n, oh*s-pt+kh, ow*s-pl+kw, ci] * w[kh,kw,ci,co]."""
    x = np.asarray(x, F64); w = np.asarray(w, F64)
    B, H, W, Ci = x.shape
    KH, KW, Ci2, Co = w.shape
    assert Ci == Ci2
    SH, SW = stride
    Ho, Wo, pt, pl = out_geometry(H, W, KH, KW, SH, SW, padding)
    y = np.zeros((B, Ho, Wo, Co), F64)
    for kh, kw, oh0, oh1, ow0, ow1 in _taps(H, W, KH, KW, SH, SW, Ho, Wo, pt, pl):
        ih0 = oh0 * SH - pt + kh
        iw0 = ow0 * SW - pl + kw
        xs = x[:, ih0:ih0 + (oh1 - oh0 - 1) * SH + 1:SH, iw0:iw0 + (ow1 - ow0 - 1) * SW + 1:SW, :]
        y[:, oh0:oh1, ow0:ow1, :] += xs @ w[kh, kw]
    if b is not None:
        y += np.asarray(b, F64)
    return y


def conv2d_bwd_data(dy, w, x_shape, stride=(2, 2), padding='SAME'):
    """Adjoint of conv2d wrt x: dx[n,ih,iw,ci] = sum dy[n,oh,ow,co] * w[kh,kw,ci,co] over all
    (oh,kh),(ow,kw) with oh*s-pt+kh == ih, ow*s-pl+kw == iw."""
    dy = np.asarray(dy, F64); w = np.asarray(w, F64)
    B, H, W, Ci = x_shape
    KH, KW, _, Co = w.shape
    SH, SW = stride
    Ho, Wo, pt, pl = out_geometry(H, W, KH, KW, SH, SW, padding)
    assert dy.shape == (B, Ho, Wo, Co), (dy.shape, (B, Ho, Wo, Co))
    dx = np.zeros((B, H, W, Ci), F64)
    for kh, kw, oh0, oh1, ow0, ow1 in _taps(H, W, KH, KW, SH, SW, Ho, Wo, pt, pl):
        ih0 = oh0 * SH - pt + kh
        iw0 = ow0 * SW - pl + kw
        dx[:, ih0:ih0 + (oh1 - oh0 - 1) * SH + 1:SH, iw0:iw0 + (ow1 - ow0 - 1) * SW + 1:SW, :] += \
            dy[:, oh0:oh1, ow0:ow1, :] @ w[kh, kw].T
    return dx


def conv2d_bwd_filter(x, dy, w_shape, stride=(2, 2), padding='SAME'):
    """Adjoint of conv2d wrt w: dw[kh,kw,ci,co] = sum_{n,oh,ow} x[n,ih,iw,ci] * dy[n,oh,ow,co]."""
    x = np.asarray(x, F64); dy = np.asarray(dy, F64)
    B, H, W, Ci = x.shape
    KH, KW, _, Co = w_shape
    SH, SW = stride
    Ho, Wo, pt, pl = out_geometry(H, W, KH, KW, SH, SW, padding)
    assert dy.shape == (B, Ho, Wo, Co)
    dw = np.zeros(w_shape, F64)
    for kh, kw, oh0, oh1, ow0, ow1 in _taps(H, W, KH, KW, SH, SW, Ho, Wo, pt, pl):
        ih0 = oh0 * SH - pt + kh
        iw0 = ow0 * SW - pl + kw
        xs = x[:, ih0:ih0 + (oh1 - oh0 - 1) * SH + 1:SH, iw0:iw0 + (ow1 - ow0 - 1) * SW + 1:SW, :]
        dw[kh, kw] = np.einsum('nhwi,nhwo->io', xs, dy[:, oh0:oh1, ow0:ow1, :])
    return dw


def conv2d_transpose(x, w, b=None, stride=(2, 2), padding='SAME'):
    """TF conv2d_transpose = gradient of conv2d wrt its input.  x [B,H,W,Cin]; w [kh,kw,Cout,Cin]
    (i.e. the HWIO filter of the conv that maps the [B,H*s,W*s,Cout] *output* back to x's shape).
    SAME: output spatial = in * stride; VALID: (in-1)*s + k."""
    x = np.asarray(x, F64); w = np.asarray(w, F64)
    B, H, W, Cin = x.shape
    KH, KW, Cout, Cin2 = w.shape
    assert Cin == Cin2
    SH, SW = stride
    if padding.upper() == 'SAME':
        Hout, Wout = H * SH, W * SW
    else:
        Hout, Wout = (H - 1) * SH + KH, (W - 1) * SW + KW
    y = conv2d_bwd_data(x, w, (B, Hout, Wout, Cout), stride, padding)
    if b is not None:
        y += np.asarray(b, F64)
    return y


def dense(x, kernel, bias=None):
    y = np.asarray(x, F64) @ np.asarray(kernel, F64)
    if bias is not None:
        y = y + np.asarray(bias, F64)
    return y


# ----------------------------------------------------------------------------------------------
# activations
# ----------------------------------------------------------------------------------------------
def lrelu(x, alpha=0.2):
    x = np.asarray(x, F64)
    return np.where(x > 0, x, alpha * x)


def lrelu_bwd(dy, y, alpha=0.2):
    """lrelu is sign preserving, so the mask can be taken from the output."""
    return np.where(np.asarray(y) > 0, dy, alpha * np.asarray(dy, F64))


def relu(x):
    return np.maximum(np.asarray(x, F64), 0.0)


def relu_bwd(dy, y):
    return np.where(np.asarray(y) > 0, dy, 0.0)


def tanh_bwd(dy, y):
    return np.asarray(dy, F64) * (1.0 - np.asarray(y, F64) ** 2)


# ----------------------------------------------------------------------------------------------
# batch norm (training mode, fused): rank-4 NHWC (per channel over N,H,W) or rank-2 (per feature)
# ----------------------------------------------------------------------------------------------
def batch_norm_train(x, gamma, beta, eps=1e-5):
    """-> (y, mean, var_biased).  Normalises with the *biased* batch variance."""
    x = np.asarray(x, F64)
    C = x.shape[-1]
    xf = x.reshape(-1, C)
    mean = xf.mean(0)
    var = ((xf - mean) ** 2).mean(0)
    y = (xf - mean) / np.sqrt(var + eps) * np.asarray(gamma, F64) + np.asarray(beta, F64)
    return y.reshape(x.shape), mean, var


def batch_norm_moving_update(moving_mean, moving_var, mean, var_biased, n, decay=0.9):
    """TF fused BN feeds the *unbiased* variance (x N/(N-1)) to the moving average;
    moving = decay*moving + (1-decay)*batch."""
    unbiased = var_biased * (n / max(n - 1, 1))
    return (decay * np.asarray(moving_mean, F64) + (1 - decay) * mean,
            decay * np.asarray(moving_var, F64) + (1 - decay) * unbiased)


def batch_norm_eval(x, gamma, beta, moving_mean, moving_var, eps=1e-5):
    x = np.asarray(x, F64)
    return (x - moving_mean) / np.sqrt(np.asarray(moving_var, F64) + eps) * gamma + beta


def batch_norm_bwd(dy, x, gamma, mean, var, eps=1e-5):
    """-> (dx, dgamma, dbeta) for training-mode BN."""
    dy = np.asarray(dy, F64); x = np.asarray(x, F64)
    C = x.shape[-1]
    dyf = dy.reshape(-1, C); xf = x.reshape(-1, C)
    n = xf.shape[0]
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (xf - mean) * rstd
    dbeta = dyf.sum(0)
    dgamma = (dyf * xhat).sum(0)
    dx = (np.asarray(gamma, F64) * rstd) * (dyf - dbeta / n - xhat * dgamma / n)
    return dx.reshape(x.shape), dgamma, dbeta


# ----------------------------------------------------------------------------------------------
# WGAN-GP pieces (reference models/wgancls/model.py:53, 62-70, 124-127)
# ----------------------------------------------------------------------------------------------
def interpolate(eps, g, x):
    return eps * np.asarray(g, F64) + (1.0 - eps) * np.asarray(x, F64)


def gp_from_grad(g):
    """slopes = sqrt(sum over non-batch axes g^2); penalty = mean(max(0, slopes-1)^2)."""
    g = np.asarray(g, F64)
    slopes = np.sqrt((g.reshape(g.shape[0], -1) ** 2).sum(1))
    return np.mean(np.maximum(0.0, slopes - 1.0) ** 2), slopes


def gp_from_grad_bwd(g, slopes, upstream=1.0):
    """d penalty / d g = upstream * 2*max(0,s-1)/(B*s) * g   (0 where s <= 1)."""
    g = np.asarray(g, F64)
    B = g.shape[0]
    coef = np.where(slopes > 1.0, 2.0 * (slopes - 1.0) / (B * np.maximum(slopes, 1e-300)), 0.0)
    return upstream * coef.reshape((B,) + (1,) * (g.ndim - 1)) * g


def kl_std_normal(mean, log_sigma):
    mean = np.asarray(mean, F64); ls = np.asarray(log_sigma, F64)
    return np.mean(-ls + 0.5 * (-1.0 + np.exp(2.0 * ls) + mean ** 2))


# ----------------------------------------------------------------------------------------------
# optimizers
# ----------------------------------------------------------------------------------------------
def adam_tf(w, g, m, v, t, lr, beta1, beta2, eps=1e-8):
    """tf.train.AdamOptimizer step t (1-based): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g^2; w -= lr_t*m/(sqrt(v)+eps)   (eps outside the correction)."""
    w = np.asarray(w, F64); g = np.asarray(g, F64)
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * np.asarray(m, F64) + (1.0 - beta1) * g
    v = beta2 * np.asarray(v, F64) + (1.0 - beta2) * g * g
    return w - lr_t * m / (np.sqrt(v) + eps), m, v


def truncated_normal(rng, shape, std=1.0):
    """N(0,1) resampled until |x| <= 2, then scaled (tf.truncated_normal)."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return out * std


def he_std(fan_in):
    """variance_scaling_initializer(factor=2.0, mode='FAN_IN', uniform=False): truncated normal with
    stddev = sqrt(1.3 * 2 / fan_in) (reference utils/ops.py:60,68,86)."""
    return math.sqrt(1.3 * 2.0 / fan_in)
