"""Oracle (b'): torch-CPU autograd restatement of the gancls iteration (the hot path's second variant).

TEST INFRASTRUCTURE ONLY (see oracle/np_ops.py).  PARITY UNPINNED against TensorFlow; pinned against
oracle (a) through the shared operator restatements in oracle/torch_step.py.

Follows reference models/gancls/model.py:36-192 (GanCls: NHWC, tf.layers.* with N(0,0.02) kernels, BatchNorm in the
discriminator with gamma ~ N(1,0.02)) and models/gancls/trainer.py:19-51,114-133 (sigmoid cross-entropy losses with
one-sided label smoothing 0.9, D_loss = match + a*mismatch + (1-a)*fake, Adam(2e-4, beta1=0.5) for both nets, BOTH
optimizers under tf.GraphKeys.UPDATE_OPS).  Variable names are the tf.layers auto-names (`conv2d_3/kernel`,
`conv2d_transpose/kernel`, `dense_1/bias`, `BatchNorm_4/gamma`).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .torch_step import AdamTF, _bn, _conv, _deconv_k4s2, is_trainable, trainable  # noqa: F401
from .torch_step import lrelu as _lrelu, relu as _relu, tape_section  # activations that honour an installed SectionTape (branch pinning)


class Cfg(object):
    """reference models/gancls/cfg/flowers.yml:9-33"""

    def __init__(self, z_dim=100, embed_dim=1024, compressed=128, gf=128, df=64, channels=3, batch=64, lr=2e-4, beta1=0.5,
                 alpha=0.5):
        self.z_dim, self.embed_dim, self.compressed, self.gf, self.df = z_dim, embed_dim, compressed, gf, df
        self.channels, self.batch, self.lr, self.beta1, self.alpha = channels, batch, lr, beta1, alpha


def _n(base, i):
    return base if i == 0 else '%s_%d' % (base, i)


def variable_shapes(cfg):
    """-> OrderedDict name -> (shape, kind); kind: 'wn' N(0,.02) kernel, 'wg' glorot-uniform kernel, 'b', 'gamma', 'beta', 'mm', 'mv'."""
    V = OrderedDict()
    d, g, ce, C = cfg.df, cfg.gf, cfg.compressed, cfg.channels

    def conv(sc, i, kh, kw, ci, co):
        V['%s/%s/kernel' % (sc, _n('conv2d', i))] = ((kh, kw, ci, co), 'wn'); V['%s/%s/bias' % (sc, _n('conv2d', i))] = ((co,), 'b')

    def deconv(sc, i, co, ci):
        V['%s/%s/kernel' % (sc, _n('conv2d_transpose', i))] = ((4, 4, co, ci), 'wn')
        V['%s/%s/bias' % (sc, _n('conv2d_transpose', i))] = ((co,), 'b')

    def dense(sc, i, ni, no, kind):
        V['%s/%s/kernel' % (sc, _n('dense', i))] = ((ni, no), kind); V['%s/%s/bias' % (sc, _n('dense', i))] = ((no,), 'b')

    def bn(sc, i, c):
        for k, kind in (('beta', 'beta'), ('gamma', 'gamma'), ('moving_mean', 'mm'), ('moving_variance', 'mv')):
            V['%s/%s/%s' % (sc, _n('BatchNorm', i), k)] = ((c,), kind)

    # generator first (model.py:47), reference model.py:111-192
    dense('g_net', 0, cfg.embed_dim, ce, 'wg'); dense('g_net', 1, cfg.z_dim + ce, 8 * g * 16, 'wn'); bn('g_net', 0, 8 * g * 16)
    conv('g_net', 0, 1, 1, 8 * g, 2 * g); bn('g_net', 1, 2 * g); conv('g_net', 1, 3, 3, 2 * g, 2 * g); bn('g_net', 2, 2 * g)
    conv('g_net', 2, 3, 3, 2 * g, 8 * g); bn('g_net', 3, 8 * g)
    deconv('g_net', 0, 4 * g, 8 * g); conv('g_net', 3, 3, 3, 4 * g, 4 * g); bn('g_net', 4, 4 * g)
    conv('g_net', 4, 1, 1, 4 * g, g); bn('g_net', 5, g); conv('g_net', 5, 3, 3, g, g); bn('g_net', 6, g)
    conv('g_net', 6, 3, 3, g, 4 * g); bn('g_net', 7, 4 * g)
    deconv('g_net', 1, 2 * g, 4 * g); conv('g_net', 7, 3, 3, 2 * g, 2 * g); bn('g_net', 8, 2 * g)
    deconv('g_net', 2, g, 2 * g); conv('g_net', 8, 3, 3, g, g); bn('g_net', 9, g)
    deconv('g_net', 3, C, g); conv('g_net', 9, 3, 3, C, C)
    # discriminator, reference model.py:54-109
    conv('d_net', 0, 4, 4, C, d); conv('d_net', 1, 4, 4, d, 2 * d); bn('d_net', 0, 2 * d)
    conv('d_net', 2, 4, 4, 2 * d, 4 * d); bn('d_net', 1, 4 * d); conv('d_net', 3, 4, 4, 4 * d, 8 * d); bn('d_net', 2, 8 * d)
    conv('d_net', 4, 1, 1, 8 * d, 2 * d); bn('d_net', 3, 2 * d); conv('d_net', 5, 3, 3, 2 * d, 2 * d); bn('d_net', 4, 2 * d)
    conv('d_net', 6, 3, 3, 2 * d, 8 * d); bn('d_net', 5, 8 * d)
    dense('d_net', 0, cfg.embed_dim, ce, 'wg')
    conv('d_net', 7, 1, 1, 8 * d + ce, 8 * d); bn('d_net', 6, 8 * d); conv('d_net', 8, 4, 4, 8 * d, 1)
    return V


def init_variables(cfg, seed=0, dtype=torch.float32):
    rng = np.random.default_rng(seed)
    P = OrderedDict()
    for name, (shape, kind) in variable_shapes(cfg).items():
        if kind == 'wn':
            a = rng.standard_normal(shape) * 0.02
        elif kind == 'wg':      # tf.layers default: glorot_uniform
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            a = rng.uniform(-lim, lim, shape)
        elif kind == 'gamma':
            a = 1.0 + rng.standard_normal(shape) * 0.02
        elif kind == 'mv':
            a = np.ones(shape)
        else:
            a = np.zeros(shape)
        P[name] = torch.tensor(a, dtype=dtype)
    return P


def generator(P, cfg, z, embed, train=True, stats=None):
    """-> img NHWC.  Activations are kept NCHW internally; the NHWC reshape of dense_1 (model.py:126) is honoured."""
    g = cfg.gf
    B = z.shape[0]
    e = embed @ P['g_net/dense/kernel'] + P['g_net/dense/bias']
    h = torch.cat([z, e], 1) @ P['g_net/dense_1/kernel'] + P['g_net/dense_1/bias']
    h = _bn(P, 'g_net/BatchNorm', h, train, stats)
    h0 = h.reshape(B, 4, 4, 8 * g).permute(0, 3, 1, 2)

    def cv(i, x, pad='SAME'):
        n = 'g_net/' + _n('conv2d', i)
        return _conv(x, P[n + '/kernel'], P[n + '/bias'], 1, pad)

    def dc(i, x):
        n = 'g_net/' + _n('conv2d_transpose', i)
        return _deconv_k4s2(x, P[n + '/kernel'], P[n + '/bias'])

    bn = lambda i, x: _bn(P, 'g_net/BatchNorm_%d' % i, x, train, stats)
    r = _relu(bn(1, cv(0, h0, 'VALID'))); r = _relu(bn(2, cv(1, r))); r = bn(3, cv(2, r))
    h1 = _relu(h0 + r)
    h2 = bn(4, cv(3, dc(0, h1)))
    r = _relu(bn(5, cv(4, h2, 'VALID'))); r = _relu(bn(6, cv(5, r))); r = bn(7, cv(6, r))
    h3 = _relu(h2 + r)
    h4 = _relu(bn(8, cv(7, dc(1, h3))))
    h5 = _relu(bn(9, cv(8, dc(2, h4))))
    return torch.tanh(cv(9, dc(3, h5))).permute(0, 2, 3, 1)


def discriminator(P, cfg, img_nhwc, embed, train=True, stats=None, tag=''):
    """-> logits [B,1,1,1].  `tag` distinguishes the three calls' batch statistics in `stats`."""
    x = img_nhwc.permute(0, 3, 1, 2)
    st = {} if stats is not None else None

    def cv(i, x, s, pad='SAME'):
        n = 'd_net/' + _n('conv2d', i)
        return _conv(x, P[n + '/kernel'], P[n + '/bias'], s, pad)

    bn = lambda i, x: _bn(P, 'd_net/' + _n('BatchNorm', i), x, train, st)
    h0 = _lrelu(cv(0, x, 2)); h1 = _lrelu(bn(0, cv(1, h0, 2))); h2 = _lrelu(bn(1, cv(2, h1, 2))); h3 = bn(2, cv(3, h2, 2))
    r = _lrelu(bn(3, cv(4, h3, 1, 'valid'))); r = _lrelu(bn(4, cv(5, r, 1))); r = bn(5, cv(6, r, 1))
    h4 = _lrelu(h3 + r)
    e = _lrelu(embed @ P['d_net/dense/kernel'] + P['d_net/dense/bias'])
    e = e[:, :, None, None].expand(-1, -1, 4, 4)
    h = _lrelu(bn(6, cv(7, torch.cat([h4, e], 1), 1, 'valid')))
    out = cv(8, h, 4, 'valid')
    if stats is not None:
        stats.append((tag, st))
    return out


def sigmoid_ce(logits, label):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(l,0) - l*y + log(1+exp(-|l|)), mean over the batch."""
    return torch.mean(torch.clamp(logits, min=0) - logits * label + torch.log1p(torch.exp(-logits.abs())))


def d_step(P, cfg, feed, term_scales=False):
    """term_scales: also return `scales`, per critic tensor max_i |coef_i| max|d term_i / d theta| over the three loss terms of
    D_loss = match + alpha mismatch + (1 - alpha) fake — the magnitude a gradient has BEFORE the terms cancel.  The logit bias is
    the case that needs it: its gradient is sum_i coef_i mean(sigmoid(l_i) - y_i), three numbers of order 0.1-0.4 that cancel to
    ~1e-5 on some batches, where fp32 (any fp32: torch-CPU float32 differs from float64 by 8e-3 there) can only promise 1e-4 of
    the terms, not of the difference."""
    names = trainable(P, 'd_net')
    Q = dict(P)
    for n in names:
        Q[n] = P[n].detach().requires_grad_(True)
    gstats, dstats = {}, []
    with torch.no_grad(), tape_section('G'):
        G = generator(P, cfg, feed['z'], feed['cond'], True, gstats)
    with tape_section('Dfake'):
        lf = discriminator(Q, cfg, G, feed['cond'], True, dstats, 'fake')
    with tape_section('Dmatch'):
        lm = discriminator(Q, cfg, feed['x'], feed['cond'], True, dstats, 'match')
    with tape_section('Dmis'):
        lw = discriminator(Q, cfg, feed['x_mismatch'], feed['cond'], True, dstats, 'mismatch')
    fake, match, mism = sigmoid_ce(lf, 0.0), sigmoid_ce(lm, 0.9), sigmoid_ce(lw, 0.0)
    D_loss = match + cfg.alpha * mism + (1.0 - cfg.alpha) * fake
    scales = None
    if term_scales:
        scales = OrderedDict((n, 0.0) for n in names)
        for coef, term in ((1.0, match), (cfg.alpha, mism), (1.0 - cfg.alpha, fake)):
            gs = torch.autograd.grad(term, [Q[n] for n in names], retain_graph=True, allow_unused=True)
            for n, g in zip(names, gs):
                if g is not None:
                    scales[n] = max(scales[n], abs(coef) * float(g.abs().max()))
    grads = torch.autograd.grad(D_loss, [Q[n] for n in names])
    f = lambda t: float(t.detach())
    return dict(D_loss=f(D_loss), D_real_match_loss=f(match), D_real_mismatch_loss=f(mism), D_synthetic_loss=f(fake),
                grads=OrderedDict((n, g.detach()) for n, g in zip(names, grads)), G=G.detach(), g_stats=gstats, d_stats=dstats,
                scales=scales)


def g_step(P, cfg, feed):
    names = trainable(P, 'g_net')
    Q = dict(P)
    for n in names:
        Q[n] = P[n].detach().requires_grad_(True)
    gstats, dstats = {}, []
    with tape_section('G'):
        G = generator(Q, cfg, feed['z'], feed['cond'], True, gstats)
    with tape_section('Dfake'):
        lf = discriminator(Q, cfg, G, feed['cond'], True, dstats, 'fake')
    G_loss = sigmoid_ce(lf, 1.0)
    grads = torch.autograd.grad(G_loss, [Q[n] for n in names])
    with torch.no_grad():   # the other two critic passes only exist in this run for their UPDATE_OPS (trainer.py:46-51)
        with tape_section('Dmatch'):
            discriminator(P, cfg, feed['x'], feed['cond'], True, dstats, 'match')
        with tape_section('Dmis'):
            discriminator(P, cfg, feed['x_mismatch'], feed['cond'], True, dstats, 'mismatch')
    return dict(G_loss=float(G_loss.detach()), grads=OrderedDict((n, g.detach()) for n, g in zip(names, grads)),
                G=G.detach(), g_stats=gstats, d_stats=dstats)


def apply_moving(P, gstats, dstats, decay=0.9):
    def upd(name, mean, var, n):
        P[name + '/moving_mean'] = decay * P[name + '/moving_mean'] + (1 - decay) * mean
        P[name + '/moving_variance'] = decay * P[name + '/moving_variance'] + (1 - decay) * var * (n / max(n - 1, 1))
    for name, (mean, var, n) in gstats.items():
        upd(name, mean, var, n)
    for _, st in dstats:       # fake, match, mismatch — the order the graph was built in (model.py:48-51)
        for name, (mean, var, n) in st.items():
            upd(name, mean, var, n)


class Trainer(object):
    """reference models/gancls/trainer.py:114-133: D update then G update, every iteration, both under UPDATE_OPS."""

    def __init__(self, cfg, P):
        self.cfg, self.P = cfg, P
        self.opt_d = AdamTF(trainable(P, 'd_net'), P, cfg.beta1, 0.999)
        self.opt_g = AdamTF(trainable(P, 'g_net'), P, cfg.beta1, 0.999)

    def iteration(self, feed):
        d = d_step(self.P, self.cfg, feed)
        self.opt_d.apply(self.P, d['grads'], self.cfg.lr)
        apply_moving(self.P, d['g_stats'], d['d_stats'])
        g = g_step(self.P, self.cfg, feed)
        self.opt_g.apply(self.P, g['grads'], self.cfg.lr)
        apply_moving(self.P, g['g_stats'], g['d_stats'])
        return {'d': d, 'g': g}


def synthetic_feed(cfg, seed=1, dtype=torch.float32):
    rng = np.random.default_rng(seed)
    B = cfg.batch
    t = lambda a: torch.tensor(a, dtype=dtype)
    return dict(x=t(rng.uniform(-1, 1, (B, 64, 64, cfg.channels))), x_mismatch=t(rng.uniform(-1, 1, (B, 64, 64, cfg.channels))),
                cond=t(rng.standard_normal((B, cfg.embed_dim))), z=t(rng.standard_normal((B, cfg.z_dim))))
