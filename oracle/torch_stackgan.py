"""Oracle (b''): torch-CPU autograd restatement of the StackGAN Stage-I and Stage-II iterations.

TEST INFRASTRUCTURE ONLY (see oracle/np_ops.py).  PARITY UNPINNED against TensorFlow; pinned against
oracle (a) through the shared operator restatements in oracle/torch_step.py.

Follows reference models/stackgan/stageI/model.py:37-171 + stageI/trainer.py:19-59 and
models/stackgan/stageII/model.py:39-201 + stageII/trainer.py:20-63:
  * NHWC graphs built from utils/ops.py conv2d / conv2d_transpose / batch_norm (tf.contrib.layers auto-names `Conv`,
    `Conv_1`, `Conv2d_transpose`, `BatchNorm_3` ...) and tf.layers.dense (`dense`, `dense_1` ...);
  * conditioning augmentation (two lrelu dense heads, mean + exp(log_sigma) * eps with eps ~ truncated normal) and its
    KL term  mean(-log_sigma + .5 * (-1 + exp(2 log_sigma) + mean^2));
  * sigmoid cross-entropy losses, real label 0.9 (Stage-I) / 0.95 (Stage-II), D = match + a*mismatch + (1-a)*fake,
    G = CE(fake, 1) + kl_coeff * KL; Adam(lr * 0.5^(epoch // 100), beta1) for both nets, BOTH under UPDATE_OPS;
  * Stage-II: the Stage-I generator runs INSIDE the Stage-II graph in training mode (stageII/model.py:50: default
    is_training=True) with frozen weights — its batch-norm moving averages keep moving; the discriminator's residual
    join is `tf.add(net, net)` (stageII/model.py:117: doubles the branch, drops the trunk); k4 s1 SAME convs pad (1,2).
Variables are created by running the graph once in "create" mode, with TF's per-scope auto-naming restated in `Vars`.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import np_ops
from .torch_step import AdamTF, _bn, _conv, _deconv_k4s2, trainable  # noqa: F401
from .torch_step import lrelu as _lrelu, relu as _relu, tape_section  # activations that honour an installed SectionTape


class Cfg(object):
    """reference models/stackgan/stageI/cfg/flowers.yml and stageII/cfg/flowers.yml (MODEL / TRAIN blocks)."""

    def __init__(self, z_dim=100, embed_dim=1024, compressed=128, gf=128, df=64, channels=3, batch=64, lr=2e-4, beta1=0.5,
                 alpha=0.5, kl=2.0, out_size=64, real_label=0.9):
        self.z_dim, self.embed_dim, self.compressed, self.gf, self.df = z_dim, embed_dim, compressed, gf, df
        self.channels, self.batch, self.lr, self.beta1, self.alpha, self.kl = channels, batch, lr, beta1, alpha, kl
        self.out_size, self.real_label = out_size, real_label


# Un-cancelled scale of the bias gradients (test infrastructure, like d_step's term_scales): dL/db[c] = sum over samples and positions of the
# gradient g that reaches the layer's output.  In front of a batch norm that sum is zero in exact arithmetic, behind a zero-padded convolution
# it is a border effect — a small difference of large terms whose rounding error scales with sum |g|, not with |sum g|.  With a dict in
# _BIAS_L1[0] every biased layer records sum |g| per channel there (name of the bias variable -> tensor); the reduced-precision parity tests
# take it as the yardstick for exactly those tensors.
_BIAS_L1 = [None]


def _record_bias_l1(y, name):
    rec = _BIAS_L1[0]
    if rec is not None and y.requires_grad:
        dims = tuple(d for d in range(y.dim()) if d != 1)

        def hook(g, name=name, dims=dims):
            v = g.detach().abs().sum(dims)
            rec[name] = v if name not in rec else torch.maximum(rec[name], v)
        y.register_hook(hook)
    return y


class Vars(object):
    """Parameter access with TF-1 auto-naming: inside one variable_scope the k-th layer of a kind is `<Base>` for k = 0
    and `<Base>_k` after; with P=None the variables are created (initializers of the reference), else looked up."""

    def __init__(self, P=None, seed=0, dtype=torch.float64):
        self.create = P is None
        self.P = OrderedDict() if P is None else P
        self.rng = np.random.default_rng(seed)
        self.dtype = dtype
        self.scope, self.count = '', {}

    def enter(self, scope):
        self.scope, self.count = scope, {}
        return self

    def _name(self, base):
        k = self.count.get(base, 0)
        self.count[base] = k + 1
        return '%s/%s' % (self.scope, base if k == 0 else '%s_%d' % (base, k))

    def _get(self, name, shape, kind):
        if self.create and name not in self.P:
            r = self.rng
            if kind == 'n02':                                   # tf.random_normal_initializer(stddev=0.02)
                a = r.standard_normal(shape) * 0.02
            elif kind == 'he':                                  # variance_scaling(2.0, FAN_IN, uniform=False), utils/ops.py:60
                fan_in = int(np.prod(shape[:-1])) if len(shape) == 4 else shape[0]
                a = np_ops.truncated_normal(r, shape, np_ops.he_std(fan_in))
            elif kind == 'glorot':                              # tf.layers.dense default
                lim = np.sqrt(6.0 / (shape[0] + shape[1]))
                a = r.uniform(-lim, lim, shape)
            elif kind == 'gamma':                               # random_normal_initializer(1., 0.02)
                a = 1.0 + r.standard_normal(shape) * 0.02
            elif kind == 'ones':
                a = np.ones(shape)
            else:
                a = np.zeros(shape)
            self.P[name] = torch.tensor(a, dtype=self.dtype)
        return self.P[name]

    # ---- layers (activations NCHW inside the oracle; weights in TF layouts) ----
    def conv(self, x, f, k, s=1, pad='SAME', init='n02'):
        n = self._name('Conv')
        w = self._get(n + '/weights', (k, k, x.shape[1], f), init)
        b = self._get(n + '/biases', (f,), 'zeros')
        return _record_bias_l1(_conv(x, w, b, s, pad), n + '/biases')

    def deconv(self, x, f, init='n02'):
        n = self._name('Conv2d_transpose')
        w = self._get(n + '/weights', (4, 4, f, x.shape[1]), init)
        b = self._get(n + '/biases', (f,), 'zeros')
        return _record_bias_l1(_deconv_k4s2(x, w, b), n + '/biases')

    def dense(self, x, units, init):
        n = self._name('dense')
        return _record_bias_l1(x @ self._get(n + '/kernel', (x.shape[1], units), init) + self._get(n + '/bias', (units,), 'zeros'), n + '/bias')

    def bn(self, x, train, stats):
        n = self._name('BatchNorm')
        C = x.shape[1]
        self._get(n + '/beta', (C,), 'zeros'); self._get(n + '/gamma', (C,), 'gamma')
        self._get(n + '/moving_mean', (C,), 'zeros'); self._get(n + '/moving_variance', (C,), 'ones')
        return _bn(self.P, n, x, train, stats)


def _ca(V, embed, noise):
    """generate_conditionals + sample_normal_conditional (stageI/model.py:59-75)."""
    mean = _lrelu(V.dense(embed, V.cfg.compressed, 'n02'))
    log_sigma = _lrelu(V.dense(embed, V.cfg.compressed, 'n02'))
    code = mean if noise is None else mean + torch.exp(log_sigma) * noise
    return code, mean, log_sigma


def kl_loss(mean, log_sigma):
    return torch.mean(-log_sigma + 0.5 * (-1.0 + torch.exp(2.0 * log_sigma) + mean ** 2))


# ------------------------------------------------------------------------------------------------------------------
# Stage I (64x64)
# ------------------------------------------------------------------------------------------------------------------
def stage1_generator(V, cfg, z, embed, noise, train=True, stats=None):
    """stageI/model.py:123-171 -> (img NHWC, mean, log_sigma)"""
    V.enter('g_net'); V.cfg = cfg
    g, B = cfg.gf, z.shape[0]
    code, mean, log_sigma = _ca(V, embed, noise)
    h = V.dense(torch.cat([z, code], 1), g * 8 * 16, 'n02')
    h = V.bn(h, train, stats)
    h0 = h.reshape(B, 4, 4, g * 8).permute(0, 3, 1, 2)
    bn = lambda x: V.bn(x, train, stats)
    r = _relu(bn(V.conv(h0, g * 2, 1, 1, 'VALID'))); r = _relu(bn(V.conv(r, g * 2, 3))); r = bn(V.conv(r, g * 8, 3))
    h1 = _relu(h0 + r)
    h2 = bn(V.conv(V.deconv(h1, g * 4), g * 4, 3))
    r = _relu(bn(V.conv(h2, g, 1, 1, 'VALID'))); r = _relu(bn(V.conv(r, g, 3))); r = bn(V.conv(r, g * 4, 3))
    h3 = _relu(h2 + r)
    h4 = _relu(bn(V.conv(V.deconv(h3, g * 2), g * 2, 3)))
    h5 = _relu(bn(V.conv(V.deconv(h4, g), g, 3)))
    out = torch.tanh(V.conv(V.deconv(h5, cfg.channels), cfg.channels, 3))
    return out.permute(0, 2, 3, 1), mean, log_sigma


def stage1_discriminator(V, cfg, img_nhwc, embed, train=True, stats=None):
    """stageI/model.py:77-121 -> logits [B,1,1,1]"""
    V.enter('d_net'); V.cfg = cfg
    d = cfg.df
    x = img_nhwc.permute(0, 3, 1, 2)
    bn = lambda t: V.bn(t, train, stats)
    h0 = _lrelu(V.conv(x, d, 4, 2))
    h1 = _lrelu(bn(V.conv(h0, d * 2, 4, 2))); h2 = _lrelu(bn(V.conv(h1, d * 4, 4, 2))); h3 = bn(V.conv(h2, d * 8, 4, 2))
    r = _lrelu(bn(V.conv(h3, d * 2, 1, 1, 'VALID'))); r = _lrelu(bn(V.conv(r, d * 2, 3))); r = bn(V.conv(r, d * 8, 3))
    h4 = _lrelu(h3 + r)
    e = _lrelu(V.dense(embed, cfg.compressed, 'glorot'))
    e = e[:, :, None, None].expand(-1, -1, 4, 4)
    h = _lrelu(bn(V.conv(torch.cat([h4, e], 1), d * 8, 1, 1, 'VALID')))
    s16 = cfg.out_size // 16
    return V.conv(h, 1, s16, s16, 'VALID')


# ------------------------------------------------------------------------------------------------------------------
# Stage II (256x256 from the Stage-I 64x64 image)
# ------------------------------------------------------------------------------------------------------------------
def stage2_generator(V, cfg, img64_nhwc, embed, noise, train=True, stats=None):
    """stageII/model.py:135-201 -> (img NHWC 256x256, mean, log_sigma).  Convs without an explicit init use He."""
    V.enter('stageII_g_net'); V.cfg = cfg
    g = cfg.gf
    bn = lambda t: V.bn(t, train, stats)
    x = img64_nhwc.permute(0, 3, 1, 2)
    e0 = _relu(V.conv(x, g, 3, 1, 'SAME', 'he'))
    e1 = _relu(bn(V.conv(e0, g * 2, 4, 2, 'SAME', 'he')))
    enc = _relu(bn(V.conv(e1, g * 4, 4, 2, 'SAME', 'he')))                       # [B, 4g, 16, 16]
    code, mean, log_sigma = _ca(V, embed, noise)
    tile = code[:, :, None, None].expand(-1, -1, 16, 16)
    h = _relu(bn(V.conv(torch.cat([enc, tile], 1), g * 4, 3, 1, 'SAME', 'he')))
    for _ in range(4):                                                             # generator_residual_layer: k4 s1 SAME
        r = _relu(bn(V.conv(h, g * 4, 4, 1, 'SAME', 'he')))
        r = bn(V.conv(r, g * 4, 4, 1, 'SAME', 'he'))
        h = _relu(h + r)
    for f in (g * 2, g, g // 2, g // 4):                                           # generator_upsample
        h = _relu(bn(V.conv(V.deconv(h, f, 'n02'), f, 3, 1, 'SAME', 'he')))
    out = torch.tanh(V.conv(h, cfg.channels, 3, 1, 'SAME', 'he'))
    return out.permute(0, 2, 3, 1), mean, log_sigma


def stage2_discriminator(V, cfg, img_nhwc, embed, train=True, stats=None):
    """stageII/model.py:78-133 -> logits [B,1,1,1]"""
    V.enter('stageII_d_net'); V.cfg = cfg
    d = cfg.df
    s16 = cfg.out_size // 64
    x = img_nhwc.permute(0, 3, 1, 2)
    bn = lambda t: V.bn(t, train, stats)
    h = _lrelu(V.conv(x, d, 4, 2))
    for mult in (2, 4, 8, 16, 32):
        h = _lrelu(bn(V.conv(h, d * mult, 4, 2)))
    h = _lrelu(bn(V.conv(h, d * 16, 4, 1)))
    h7 = bn(V.conv(h, d * 8, 4, 1))
    r = _lrelu(bn(V.conv(h7, d * 2, 1, 1))); r = _lrelu(bn(V.conv(r, d * 2, 3))); r = bn(V.conv(r, d * 8, 3))
    h8 = _lrelu(r + r)                                                             # tf.add(net, net): the trunk is dropped
    e = _lrelu(V.dense(embed, cfg.compressed, 'glorot'))
    e = e[:, :, None, None].expand(-1, -1, s16, s16)
    h9 = _lrelu(bn(V.conv(torch.cat([h8, e], 1), d * 8, 1, 1)))
    return V.conv(h9, 1, s16, s16, 'SAME')                                        # ops default padding; k == s: no pad


def sigmoid_ce(logits, label):
    return torch.mean(torch.clamp(logits, min=0) - logits * label + torch.log1p(torch.exp(-logits.abs())))


# ------------------------------------------------------------------------------------------------------------------
# steps.  `stage` = 1 or 2; for stage 2 the generator input is the Stage-I generator's output (training mode, frozen).
# ------------------------------------------------------------------------------------------------------------------
def _gen(P, cfg, cfg1, stage, feed, noise_key, stats):
    V = Vars(P)
    if stage == 1:
        return stage1_generator(V, cfg, feed['z'], feed['cond'], feed[noise_key], True, stats)
    with torch.no_grad():     # frozen: not in any var_list (stageII/model.py:62-63); its own CA noise is an in-graph draw
        img64, _, _ = stage1_generator(V, cfg1, feed['z'], feed['cond'], feed[noise_key + '_s1'], True, stats)
    return stage2_generator(V, cfg, img64, feed['cond'], feed[noise_key], True, stats)


def _disc(P, cfg, stage, img, cond, stats):
    V = Vars(P)
    return (stage1_discriminator if stage == 1 else stage2_discriminator)(V, cfg, img, cond, True, stats)


def scopes(stage):
    return ('g_net', 'd_net') if stage == 1 else ('stageII_g_net', 'stageII_d_net')


def d_step(P, cfg, feed, stage=1, cfg1=None, term_scales=False, bias_l1=False):
    """term_scales: see oracle/torch_gancls.d_step (per tensor, the gradient magnitude before the three loss terms cancel).
    bias_l1: also return, per bias variable, sum |g| of the gradient reaching its layer's output (_BIAS_L1)."""
    _BIAS_L1[0] = {} if bias_l1 else None
    try:
        return _d_step(P, cfg, feed, stage, cfg1, term_scales)
    finally:
        _BIAS_L1[0] = None


def _d_step(P, cfg, feed, stage, cfg1, term_scales):
    gs, ds = scopes(stage)
    names = trainable(P, ds)
    Q = dict(P)
    for n in names:
        Q[n] = P[n].detach().requires_grad_(True)
    gstats, dstats = {}, [{}, {}, {}]
    with torch.no_grad(), tape_section('G'):
        G, _, _ = _gen(P, cfg, cfg1, stage, feed, 'ca_noise_d', gstats)
    with tape_section('Dfake'):
        lf = _disc(Q, cfg, stage, G, feed['cond'], dstats[0])
    with tape_section('Dmatch'):
        lm = _disc(Q, cfg, stage, feed['x'], feed['cond'], dstats[1])
    with tape_section('Dmis'):
        lw = _disc(Q, cfg, stage, feed['x_mismatch'], feed['cond'], dstats[2])
    fake, match, mism = sigmoid_ce(lf, 0.0), sigmoid_ce(lm, cfg.real_label), sigmoid_ce(lw, 0.0)
    D_loss = match + cfg.alpha * mism + (1.0 - cfg.alpha) * fake
    scales = None
    if term_scales:
        scales = OrderedDict((n, 0.0) for n in names)
        for coef, term in ((1.0, match), (cfg.alpha, mism), (1.0 - cfg.alpha, fake)):
            gs_ = torch.autograd.grad(term, [Q[n] for n in names], retain_graph=True, allow_unused=True)
            for n, g in zip(names, gs_):
                if g is not None:
                    scales[n] = max(scales[n], abs(coef) * float(g.abs().max()))
    grads = torch.autograd.grad(D_loss, [Q[n] for n in names])
    f = lambda t: float(t.detach())
    return dict(D_loss=f(D_loss), D_real_match_loss=f(match), D_real_mismatch_loss=f(mism), D_synthetic_loss=f(fake),
                grads=OrderedDict((n, g.detach()) for n, g in zip(names, grads)), G=G.detach(), g_stats=gstats, d_stats=dstats,
                scales=scales, bias_l1=_BIAS_L1[0])


def g_step(P, cfg, feed, stage=1, cfg1=None, bias_l1=False):
    _BIAS_L1[0] = {} if bias_l1 else None
    try:
        return _g_step(P, cfg, feed, stage, cfg1)
    finally:
        _BIAS_L1[0] = None


def _g_step(P, cfg, feed, stage, cfg1):
    gs, ds = scopes(stage)
    names = trainable(P, gs)
    Q = dict(P)
    for n in names:
        Q[n] = P[n].detach().requires_grad_(True)
    gstats, dstats = {}, [{}, {}, {}]
    with tape_section('G'):
        G, mean, log_sigma = _gen(Q, cfg, cfg1, stage, feed, 'ca_noise_g', gstats)
    with tape_section('Dfake'):
        lf = _disc(Q, cfg, stage, G, feed['cond'], dstats[0])
    G_gan, G_kl = sigmoid_ce(lf, 1.0), kl_loss(mean, log_sigma)
    G_loss = G_gan + cfg.kl * G_kl
    grads = torch.autograd.grad(G_loss, [Q[n] for n in names])
    with torch.no_grad():     # G_optim sits under ALL update ops: the two real critic passes run for their moving averages
        with tape_section('Dmatch'):
            _disc(P, cfg, stage, feed['x'], feed['cond'], dstats[1])
        with tape_section('Dmis'):
            _disc(P, cfg, stage, feed['x_mismatch'], feed['cond'], dstats[2])
    f = lambda t: float(t.detach())
    return dict(G_loss=f(G_loss), G_gan_loss=f(G_gan), G_kl_loss=f(G_kl), G=G.detach(),
                grads=OrderedDict((n, g.detach()) for n, g in zip(names, grads)), g_stats=gstats, d_stats=dstats, bias_l1=_BIAS_L1[0])


def apply_moving(P, gstats, dstats, decay=0.9):
    def upd(name, mean, var, n):
        P[name + '/moving_mean'] = decay * P[name + '/moving_mean'] + (1 - decay) * mean
        P[name + '/moving_variance'] = decay * P[name + '/moving_variance'] + (1 - decay) * var * (n / max(n - 1, 1))
    for name, (mean, var, n) in gstats.items():
        upd(name, mean, var, n)
    for st in dstats:          # fake, match, mismatch — graph build order (model.py: D_synthetic, D_real_match, D_real_mismatch)
        for name, (mean, var, n) in st.items():
            upd(name, mean, var, n)


def init_variables(cfg, stage=1, cfg1=None, seed=0, dtype=torch.float64):
    """All variables of the stage's graph in creation order (stage 2: Stage-I generator first, stageII/model.py:50-53)."""
    V = Vars(None, seed, dtype)
    B = 2
    z = torch.zeros(B, (cfg1 or cfg).z_dim, dtype=dtype); cond = torch.zeros(B, cfg.embed_dim, dtype=dtype)
    with torch.no_grad():
        if stage == 1:
            G, _, _ = stage1_generator(V, cfg, z, cond, None)
            stage1_discriminator(V, cfg, G, cond)
        else:
            img64, _, _ = stage1_generator(V, cfg1, z, cond, None)
            G, _, _ = stage2_generator(V, cfg, img64, cond, None)
            stage2_discriminator(V, cfg, G, cond)
    return V.P


class Trainer(object):
    """stageI/trainer.py:119-147 / stageII/trainer.py:130-160: per update lr = LR * 0.5 ** (epoch // 100); D then G."""

    def __init__(self, cfg, P, stage=1, cfg1=None):
        self.cfg, self.P, self.stage, self.cfg1 = cfg, P, stage, cfg1
        gs, ds = scopes(stage)
        self.opt_d = AdamTF(trainable(P, ds), P, cfg.beta1, 0.999)
        self.opt_g = AdamTF(trainable(P, gs), P, cfg.beta1, 0.999)

    def iteration(self, feed, epoch=0):
        lr = self.cfg.lr * (0.5 ** (epoch // 100))
        d = d_step(self.P, self.cfg, feed, self.stage, self.cfg1)
        self.opt_d.apply(self.P, d['grads'], lr)
        apply_moving(self.P, d['g_stats'], d['d_stats'])
        g = g_step(self.P, self.cfg, feed, self.stage, self.cfg1)
        self.opt_g.apply(self.P, g['grads'], lr)
        apply_moving(self.P, g['g_stats'], g['d_stats'])
        return {'d': d, 'g': g}


def synthetic_feed(cfg, stage=1, cfg1=None, seed=1, dtype=torch.float64):
    rng = np.random.default_rng(seed)
    B, S = cfg.batch, cfg.out_size
    t = lambda a: torch.tensor(a, dtype=dtype)
    tn = lambda shape: np.clip(rng.standard_normal(shape), -2, 2)
    f = dict(x=t(rng.uniform(-1, 1, (B, S, S, cfg.channels))), x_mismatch=t(rng.uniform(-1, 1, (B, S, S, cfg.channels))),
             cond=t(rng.standard_normal((B, cfg.embed_dim))), z=t(rng.standard_normal((B, (cfg1 or cfg).z_dim))),
             ca_noise_d=t(tn((B, cfg.compressed))), ca_noise_g=t(tn((B, cfg.compressed))))
    if stage == 2:
        f['ca_noise_d_s1'] = t(tn((B, cfg1.compressed))); f['ca_noise_g_s1'] = t(tn((B, cfg1.compressed)))
    return f
