"""Oracle (b'''): torch-CPU autograd restatement of the conditional progressive-growing GAN step.

TEST INFRASTRUCTURE ONLY (see oracle/np_ops.py).  PARITY UNPINNED against TensorFlow; the operator
restatements are shared with oracle/torch_step.py, which is pinned against the loop oracle (a).

Follows reference models/pggan/pggan.py:
  * networks `generator` :283-316 and `discriminator` :251-281 for a given `stage` (output 4 * 2^(stage-1)) with or
    without the transition branch `t` (fade-in by alpha = iter / steps, :76-77,267,314); channel schedule `get_nf` /
    `get_dnf` :339-343; `to_rgb` :367-371 (k2 s1 SAME 9-channel relu conv then 1x1), `from_rgb` :345-347;
    conditioning augmentation :349-361; all kernels He-initialised (utils/ops.py defaults), layer norm only in G;
  * losses :84-107: D = -wdist - wdist2 + 200 (gp + gp2) (no kt), G = -D_fake + 5 KL, x_hat = eps G + (1-eps) x with
    eps drawn in-graph (:68), Adam(2e-6, beta1 = 0, beta2 = 0.99) hard-coded for both nets (:109-110).
  * operators utils/ops.py:74-81 layer_norm (per-sample over H,W,C, gamma/beta per channel, eps 1e-12), :100-101 pool
    (2x2 average), :109-111 upscale (nearest x2).
Variable names: tf.contrib.layers / tf.layers auto-names inside nested scopes (`g_net/conv_stage_1/Conv_1/weights`,
`d_net/rgb_stage_2/Conv/biases`, `g_net/conv_stage_0/LayerNorm_2/gamma`, `d_net/conv_stage_0/dense_1/kernel`).
`base` / `cap` generalise the hard-coded 1024 / 512 of the channel schedule so that the golden step can be tiny."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .torch_stackgan import Vars as _Vars, kl_loss
from .torch_step import AdamTF, trainable  # noqa: F401


class Cfg(object):
    def __init__(self, z_dim=128, embed_dim=1024, compressed=128, channels=3, batch=16, base=1024, cap=512, lr=2e-6,
                 beta1=0.0, beta2=0.99, gp_coeff=200.0, kl_coeff=5.0):
        self.z_dim, self.embed_dim, self.compressed, self.channels, self.batch = z_dim, embed_dim, compressed, channels, batch
        self.base, self.cap, self.lr, self.beta1, self.beta2 = base, cap, lr, beta1, beta2
        self.gp_coeff, self.kl_coeff = gp_coeff, kl_coeff

    def nf(self, stage):
        return min(self.base // (2 ** stage) * 4, self.cap)

    def dnf(self, stage):
        return min(self.base // (2 ** stage) * 2, self.cap)


class Vars(_Vars):
    """+ layer norm, He-initialised dense; `enter` takes the full nested scope path."""

    def ln(self, x, act=None):
        n = self._name('LayerNorm')
        C = x.shape[1]
        beta = self._get(n + '/beta', (C,), 'zeros'); gamma = self._get(n + '/gamma', (C,), 'ones')
        dims = tuple(range(1, x.dim()))
        mean = x.mean(dims, keepdim=True)
        var = ((x - mean) ** 2).mean(dims, keepdim=True)
        shape = (1, -1) + (1,) * (x.dim() - 2)
        y = (x - mean) / torch.sqrt(var + 1e-12) * gamma.view(shape) + beta.view(shape)
        return act(y) if act else y


from .torch_step import lrelu as _lrelu, relu as _relu, tape_section  # noqa: E402  (honour an installed SectionTape)


def _to_rgb(V, scope, x, stage, cfg):
    V.enter('%s/rgb_stage_%d' % (scope, stage))
    x = _relu(V.conv(x, 9, 2, 1, 'SAME', 'he'))
    return V.conv(x, cfg.channels, 1, 1, 'SAME', 'he')


def _from_rgb(V, scope, x, stage, cfg):
    V.enter('%s/rgb_stage_%d' % (scope, stage))
    return _lrelu(V.conv(x, cfg.dnf(stage), 1, 1, 'SAME', 'he'))


def generator(V, cfg, z, cond, stages, t, alpha, noise):
    """-> (img NHWC, mean, log_sigma); activations NCHW inside"""
    V.enter('g_net/conv_stage_0')
    mean = _lrelu(V.dense(cond, cfg.compressed, 'he'))
    log_sigma = _lrelu(V.dense(cond, cfg.compressed, 'he'))
    code = mean if noise is None else mean + torch.exp(log_sigma) * noise
    x = V.dense(torch.cat([z, code], 1), 4 * 4 * cfg.nf(0), 'he')
    x = V.ln(x)
    x = x.reshape(-1, 4, 4, cfg.nf(0)).permute(0, 3, 1, 2)
    x = V.ln(V.conv(x, cfg.nf(0), 3, 1, 'SAME', 'he'), _relu)
    x = V.ln(V.conv(x, cfg.nf(0), 3, 1, 'SAME', 'he'), _relu)
    x_iden = None
    for i in range(1, stages):
        if i == stages - 1 and t:
            x_iden = _to_rgb(V, 'g_net', x, stages - 2, cfg)
            x_iden = F.interpolate(x_iden, scale_factor=2, mode='nearest')
        V.enter('g_net/conv_stage_%d' % i)
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        x = V.ln(V.conv(x, cfg.nf(i), 3, 1, 'SAME', 'he'), _relu)
        x = V.ln(V.conv(x, cfg.nf(i), 3, 1, 'SAME', 'he'), _relu)
    x = _to_rgb(V, 'g_net', x, stages - 1, cfg)
    if t:
        x = (1.0 - alpha) * x_iden + alpha * x
    return x.permute(0, 2, 3, 1), mean, log_sigma


def discriminator(V, cfg, img_nhwc, cond, stages, t, alpha):
    """-> logits [B]"""
    inp = img_nhwc.permute(0, 3, 1, 2)
    x_iden = None
    if t:
        x_iden = _from_rgb(V, 'd_net', F.avg_pool2d(inp, 2), stages - 2, cfg)
    x = _from_rgb(V, 'd_net', inp, stages - 1, cfg)
    for i in range(stages - 1, 0, -1):
        V.enter('d_net/conv_stage_%d' % i)
        x = _lrelu(V.conv(x, cfg.dnf(i), 3, 1, 'SAME', 'he'))
        x = _lrelu(V.conv(x, cfg.dnf(i - 1), 3, 1, 'SAME', 'he'))
        x = F.avg_pool2d(x, 2)
        if i == stages - 1 and t:
            x = alpha * x + (1.0 - alpha) * x_iden
    V.enter('d_net/conv_stage_0')
    e = _lrelu(V.dense(cond, cfg.compressed, 'he'))          # fc(cond, units=128): the compressed size, hard-coded
    x = torch.cat([x, e[:, :, None, None].expand(-1, -1, 4, 4)], 1)
    x = _lrelu(V.conv(x, cfg.dnf(0), 3, 1, 'SAME', 'he'))
    x = _lrelu(V.conv(x, cfg.dnf(0), 4, 1, 'VALID', 'he'))
    return V.dense(x.reshape(x.shape[0], -1), 1, 'he').reshape(-1)


def init_variables(cfg, stages, t, seed=0, dtype=torch.float64):
    V = Vars(None, seed, dtype)
    V.cfg = cfg
    z = torch.zeros(2, cfg.z_dim, dtype=dtype); cond = torch.zeros(2, cfg.embed_dim, dtype=dtype)
    with torch.no_grad():
        G, _, _ = generator(V, cfg, z, cond, stages, t, 0.5, None)
        discriminator(V, cfg, G, cond, stages, t, 0.5)
    return V.P


def _gp(grad):
    slopes = torch.sqrt((grad ** 2).reshape(grad.shape[0], -1).sum(1))
    return torch.mean(torch.clamp(slopes - 1.0, min=0.0) ** 2)


def d_step(P, cfg, feed, stages, t, alpha):
    names = trainable(P, 'd_net')
    Q = dict(P)
    for n in names:
        Q[n] = P[n].detach().requires_grad_(True)
    with torch.no_grad(), tape_section('G'):
        G, _, _ = generator(Vars(P), cfg, feed['z'], feed['cond'], stages, t, alpha, feed['ca_noise_d'])
    D = lambda img, c: discriminator(Vars(Q), cfg, img, c, stages, t, alpha)
    with tape_section('Dg'):
        Dg = D(G, feed['cond'])
    with tape_section('Dx'):
        Dx = D(feed['x'], feed['cond'])
    with tape_section('Dxmi'):
        Dxmi = D(feed['x_mismatch'], feed['cond'])
    eps = feed['eps'].reshape(-1, 1, 1, 1)
    x_hat = (eps * G + (1.0 - eps) * feed['x']).detach().requires_grad_(True)
    cond_inp = feed['cond'].detach().clone().requires_grad_(True)
    with tape_section('Dxh'):
        Dx_hat = D(x_hat, cond_inp)
    gx, gc = torch.autograd.grad(Dx_hat.sum(), [x_hat, cond_inp], create_graph=True)
    real_gp, real_gp2 = _gp(gx), _gp(gc)
    wdist, wdist2 = Dx.mean() - Dg.mean(), Dx.mean() - Dxmi.mean()
    D_loss = -wdist - wdist2 + cfg.gp_coeff * (real_gp + real_gp2)
    grads = torch.autograd.grad(D_loss, [Q[n] for n in names])
    f = lambda v: float(v.detach())
    return dict(D_loss=f(D_loss), wdist=f(wdist), wdist2=f(wdist2), real_gp=f(real_gp), real_gp2=f(real_gp2),
                grads=OrderedDict((n, g.detach()) for n, g in zip(names, grads)), G=G.detach(), Dx_hat=Dx_hat.detach())


def g_step(P, cfg, feed, stages, t, alpha):
    names = trainable(P, 'g_net')
    Q = dict(P)
    for n in names:
        Q[n] = P[n].detach().requires_grad_(True)
    with tape_section('G'):
        G, mean, log_sigma = generator(Vars(Q), cfg, feed['z'], feed['cond'], stages, t, alpha, feed['ca_noise_g'])
    with tape_section('Dg'):
        Dg = discriminator(Vars(Q), cfg, G, feed['cond'], stages, t, alpha)
    G_kl = kl_loss(mean, log_sigma)
    G_loss = -Dg.mean() + cfg.kl_coeff * G_kl
    grads = torch.autograd.grad(G_loss, [Q[n] for n in names])
    f = lambda v: float(v.detach())
    return dict(G_loss=f(G_loss), G_kl_loss=f(G_kl), G=G.detach(),
                grads=OrderedDict((n, g.detach()) for n, g in zip(names, grads)))


class Trainer(object):
    """pggan.py:147-203: per iteration alpha = idx / steps, D update then G update."""

    def __init__(self, cfg, P, stages, t, steps):
        self.cfg, self.P, self.stages, self.t, self.steps = cfg, P, stages, t, steps
        self.opt_d = AdamTF(trainable(P, 'd_net'), P, cfg.beta1, cfg.beta2)
        self.opt_g = AdamTF(trainable(P, 'g_net'), P, cfg.beta1, cfg.beta2)

    def iteration(self, idx, feed):
        alpha = idx / float(self.steps)
        d = d_step(self.P, self.cfg, feed, self.stages, self.t, alpha)
        self.opt_d.apply(self.P, d['grads'], self.cfg.lr)
        g = g_step(self.P, self.cfg, feed, self.stages, self.t, alpha)
        self.opt_g.apply(self.P, g['grads'], self.cfg.lr)
        return {'d': d, 'g': g}


def synthetic_feed(cfg, stages, seed=1, dtype=torch.float64):
    rng = np.random.default_rng(seed)
    B, S = cfg.batch, 4 * 2 ** (stages - 1)
    t = lambda a: torch.tensor(a, dtype=dtype)
    tn = lambda shape: np.clip(rng.standard_normal(shape), -2, 2)
    return dict(x=t(rng.uniform(-1, 1, (B, S, S, cfg.channels))), x_mismatch=t(rng.uniform(-1, 1, (B, S, S, cfg.channels))),
                cond=t(rng.standard_normal((B, cfg.embed_dim))), z=t(rng.standard_normal((B, cfg.z_dim))),
                eps=t(rng.uniform(0, 1, (B,))), ca_noise_d=t(tn((B, cfg.compressed))), ca_noise_g=t(tn((B, cfg.compressed))))
