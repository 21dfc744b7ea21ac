"""Oracle (b): torch-CPU autograd restatement of the whole wgancls / gancls training iteration.

TEST INFRASTRUCTURE ONLY (see oracle/np_ops.py header).  Used as (1) the full-shape comparator for the
HIP path on the GPU box, (2) the generator of the tiny-model golden step fixture, (3) the timed
``cpu_baseline`` ("port") in bench.py.  It never runs on a GPU and the product never imports it.

PARITY UNPINNED against TensorFlow itself (no TF here, no upstream tests).  It is pinned
instead against oracle (a) (``np_ops``: direct float64 loops) in tests/test_oracle.py, and its
double-backward is pinned by a finite-difference check there.

Follows, line by line in *behaviour* (not text):
  reference models/wgancls/model.py:34-60   build_model: G, 3 critic passes, x_hat, D(x_hat)
  reference models/wgancls/model.py:62-70   the two one-sided gradient penalties
  reference models/wgancls/model.py:72-106  losses, Adam(D), SGD(kt), Adam(G) under UPDATE_OPS
  reference models/wgancls/model.py:108-127 conditioning augmentation + KL
  reference models/wgancls/model.py:129-161 discriminator     (variable names: SURVEY.md appendix A)
  reference models/wgancls/model.py:163-225 generator
  reference models/wgancls/trainer.py:73-102 iteration order: D step (+kt) then G step

Parameters live in a flat dict keyed by the TF-1.x variable names (``d_net/Conv_3/weights`` ...), in TF
layouts (conv HWIO, deconv [kh,kw,Cout,Cin], dense [in,out]).  Activations are NCHW like the
reference (df=NCHW everywhere in wgancls), so the dense_2 -> [B,C,4,4] reshape needs no permutation.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import np_ops


class Cfg(object):
    """Model dims (reference models/wgancls/cfg/flowers.yml:9-37)."""

    def __init__(self, z_dim=128, embed_dim=1024, compressed=128, gf=128, df=128, out_size=64, channels=3,
                 batch=64, beta1=0.0, beta2=0.9, kl_coeff=1.0, gp_coeff=150.0, kt_lr=1e-3):
        self.z_dim, self.embed_dim, self.compressed = z_dim, embed_dim, compressed
        self.gf, self.df, self.out_size, self.channels, self.batch = gf, df, out_size, channels, batch
        self.beta1, self.beta2, self.kl_coeff, self.gp_coeff, self.kt_lr = beta1, beta2, kl_coeff, gp_coeff, kt_lr
        assert out_size == 64, 'the reference hard-codes a 4x4 text tile (model.py:154) => 64x64 images'


# ----------------------------------------------------------------------------------------------
# variable registry (names/shapes as TF would auto-name them)
# ----------------------------------------------------------------------------------------------
def variable_shapes(cfg):
    """-> OrderedDict name -> (shape, kind, fan_in); kind in {'w','b','gamma','beta','mm','mv'}."""
    V = OrderedDict()

    def conv(scope, idx, kh, kw, ci, co):
        n = '%s/Conv%s' % (scope, '' if idx == 0 else '_%d' % idx)
        V[n + '/weights'] = ((kh, kw, ci, co), 'w', kh * kw * ci)
        V[n + '/biases'] = ((co,), 'b', 0)

    def deconv(scope, idx, kh, kw, co, ci):
        n = '%s/Conv2d_transpose%s' % (scope, '' if idx == 0 else '_%d' % idx)
        # He fan_in quirk: variance_scaling uses shape[-2] = Cout of the deconv
        V[n + '/weights'] = ((kh, kw, co, ci), 'w', kh * kw * co)
        V[n + '/biases'] = ((co,), 'b', 0)

    def dense(scope, idx, i, o):
        n = '%s/dense%s' % (scope, '' if idx == 0 else '_%d' % idx)
        V[n + '/kernel'] = ((i, o), 'w', i)
        V[n + '/bias'] = ((o,), 'b', 0)

    def bn(scope, idx, c):
        n = '%s/BatchNorm%s' % (scope, '' if idx == 0 else '_%d' % idx)
        V[n + '/beta'] = ((c,), 'beta', 0)
        V[n + '/gamma'] = ((c,), 'gamma', 0)
        V[n + '/moving_mean'] = ((c,), 'mm', 0)
        V[n + '/moving_variance'] = ((c,), 'mv', 0)

    d, g, ce, C = cfg.df, cfg.gf, cfg.compressed, cfg.channels
    # d_net (reference model.py:134-161)
    conv('d_net', 0, 4, 4, C, d); conv('d_net', 1, 4, 4, d, 2 * d); conv('d_net', 2, 4, 4, 2 * d, 4 * d)
    conv('d_net', 3, 4, 4, 4 * d, 8 * d)
    conv('d_net', 4, 1, 1, 8 * d, 2 * d); conv('d_net', 5, 3, 3, 2 * d, 4 * d); conv('d_net', 6, 3, 3, 4 * d, 8 * d)
    dense('d_net', 0, cfg.embed_dim, ce)
    conv('d_net', 7, 3, 3, 8 * d + ce, 8 * d); conv('d_net', 8, 1, 1, 8 * d, 8 * d); conv('d_net', 9, 4, 4, 8 * d, 1)
    # g_net (reference model.py:167-225)
    dense('g_net', 0, cfg.embed_dim, ce); dense('g_net', 1, cfg.embed_dim, ce)
    dense('g_net', 2, cfg.z_dim + ce, 8 * g * 16); bn('g_net', 0, 8 * g * 16)
    conv('g_net', 0, 1, 1, 8 * g, 2 * g); bn('g_net', 1, 2 * g)
    conv('g_net', 1, 3, 3, 2 * g, 2 * g); bn('g_net', 2, 2 * g)
    conv('g_net', 2, 3, 3, 2 * g, 8 * g); bn('g_net', 3, 8 * g)
    deconv('g_net', 0, 4, 4, 4 * g, 8 * g); conv('g_net', 3, 3, 3, 4 * g, 4 * g); bn('g_net', 4, 4 * g)
    conv('g_net', 4, 1, 1, 4 * g, g); bn('g_net', 5, g)
    conv('g_net', 5, 3, 3, g, g); bn('g_net', 6, g)
    conv('g_net', 6, 3, 3, g, 4 * g); bn('g_net', 7, 4 * g)
    deconv('g_net', 1, 4, 4, 2 * g, 4 * g); conv('g_net', 7, 3, 3, 2 * g, 2 * g); bn('g_net', 8, 2 * g)
    deconv('g_net', 2, 4, 4, g, 2 * g); conv('g_net', 8, 3, 3, g, g); bn('g_net', 9, g)
    deconv('g_net', 3, 4, 4, C, g); conv('g_net', 9, 3, 3, C, C)
    return V


def init_variables(cfg, seed=0, dtype=torch.float32):
    """He-truncated-normal weights, zero biases, gamma=1, beta=0, moving stats (0,1)."""
    rng = np.random.default_rng(seed)
    P = OrderedDict()
    for name, (shape, kind, fan_in) in variable_shapes(cfg).items():
        if kind == 'w':
            a = np_ops.truncated_normal(rng, shape, np_ops.he_std(fan_in))
        elif kind in ('gamma', 'mv'):
            a = np.ones(shape)
        else:
            a = np.zeros(shape)
        P[name] = torch.tensor(a, dtype=dtype)
    return P


def is_trainable(name):
    return not (name.endswith('moving_mean') or name.endswith('moving_variance'))


def trainable(P, scope):
    return [n for n in P if n.startswith(scope + '/') and is_trainable(n)]


# ----------------------------------------------------------------------------------------------
# operator restatements on NCHW activations with TF-layout weights
# ----------------------------------------------------------------------------------------------
def _conv(x, w_hwio, b, stride, padding):
    KH, KW = w_hwio.shape[0], w_hwio.shape[1]
    H, W = x.shape[2], x.shape[3]
    if padding.upper() == 'SAME':
        _, pt, pb = np_ops.same_pad(H, KH, stride)
        _, pl, pr = np_ops.same_pad(W, KW, stride)
        if pt or pb or pl or pr:
            x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w_hwio.permute(3, 2, 0, 1), b, stride=stride)


def _deconv_k4s2(x, w, b):
    """TF conv2d_transpose k4 s2 SAME, weights [kh,kw,Cout,Cin] == torch conv_transpose2d(stride 2, padding 1)
    with weight.permute(3,2,0,1) (= [Cin,Cout,kh,kw]), no spatial flip (pinned vs np_ops)."""
    assert w.shape[0] == 4 and w.shape[1] == 4
    return F.conv_transpose2d(x, w.permute(3, 2, 0, 1), b, stride=2, padding=1)


class MaskTape(object):
    """Pins the branch every lrelu / relu takes.  A piecewise-linear net evaluated in two arithmetics (fp32 MFMA chains on
    the GPU, float64 here) agrees to rounding EXCEPT at units whose pre-activation lies within rounding distance of the
    kink: there the two sides pick different slopes and that unit's gradient differs by O(itself).  To compare gradients at
    the tolerance SURVEY.md §8(c) states (1e-4) the parity tests record the branch decisions of one side (`record`) and
    replay them on the other (`masks`): both then differentiate the SAME piecewise-linear function; a replayed unit whose
    own pre-activation has the other sign is within rounding distance of zero, so values move by rounding only.
    masks / record: lists of bool tensors in call order (NCHW for feature maps, [B, C] for dense layers)."""

    def __init__(self, masks=None, keep_values=False):
        self.masks, self.record, self.i = masks, [], 0
        self.values = [] if keep_values else None      # the activation outputs, in call order (per-layer error tables: tools/bf16_error_table.py)

    def act(self, x, slope):
        if self.masks is None:
            m = x > 0
        else:
            m = self.masks[self.i]
            assert m.shape == x.shape, ('mask %d' % self.i, tuple(m.shape), tuple(x.shape))
        self.i += 1
        self.record.append(m)
        y = x * torch.where(m, torch.ones((), dtype=x.dtype), torch.full((), slope, dtype=x.dtype))
        if self.values is not None:
            self.values.append(y.detach())
        return y


def _lrelu(x, tape=None):
    return F.leaky_relu(x, 0.2) if tape is None else tape.act(x, 0.2)


def _relu(x, tape=None):
    return F.relu(x) if tape is None else tape.act(x, 0.0)


class SectionTape(object):
    """MaskTape for the other oracles (torch_stackgan, torch_pggan, torch_gancls), installed globally with `use_tape` instead
    of being threaded through every call: activations are grouped into named sections, one per network pass
    (`tape_section('G')`, `'Dfake'`, ...), so that the GPU side — which may batch several passes into one — can be matched
    section by section.  masks: {section: [bool tensors in call order]} to replay, or None to record (`.record`)."""

    def __init__(self, masks=None):
        self.masks, self.record, self.sec, self.pos = masks, OrderedDict(), None, {}

    def act(self, x, slope):
        i = self.pos.get(self.sec, 0)
        self.pos[self.sec] = i + 1
        if self.masks is None:
            m = x > 0
        else:
            m = self.masks[self.sec][i]
            assert m.shape == x.shape, (self.sec, i, tuple(m.shape), tuple(x.shape))
        self.record.setdefault(self.sec, []).append(m)
        return x * torch.where(m, torch.ones((), dtype=x.dtype), torch.full((), slope, dtype=x.dtype))


TAPE = [None]


class use_tape(object):
    def __init__(self, tape):
        self.tape = tape

    def __enter__(self):
        self.prev, TAPE[0] = TAPE[0], self.tape
        return self.tape

    def __exit__(self, *a):
        TAPE[0] = self.prev


class forward_only(object):
    """Run an oracle step for its FORWARD passes only: inside, torch.autograd.grad returns zeros (nothing is differentiated) and no
    autograd graph is built.  For the branch-recording pass of a mask-pinned comparison (tests/branches.py): the recording tape needs
    the activations' signs and shapes, not the gradients — the float64 backward (and, for the WGAN-GP steps, the double backward) is
    two thirds of a step's CPU time.  Everything the step returns inside this context is meaningless except what the tape recorded."""

    def __enter__(self):
        self.real = torch.autograd.grad
        torch.autograd.grad = lambda outputs, inputs, *a, **k: tuple(torch.zeros_like(t) for t in inputs)
        self.ng = torch.no_grad()
        self.ng.__enter__()
        return self

    def __exit__(self, *a):
        self.ng.__exit__(*a)
        torch.autograd.grad = self.real


class tape_section(object):
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        t = TAPE[0]
        if t is not None:
            self.prev, t.sec = t.sec, self.name

    def __exit__(self, *a):
        t = TAPE[0]
        if t is not None:
            t.sec = self.prev


def lrelu(x):
    """leaky relu (0.2) that honours the globally installed SectionTape"""
    t = TAPE[0]
    return F.leaky_relu(x, 0.2) if t is None else t.act(x, 0.2)


def relu(x):
    t = TAPE[0]
    return F.relu(x) if t is None else t.act(x, 0.0)


def _bn(P, name, x, train, stats_out, eps=1e-5):
    gamma, beta = P[name + '/gamma'], P[name + '/beta']
    dims = (0,) if x.dim() == 2 else (0, 2, 3)
    shape = (1, -1) if x.dim() == 2 else (1, -1, 1, 1)
    if train:
        mean = x.mean(dims)
        var = ((x - mean.view(shape)) ** 2).mean(dims)
        if stats_out is not None:
            n = x.numel() // x.shape[1]
            stats_out[name] = (mean.detach(), var.detach(), n)
    else:
        mean, var = P[name + '/moving_mean'], P[name + '/moving_variance']
    return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + eps) * gamma.view(shape) + beta.view(shape)


def generator(P, cfg, z, embed, ca_noise, train=True, stats_out=None, tape=None, aux=None):
    """-> (img NHWC in [-1,1], mean, log_sigma).  ca_noise: explicit truncated-normal draw [B,compressed]
    (the reference resamples in-graph, model.py:119) or None for cond_noise=False.  tape: optional MaskTape;
    aux: optional dict that receives 'logits_absmax', the scale of the pre-tanh output (the tests' yardstick for G)."""
    g = cfg.gf
    B = z.shape[0]
    mean = _lrelu(embed @ P['g_net/dense/kernel'] + P['g_net/dense/bias'], tape)
    log_sigma = _lrelu(embed @ P['g_net/dense_1/kernel'] + P['g_net/dense_1/bias'], tape)
    c = mean + torch.exp(log_sigma) * ca_noise if ca_noise is not None else mean
    h = torch.cat([z, c], 1) @ P['g_net/dense_2/kernel'] + P['g_net/dense_2/bias']
    h = _bn(P, 'g_net/BatchNorm', h, train, stats_out)
    h0 = h.reshape(B, 8 * g, 4, 4)

    def cv(i, x, pad='SAME'):
        n = 'g_net/Conv%s' % ('' if i == 0 else '_%d' % i)
        return _conv(x, P[n + '/weights'], P[n + '/biases'], 1, pad)

    def dc(i, x):
        n = 'g_net/Conv2d_transpose%s' % ('' if i == 0 else '_%d' % i)
        return _deconv_k4s2(x, P[n + '/weights'], P[n + '/biases'])

    def bn(i, x):
        return _bn(P, 'g_net/BatchNorm_%d' % i, x, train, stats_out)

    r = _relu(bn(1, cv(0, h0, 'VALID')), tape)
    r = _relu(bn(2, cv(1, r)), tape)
    r = bn(3, cv(2, r))
    h1 = _relu(h0 + r, tape)
    h2 = bn(4, cv(3, dc(0, h1)))
    r = _relu(bn(5, cv(4, h2, 'VALID')), tape)
    r = _relu(bn(6, cv(5, r)), tape)
    r = bn(7, cv(6, r))
    h3 = _relu(h2 + r, tape)
    h4 = _relu(bn(8, cv(7, dc(1, h3))), tape)
    h5 = _relu(bn(9, cv(8, dc(2, h4))), tape)
    logits = cv(9, dc(3, h5))
    if aux is not None:
        aux['logits_absmax'] = float(logits.detach().abs().max())
    return torch.tanh(logits).permute(0, 2, 3, 1), mean, log_sigma


def discriminator(P, cfg, img_nhwc, embed, tape=None):
    """-> logit [B,1,1,1].  No batch norm: samples are independent.  tape: optional MaskTape."""
    x = img_nhwc.permute(0, 3, 1, 2)

    def cv(i, x, s, pad='SAME'):
        n = 'd_net/Conv%s' % ('' if i == 0 else '_%d' % i)
        return _conv(x, P[n + '/weights'], P[n + '/biases'], s, pad)

    h0 = _lrelu(cv(0, x, 2), tape); h1 = _lrelu(cv(1, h0, 2), tape); h2 = _lrelu(cv(2, h1, 2), tape); h3 = cv(3, h2, 2)
    r = _lrelu(cv(4, h3, 1, 'valid'), tape); r = _lrelu(cv(5, r, 1), tape); r = cv(6, r, 1)
    h4 = _lrelu(h3 + r, tape)
    e = _lrelu(embed @ P['d_net/dense/kernel'] + P['d_net/dense/bias'], tape)
    e = e[:, :, None, None].expand(-1, -1, 4, 4)
    h5 = _lrelu(cv(7, torch.cat([h4, e], 1), 1, 'same'), tape)
    h6 = _lrelu(cv(8, h5, 1, 'valid'), tape)
    return cv(9, h6, 4, 'valid')


def _gp(grad):
    slopes = torch.sqrt((grad.reshape(grad.shape[0], -1) ** 2).sum(1))
    return torch.mean(torch.clamp(slopes - 1.0, min=0.0) ** 2)


# ----------------------------------------------------------------------------------------------
# the two halves of one iteration
# ----------------------------------------------------------------------------------------------
def _tapes(masks, keys):
    """masks: None or {pass name: list of bool tensors} -> {pass name: MaskTape or None}"""
    return {k: (MaskTape(masks[k]) if masks is not None else None) for k in keys}


def d_step(P, cfg, feed, kt, masks=None):
    """Critic half (reference trainer.py:97; model.py:48-55,79-100).  All losses / gradients are taken at the
    pre-update values.  feed: x, x_mismatch [B,64,64,C] NHWC; cond [B,E]; z [B,Z]; eps [B,1,1,1]; ca_noise_d.
    masks: optional activation branches to replay, {'G','Dg','Dx','Dxmi','Dxh': [bool tensors]} (MaskTape).
    -> dict(scalars..., grads={name: tensor for d vars}, kt_grad, kt_new)"""
    names = trainable(P, 'd_net')
    Q = dict(P)
    for n in names:
        Q[n] = P[n].detach().requires_grad_(True)
    tp = _tapes(masks, ('G', 'Dg', 'Dx', 'Dxmi', 'Dxh'))
    with torch.no_grad():
        aux = {}
        G, _, _ = generator(P, cfg, feed['z'], feed['cond'], feed['ca_noise_d'], train=True, tape=tp['G'], aux=aux)
    x, xm, cond = feed['x'], feed['x_mismatch'], feed['cond']
    Dg = discriminator(Q, cfg, G, cond, tp['Dg'])
    Dx = discriminator(Q, cfg, x, cond, tp['Dx'])
    Dxmi = discriminator(Q, cfg, xm, cond, tp['Dxmi'])
    x_hat = (feed['eps'] * G + (1.0 - feed['eps']) * x).requires_grad_(True)
    cond_inp = (cond + 0.0).requires_grad_(True)
    Dxh = discriminator(Q, cfg, x_hat, cond_inp, tp['Dxh'])
    gx, gc = torch.autograd.grad(Dxh.sum(), [x_hat, cond_inp], create_graph=True)
    real_gp, real_gp2 = _gp(gx), _gp(gc)
    loss_real, loss_fake, loss_mis = Dx.mean(), Dg.mean(), Dxmi.mean()
    wdist, wdist2 = loss_real - loss_fake, loss_real - loss_mis
    D_loss = -wdist - kt * wdist2 + cfg.gp_coeff * (real_gp + real_gp2)
    grads = torch.autograd.grad(D_loss, [Q[n] for n in names])
    wd, wd2 = float(wdist.detach()), float(wdist2.detach())
    balance = (kt * wd2 - wd) ** 2
    kt_grad = 2.0 * (kt * wd2 - wd) * wd2
    f = lambda t: float(t.detach())
    return dict(D_loss=f(D_loss), D_loss_real=f(loss_real), D_loss_fake=f(loss_fake),
                D_loss_mismatch=f(loss_mis), wdist=wd, wdist2=wd2, real_gp=f(real_gp),
                real_gp2=f(real_gp2), reg_loss=f((Dxmi ** 2).mean()), balance_loss=balance,
                kt_grad=kt_grad, kt_new=kt - cfg.kt_lr * kt_grad,
                grads=OrderedDict((n, g.detach()) for n, g in zip(names, grads)),
                G=G.detach(), G_logits_absmax=aux['logits_absmax'], Dx_hat=Dxh.detach(), grad_x_hat=gx.detach(),
                grad_cond=gc.detach())


def g_step(P, cfg, feed, masks=None):
    """Generator half (reference trainer.py:100-102; model.py:87,92,102-106).  masks: {'G','Dg': [...]} (MaskTape).
    -> dict(G_loss, kl, grads, bn_stats)."""
    names = trainable(P, 'g_net')
    Q = dict(P)
    for n in names:
        Q[n] = P[n].detach().requires_grad_(True)
    stats = {}
    tp = _tapes(masks, ('G', 'Dg'))
    aux = {}
    G, mean, log_sigma = generator(Q, cfg, feed['z'], feed['cond'], feed['ca_noise_g'], train=True, stats_out=stats,
                                   tape=tp['G'], aux=aux)
    Dg = discriminator(Q, cfg, G, feed['cond'], tp['Dg'])
    kl = torch.mean(-log_sigma + 0.5 * (-1.0 + torch.exp(2.0 * log_sigma) + mean ** 2))
    G_loss = -Dg.mean() + cfg.kl_coeff * kl
    grads = torch.autograd.grad(G_loss, [Q[n] for n in names])
    f = lambda t: float(t.detach())
    return dict(G_loss=f(G_loss), G_kl_loss=f(kl), D_loss_fake=f(Dg.mean()),
                grads=OrderedDict((n, g.detach()) for n, g in zip(names, grads)), bn_stats=stats,
                G=G.detach(), G_logits_absmax=aux['logits_absmax'])


class AdamTF(object):
    """tf.train.AdamOptimizer: eps outside the bias correction."""

    def __init__(self, names, P, beta1, beta2, eps=1e-8):
        self.beta1, self.beta2, self.eps, self.t = beta1, beta2, eps, 0
        self.m = {n: torch.zeros_like(P[n]) for n in names}
        self.v = {n: torch.zeros_like(P[n]) for n in names}

    def apply(self, P, grads, lr):
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - self.beta2 ** self.t) / (1.0 - self.beta1 ** self.t)
        for n, g in grads.items():
            self.m[n].mul_(self.beta1).add_(g, alpha=1.0 - self.beta1)
            self.v[n].mul_(self.beta2).addcmul_(g, g, value=1.0 - self.beta2)
            P[n] = P[n] - lr_t * self.m[n] / (torch.sqrt(self.v[n]) + self.eps)


def apply_bn_moving(P, stats, decay=0.9):
    for name, (mean, var, n) in stats.items():
        P[name + '/moving_mean'] = decay * P[name + '/moving_mean'] + (1 - decay) * mean
        P[name + '/moving_variance'] = decay * P[name + '/moving_variance'] + (1 - decay) * var * (n / max(n - 1, 1))


class Trainer(object):
    """One iteration = D step (+kt) then G step, exactly the order of reference trainer.py:73-102."""

    def __init__(self, cfg, P, lr_d=1e-4, lr_g=1e-4, n_critic=1):
        self.cfg, self.P, self.kt = cfg, P, 0.7
        self.lr_d, self.lr_g, self.n_critic = lr_d, lr_g, n_critic
        self.opt_d = AdamTF(trainable(P, 'd_net'), P, cfg.beta1, cfg.beta2)
        self.opt_g = AdamTF(trainable(P, 'g_net'), P, cfg.beta1, cfg.beta2)

    def lr_scale(self, idx):
        return 0.95 ** ((idx // self.n_critic) // 10000)

    def iteration(self, idx, feed):
        s = self.lr_scale(idx)
        d = d_step(self.P, self.cfg, feed, self.kt)
        self.opt_d.apply(self.P, d['grads'], self.lr_d * s)
        self.kt = d['kt_new']
        out = {'d': d}
        if idx % self.n_critic == 0:
            g = g_step(self.P, self.cfg, feed)
            self.opt_g.apply(self.P, g['grads'], self.lr_g * s)
            apply_bn_moving(self.P, g['bn_stats'])
            out['g'] = g
        return out


def synthetic_feed(cfg, seed=1, dtype=torch.float32, batch=None):
    """BASELINE.md §2 synthetic inputs: x,x_mis ~ U[-1,1); cond,z ~ N(0,1); eps ~ U[0,1); CA noise trunc-normal."""
    B = batch or cfg.batch
    rng = np.random.default_rng(seed)
    t = lambda a: torch.tensor(a, dtype=dtype)
    return dict(
        x=t(rng.uniform(-1, 1, (B, 64, 64, cfg.channels))), x_mismatch=t(rng.uniform(-1, 1, (B, 64, 64, cfg.channels))),
        cond=t(rng.standard_normal((B, cfg.embed_dim))), z=t(rng.standard_normal((B, cfg.z_dim))),
        eps=t(rng.uniform(0, 1, (B, 1, 1, 1))),
        ca_noise_d=t(np_ops.truncated_normal(rng, (B, cfg.compressed))),
        ca_noise_g=t(np_ops.truncated_normal(rng, (B, cfg.compressed))))


def d_step_term_scales(P, cfg, feed, kt):
    """Per-variable magnitude of the individual terms of dD_loss/dtheta before they cancel.

    D_loss = -(1+kt)*mean D(x) + mean D(G) + kt*mean D(x_mis) + gp_coeff*(gp + gp2): at initialisation the three
    critic-mean terms carry a large sample-independent component whose coefficients sum to ZERO (-(1+kt) + 1 + kt), so
    some gradients (notably biases) are small differences of large numbers and amplify fp32 rounding by 1e2-1e4.  The
    parity tests therefore bound the error of a gradient by eps * (this un-cancelled scale), not by eps * |gradient|.
    -> {name: max_i |coef_i| * max|d term_i / d theta|}"""
    names = trainable(P, 'd_net')
    Q = dict(P)
    for n in names:
        Q[n] = P[n].detach().requires_grad_(True)
    with torch.no_grad():
        G, _, _ = generator(P, cfg, feed['z'], feed['cond'], feed['ca_noise_d'], train=True)
    x, xm, cond = feed['x'], feed['x_mismatch'], feed['cond']
    x_hat = (feed['eps'] * G + (1.0 - feed['eps']) * x).requires_grad_(True)
    cond_inp = (cond + 0.0).requires_grad_(True)
    Dxh = discriminator(Q, cfg, x_hat, cond_inp)
    gx, gc = torch.autograd.grad(Dxh.sum(), [x_hat, cond_inp], create_graph=True)
    terms = [(1.0 + kt, discriminator(Q, cfg, x, cond).mean()), (1.0, discriminator(Q, cfg, G, cond).mean()),
             (kt, discriminator(Q, cfg, xm, cond).mean()), (cfg.gp_coeff, _gp(gx)), (cfg.gp_coeff, _gp(gc))]
    scale = {n: 0.0 for n in names}
    for coef, t in terms:
        gs = torch.autograd.grad(t, [Q[n] for n in names], allow_unused=True, retain_graph=True)
        for n, g in zip(names, gs):
            if g is not None:
                scale[n] = max(scale[n], abs(coef) * float(g.abs().max()))
    return scale
