"""Generates the committed golden fixtures from the oracle (run from the repo root):

    python tests/golden/make_golden.py

The reference has no golden vectors of its own and cannot be imported (TF-1.4), so these are
produced by the float64 oracle: per-op cases by the direct-loop NumPy oracle (a), the tiny full-model step by the
torch-CPU autograd oracle (b) in float64.  Fixtures are data only (inputs + expected outputs).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import np_ops as O  # noqa: E402
from oracle import torch_step as T  # noqa: E402

# (name, B, H, W, Cin, Cout, KH, KW, stride, padding): covers every conv class on the hot path + "next" rows' pads
CONV_CASES = [
    ('k4s2_same', 2, 8, 8, 8, 8, 4, 4, 2, 'SAME'),
    ('k4s2_same_c3', 2, 8, 8, 3, 8, 4, 4, 2, 'SAME'),
    ('k4s2_same_odd', 1, 7, 5, 4, 8, 4, 4, 2, 'SAME'),
    ('k3s1_same', 2, 4, 4, 8, 8, 3, 3, 1, 'SAME'),
    ('k3s1_same_c3', 2, 8, 8, 3, 3, 3, 3, 1, 'SAME'),
    ('k1s1_valid', 2, 4, 4, 8, 12, 1, 1, 1, 'VALID'),
    ('k4s4_valid', 3, 4, 4, 8, 1, 4, 4, 4, 'VALID'),
    ('k4s1_same', 1, 6, 6, 4, 4, 4, 4, 1, 'SAME'),     # asymmetric pad (1,2)
    ('k2s1_same', 1, 5, 5, 4, 4, 2, 2, 1, 'SAME'),     # asymmetric pad (0,1)
]
DECONV_CASES = [
    ('dk4s2', 2, 4, 4, 8, 8),
    ('dk4s2_c3', 2, 8, 8, 8, 3),
]

SCALE_D = 1.25
TINY = dict(z_dim=8, embed_dim=32, compressed=16, gf=8, df=8, batch=4)


def make_ops():
    rng = np.random.default_rng(1234)
    out = {}
    for name, B, H, W, Ci, Co, KH, KW, s, pad in CONV_CASES:
        x = rng.standard_normal((B, H, W, Ci)); w = rng.standard_normal((KH, KW, Ci, Co)); b = rng.standard_normal(Co)
        y = O.conv2d(x, w, b, (s, s), pad)
        dy = rng.standard_normal(y.shape)
        out['conv/%s/x' % name] = x; out['conv/%s/w' % name] = w; out['conv/%s/b' % name] = b
        out['conv/%s/y' % name] = y; out['conv/%s/dy' % name] = dy
        out['conv/%s/dx' % name] = O.conv2d_bwd_data(dy, w, x.shape, (s, s), pad)
        out['conv/%s/dw' % name] = O.conv2d_bwd_filter(x, dy, w.shape, (s, s), pad)
        out['conv/%s/db' % name] = dy.reshape(-1, Co).sum(0)
    for name, B, H, W, Ci, Co in DECONV_CASES:
        x = rng.standard_normal((B, H, W, Ci)); w = rng.standard_normal((4, 4, Co, Ci)); b = rng.standard_normal(Co)
        out['deconv/%s/x' % name] = x; out['deconv/%s/w' % name] = w; out['deconv/%s/b' % name] = b
        out['deconv/%s/y' % name] = O.conv2d_transpose(x, w, b, (2, 2), 'SAME')
    # dense
    x = rng.standard_normal((5, 12)); k = rng.standard_normal((12, 7)); b = rng.standard_normal(7)
    out['dense/x'] = x; out['dense/k'] = k; out['dense/b'] = b; out['dense/y'] = O.dense(x, k, b)
    # batch norm rank-4 and rank-2
    for tag, shape in (('r4', (3, 4, 4, 8)), ('r2', (6, 8))):
        x = rng.standard_normal(shape) * 2 + 0.5; gamma = rng.standard_normal(8); beta = rng.standard_normal(8)
        dy = rng.standard_normal(shape)
        y, mean, var = O.batch_norm_train(x, gamma, beta)
        dx, dg, db = O.batch_norm_bwd(dy, x, gamma, mean, var)
        n = x.size // 8
        mm, mv = O.batch_norm_moving_update(np.zeros(8), np.ones(8), mean, var, n)
        for k_, v in dict(x=x, gamma=gamma, beta=beta, dy=dy, y=y, mean=mean, var=var, dx=dx, dgamma=dg, dbeta=db,
                          moving_mean=mm, moving_var=mv).items():
            out['bn/%s/%s' % (tag, k_)] = v
    # Adam KAT (beta1=0, t=1): dw = -lr*sqrt(.1)*g/(sqrt(.1)*|g|+1e-8)
    w = rng.standard_normal(16); g = rng.standard_normal(16) * 1e-3
    w1, m1, v1 = O.adam_tf(w, g, np.zeros(16), np.zeros(16), 1, 1e-4, 0.0, 0.9)
    w2, m2, v2 = O.adam_tf(w1, g * 0.5, m1, v1, 2, 1e-4, 0.0, 0.9)
    out.update({'adam/w': w, 'adam/g': g, 'adam/w1': w1, 'adam/m1': m1, 'adam/v1': v1, 'adam/w2': w2, 'adam/m2': m2,
                'adam/v2': v2})
    # gradient penalty on a raw gradient tensor
    g = rng.standard_normal((4, 4, 4, 3)) * 0.3
    g[0] *= 0.01  # one sample below the slope-1 hinge
    gp, slopes = O.gp_from_grad(g)
    out.update({'gp/g': g, 'gp/slopes': slopes, 'gp/value': np.array(gp), 'gp/dg': O.gp_from_grad_bwd(g, slopes)})
    return out


def make_step():
    cfg = T.Cfg(**TINY)
    P = T.init_variables(cfg, seed=0, dtype=torch.float64)
    for n in P:   # widen the critic so that both hinged penalties are active on the tiny model
        if n.startswith('d_net') and (n.endswith('weights') or n.endswith('kernel')):
            P[n] = P[n] * SCALE_D
    rng = np.random.default_rng(7)
    for n in P:   # non-trivial biases / BN affine so their gradients are exercised
        if n.endswith('biases') or n.endswith('bias') or n.endswith('beta'):
            P[n] = torch.tensor(rng.standard_normal(tuple(P[n].shape)) * 0.1)
        if n.endswith('gamma'):
            P[n] = torch.tensor(1.0 + rng.standard_normal(tuple(P[n].shape)) * 0.1)
    feed = T.synthetic_feed(cfg, seed=1, dtype=torch.float64)
    # inputs are rounded to float32-representable values so an fp32 implementation starts from identical bits
    for n in P:
        P[n] = P[n].float().double()
    for n in feed:
        feed[n] = feed[n].float().double()
    out = {}
    for n, v in P.items():
        out['param/' + n] = v.numpy().astype(np.float32)
    for n, v in feed.items():
        out['feed/' + n] = v.numpy().astype(np.float32)
    kt = 0.7
    d = T.d_step(P, cfg, feed, kt)
    for k_, v in d.items():
        if isinstance(v, float) and k_ != 'G_logits_absmax':      # (a test yardstick added after the fixture was cut)
            out['d/' + k_] = np.array(v)
    for n, v in d['grads'].items():
        out['d/grad/' + n] = v.numpy()
    out['d/G'] = d['G'].numpy(); out['d/Dx_hat'] = d['Dx_hat'].numpy()
    out['d/grad_x_hat'] = d['grad_x_hat'].numpy(); out['d/grad_cond'] = d['grad_cond'].numpy()
    g = T.g_step(P, cfg, feed)
    out['g/G_loss'] = np.array(g['G_loss']); out['g/G_kl_loss'] = np.array(g['G_kl_loss']); out['g/G'] = g['G'].numpy()
    for n, v in g['grads'].items():
        out['g/grad/' + n] = v.numpy()
    for n, (mean, var, cnt) in g['bn_stats'].items():
        out['g/bn_mean/' + n] = mean.numpy(); out['g/bn_var/' + n] = var.numpy()
    # one full trainer iteration (Adam + kt + moving stats) for the post-update state
    tr = T.Trainer(cfg, dict(P))
    tr.iteration(1, feed)
    for n, v in tr.P.items():
        out['after/' + n] = v.numpy().copy()
    out['after/kt'] = np.array(tr.kt)
    return out


GANCLS_TINY = dict(z_dim=12, embed_dim=32, compressed=16, gf=8, df=8, batch=4)


def make_gancls_step():
    """Tiny gancls iteration (reference models/gancls): losses, all gradients, post-update weights and BN moving stats."""
    from oracle import torch_gancls as GC
    cfg = GC.Cfg(**GANCLS_TINY)
    P = GC.init_variables(cfg, seed=0, dtype=torch.float64)
    rng = np.random.default_rng(11)
    for n in P:   # the reference's N(0,0.02) kernels make a tiny net almost linear: widen them so every branch is exercised
        if n.endswith('kernel'):
            P[n] = P[n] * (12.0 if 'conv' in n else 4.0)
        if n.endswith('bias') or n.endswith('beta'):
            P[n] = torch.tensor(rng.standard_normal(tuple(P[n].shape)) * 0.1)
    feed = GC.synthetic_feed(cfg, seed=1, dtype=torch.float64)
    for n in P:
        P[n] = P[n].float().double()
    for n in feed:
        feed[n] = feed[n].float().double()
    out = {}
    for n, v in P.items():
        out['param/' + n] = v.numpy().astype(np.float32)
    for n, v in feed.items():
        out['feed/' + n] = v.numpy().astype(np.float32)
    d = GC.d_step(P, cfg, feed)
    for k_ in ('D_loss', 'D_real_match_loss', 'D_real_mismatch_loss', 'D_synthetic_loss'):
        out['d/' + k_] = np.array(d[k_])
    for n, v in d['grads'].items():
        out['d/grad/' + n] = v.numpy()
    out['d/G'] = d['G'].numpy()
    g = GC.g_step(P, cfg, feed)
    out['g/G_loss'] = np.array(g['G_loss'])
    for n, v in g['grads'].items():
        out['g/grad/' + n] = v.numpy()
    tr = GC.Trainer(cfg, dict(P))
    tr.iteration(feed)
    for n, v in tr.P.items():
        out['after/' + n] = v.numpy().copy()
    return out


STACKGAN1_TINY = dict(z_dim=8, embed_dim=32, compressed=16, gf=8, df=8, batch=4)
STACKGAN2_TINY = dict(z_dim=8, embed_dim=16, compressed=8, gf=4, df=2, batch=2, out_size=256, real_label=0.95)
STACKGAN_EPOCH = 150          # lr = D_LR * 0.5 ** (150 // 100): the decay schedule is part of the fixture


def make_stackgan_step(stage):
    """Tiny StackGAN Stage-I / Stage-II iteration (reference models/stackgan/stage{I,II}): losses, all gradients and the
    post-update weights + BN moving statistics after one trainer iteration.  Stage II carries the Stage-I generator's
    variables too (it runs inside the Stage-II graph).  Image inputs are quantised to 256 levels so the file deflates."""
    from oracle import torch_stackgan as SG
    c1 = SG.Cfg(**STACKGAN1_TINY)
    cfg = c1 if stage == 1 else SG.Cfg(**STACKGAN2_TINY)
    cfg1 = None if stage == 1 else SG.Cfg(**dict(STACKGAN1_TINY, embed_dim=STACKGAN2_TINY['embed_dim'], batch=STACKGAN2_TINY['batch']))
    P = SG.init_variables(cfg, stage, cfg1, seed=0)
    rng = np.random.default_rng(21 + stage)
    for n in P:   # N(0,0.02) / He kernels at these widths make the nets almost linear: widen them so every branch is live
        if n.endswith('weights') or n.endswith('kernel'):
            he = n.startswith('stageII_g_net') and 'Conv2d_transpose' not in n and 'dense' not in n
            P[n] = P[n] * (1.5 if he else (12.0 if 'Conv' in n else 4.0))
        if n.endswith('biases') or n.endswith('bias') or n.endswith('beta'):
            P[n] = torch.tensor(rng.standard_normal(tuple(P[n].shape)) * 0.1)
    feed = SG.synthetic_feed(cfg, stage, cfg1, seed=1)
    for k in ('x', 'x_mismatch'):
        feed[k] = torch.round((feed[k] + 1.0) * 127.5) / 127.5 - 1.0
    for n in P:
        P[n] = P[n].float().double()
    for n in feed:
        feed[n] = feed[n].float().double()
    out = {}
    for n, v in P.items():
        out['param/' + n] = v.numpy().astype(np.float32)
    for n, v in feed.items():
        out['feed/' + n] = v.numpy().astype(np.float32)
    d = SG.d_step(P, cfg, feed, stage, cfg1)
    for k_ in ('D_loss', 'D_real_match_loss', 'D_real_mismatch_loss', 'D_synthetic_loss'):
        out['d/' + k_] = np.array(d[k_])
    for n, v in d['grads'].items():
        out['d/grad/' + n] = v.numpy().astype(np.float32)
    g = SG.g_step(P, cfg, feed, stage, cfg1)
    for k_ in ('G_loss', 'G_gan_loss', 'G_kl_loss'):
        out['g/' + k_] = np.array(g[k_])
    for n, v in g['grads'].items():
        out['g/grad/' + n] = v.numpy().astype(np.float32)
    gimg = g['G'].numpy()
    out['g/G_sample'] = gimg[:, ::4, ::4, :].astype(np.float32)          # a strided sample of the generated image
    tr = SG.Trainer(cfg, dict(P), stage, cfg1)
    tr.iteration(feed, epoch=STACKGAN_EPOCH)
    for n, v in tr.P.items():
        out['after/' + n] = v.numpy().astype(np.float32)
    return out


PGGAN_TINY = dict(z_dim=8, embed_dim=32, compressed=16, batch=3, base=32, cap=16)
PGGAN_STAGE, PGGAN_STEPS, PGGAN_IDX = 3, 10, 3          # 16x16 output, transition stage, alpha = 3/10


def make_pggan_step(trans=True):
    """Tiny PGGAN iteration (reference models/pggan/pggan.py) at stage 3: losses incl. both gradient penalties, all
    gradients (double backward through pool / fade-in / convs), post-update weights after one trainer iteration."""
    from oracle import torch_pggan as PG
    cfg = PG.Cfg(**PGGAN_TINY)
    P = PG.init_variables(cfg, PGGAN_STAGE, trans, seed=0)
    rng = np.random.default_rng(31)
    for n in P:   # widen the critic so that both hinged penalties are active; non-trivial biases / layer-norm affine
        if n.startswith('d_net') and (n.endswith('weights') or n.endswith('kernel')):
            P[n] = P[n] * 1.6
        if n.endswith('biases') or n.endswith('bias') or n.endswith('beta'):
            P[n] = torch.tensor(rng.standard_normal(tuple(P[n].shape)) * 0.1)
        if n.endswith('gamma'):
            P[n] = torch.tensor(1.0 + rng.standard_normal(tuple(P[n].shape)) * 0.1)
    feed = PG.synthetic_feed(cfg, PGGAN_STAGE, seed=1)
    for n in P:
        P[n] = P[n].float().double()
    for n in feed:
        feed[n] = feed[n].float().double()
    out = {}
    for n, v in P.items():
        out['param/' + n] = v.numpy().astype(np.float32)
    for n, v in feed.items():
        out['feed/' + n] = v.numpy().astype(np.float32)
    alpha = PGGAN_IDX / float(PGGAN_STEPS)
    d = PG.d_step(P, cfg, feed, PGGAN_STAGE, trans, alpha)
    for k_ in ('D_loss', 'wdist', 'wdist2', 'real_gp', 'real_gp2'):
        out['d/' + k_] = np.array(d[k_])
    out['d/G'] = d['G'].numpy(); out['d/Dx_hat'] = d['Dx_hat'].numpy()
    for n, v in d['grads'].items():
        out['d/grad/' + n] = v.numpy()
    g = PG.g_step(P, cfg, feed, PGGAN_STAGE, trans, alpha)
    out['g/G_loss'] = np.array(g['G_loss']); out['g/G_kl_loss'] = np.array(g['G_kl_loss']); out['g/G'] = g['G'].numpy()
    for n, v in g['grads'].items():
        out['g/grad/' + n] = v.numpy()
    tr = PG.Trainer(cfg, dict(P), PGGAN_STAGE, trans, PGGAN_STEPS)
    tr.iteration(PGGAN_IDX, feed)
    for n, v in tr.P.items():
        out['after/' + n] = v.numpy().copy()
    return out


if __name__ == '__main__':
    np.savez_compressed(os.path.join(HERE, 'pggan_tiny.npz'), **make_pggan_step())
    np.savez_compressed(os.path.join(HERE, 'stackgan1_tiny.npz'), **make_stackgan_step(1))
    np.savez_compressed(os.path.join(HERE, 'stackgan2_tiny.npz'), **make_stackgan_step(2))
    np.savez_compressed(os.path.join(HERE, 'gancls_tiny.npz'), **make_gancls_step())
    ops = make_ops()
    np.savez_compressed(os.path.join(HERE, 'ops_tiny.npz'), **ops)
    step = make_step()
    np.savez_compressed(os.path.join(HERE, 'step_tiny.npz'), **step)
    print('ops_tiny: %d arrays; step_tiny: %d arrays' % (len(ops), len(step)))
    print({k: float(v) for k, v in step.items() if k.startswith('d/') and v.ndim == 0})
