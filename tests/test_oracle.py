"""CPU tests that pin the oracle: (a) NumPy float64 loops against hand-computed KATs and adjoint identities,
(b) the torch-CPU step against (a), finite differences and the committed golden fixtures."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import np_ops as O
from oracle import torch_step as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(0)


def test_same_padding_kats():
    # k4s2 even-in (1,1); k3s1 (1,1); k4s1 (1,2); k2s1 (0,1)
    assert O.same_pad(64, 4, 2) == (32, 1, 1)
    assert O.same_pad(4, 3, 1) == (4, 1, 1)
    assert O.same_pad(6, 4, 1) == (6, 1, 2)
    assert O.same_pad(5, 2, 1) == (5, 0, 1)
    assert O.same_pad(7, 4, 2) == (4, 1, 2)
    assert O.out_geometry(4, 4, 4, 4, 4, 4, 'valid') == (1, 1, 0, 0)


def test_k4s2_ramp_corners():
    # 8x8 ramp, all-ones 4x4 filter: corner (0,0) sees rows/cols 0..2 only (pad 1 before)
    x = np.arange(64, dtype=np.float64).reshape(1, 8, 8, 1)
    w = np.ones((4, 4, 1, 1))
    y = O.conv2d(x, w, None, (2, 2), 'SAME')
    assert y.shape == (1, 4, 4, 1)
    assert y[0, 0, 0, 0] == x[0, 0:3, 0:3, 0].sum()
    assert y[0, 3, 3, 0] == x[0, 5:8, 5:8, 0].sum()
    assert y[0, 1, 2, 0] == x[0, 1:5, 3:7, 0].sum()


@pytest.mark.parametrize('case', [(2, 5, 6, 3, 4, 4, 4, 2, 'SAME'), (1, 4, 4, 2, 3, 3, 3, 1, 'SAME'),
                                  (2, 4, 4, 3, 2, 4, 4, 4, 'VALID'), (1, 6, 6, 2, 2, 4, 4, 1, 'SAME'),
                                  (1, 5, 5, 2, 2, 2, 2, 1, 'SAME'), (2, 3, 3, 4, 5, 1, 1, 1, 'valid')])
def test_tap_loop_matches_scalar_loop(case):
    B, H, W, Ci, Co, KH, KW, s, pad = case
    x = rng.standard_normal((B, H, W, Ci)); w = rng.standard_normal((KH, KW, Ci, Co)); b = rng.standard_normal(Co)
    np.testing.assert_allclose(O.conv2d(x, w, b, (s, s), pad), O.conv2d_scalar(x, w, b, (s, s), pad), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('case', [(2, 8, 8, 3, 4, 4, 4, 2, 'SAME'), (1, 7, 5, 2, 3, 4, 4, 2, 'SAME'),
                                  (2, 4, 4, 3, 2, 3, 3, 1, 'SAME'), (1, 6, 6, 2, 2, 4, 4, 1, 'SAME')])
def test_adjoint_identities(case):
    """<conv(x), dy> == <x, bwd_data(dy)> == <w, bwd_filter(x, dy)>."""
    B, H, W, Ci, Co, KH, KW, s, pad = case
    x = rng.standard_normal((B, H, W, Ci)); w = rng.standard_normal((KH, KW, Ci, Co))
    y = O.conv2d(x, w, None, (s, s), pad)
    dy = rng.standard_normal(y.shape)
    lhs = (y * dy).sum()
    assert abs(lhs - (x * O.conv2d_bwd_data(dy, w, x.shape, (s, s), pad)).sum()) < 1e-9 * abs(lhs) + 1e-9
    assert abs(lhs - (w * O.conv2d_bwd_filter(x, dy, w.shape, (s, s), pad)).sum()) < 1e-9 * abs(lhs) + 1e-9


def test_torch_ops_match_numpy_loops():
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    for (B, H, W, Ci, Co, KH, KW, s, pad) in [(2, 8, 8, 3, 4, 4, 4, 2, 'SAME'), (2, 4, 4, 4, 4, 3, 3, 1, 'same'),
                                              (2, 4, 4, 4, 2, 1, 1, 1, 'valid'), (2, 4, 4, 4, 1, 4, 4, 4, 'VALID'),
                                              (1, 6, 6, 2, 2, 4, 4, 1, 'SAME'), (1, 5, 5, 2, 2, 2, 2, 1, 'SAME')]:
        x = rng.standard_normal((B, H, W, Ci)); w = rng.standard_normal((KH, KW, Ci, Co)); b = rng.standard_normal(Co)
        yt = T._conv(t(x).permute(0, 3, 1, 2), t(w), t(b), s, pad).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(yt, O.conv2d(x, w, b, (s, s), pad), rtol=1e-11, atol=1e-11)
    x = rng.standard_normal((2, 4, 4, 5)); w = rng.standard_normal((4, 4, 3, 5)); b = rng.standard_normal(3)
    yt = T._deconv_k4s2(t(x).permute(0, 3, 1, 2), t(w), t(b)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(yt, O.conv2d_transpose(x, w, b, (2, 2), 'SAME'), rtol=1e-11, atol=1e-11)


def test_torch_bn_matches_numpy():
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    x = rng.standard_normal((3, 4, 4, 6)) * 2 + 1; g = rng.standard_normal(6); b = rng.standard_normal(6)
    P = {'bn/gamma': t(g), 'bn/beta': t(b)}
    st = {}
    y = T._bn(P, 'bn', t(x).permute(0, 3, 1, 2), True, st).permute(0, 2, 3, 1).numpy()
    y0, mean, var = O.batch_norm_train(x, g, b)
    np.testing.assert_allclose(y, y0, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(st['bn'][0].numpy(), mean, rtol=1e-12)
    np.testing.assert_allclose(st['bn'][1].numpy(), var, rtol=1e-12)
    # backward of the numpy restatement against torch autograd
    xt = t(x).permute(0, 3, 1, 2).requires_grad_(True)
    P = {'bn/gamma': t(g).requires_grad_(True), 'bn/beta': t(b).requires_grad_(True)}
    dy = rng.standard_normal(x.shape)
    out = T._bn(P, 'bn', xt, True, None)
    gx, gg, gb = torch.autograd.grad((out * t(dy).permute(0, 3, 1, 2)).sum(), [xt, P['bn/gamma'], P['bn/beta']])
    dx, dgm, dbt = O.batch_norm_bwd(dy, x, g, mean, var)
    np.testing.assert_allclose(gx.permute(0, 2, 3, 1).numpy(), dx, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(gg.numpy(), dgm, rtol=1e-10); np.testing.assert_allclose(gb.numpy(), dbt, rtol=1e-10)


def test_adam_kat():
    # beta1=0, t=1  =>  dw = -lr*sqrt(0.1)*g / (sqrt(0.1)*|g| + 1e-8)   (SURVEY §8c KAT 4)
    g = np.array([1e-3, -2.0, 1e-9, 0.0]); w = np.zeros(4)
    w1, m, v = O.adam_tf(w, g, np.zeros(4), np.zeros(4), 1, 1e-4, 0.0, 0.9)
    expect = -1e-4 * np.sqrt(0.1) * g / (np.sqrt(0.1) * np.abs(g) + 1e-8)
    np.testing.assert_allclose(w1, expect, rtol=1e-12, atol=0)
    # torch AdamTF class agrees with the numpy restatement over two steps
    P = {'a': torch.tensor(w)}
    opt = T.AdamTF(['a'], P, 0.0, 0.9)
    opt.apply(P, {'a': torch.tensor(g)}, 1e-4); opt.apply(P, {'a': torch.tensor(g * 0.5)}, 1e-4)
    w2, _, _ = O.adam_tf(w1, g * 0.5, m, v, 2, 1e-4, 0.0, 0.9)
    np.testing.assert_allclose(P['a'].numpy(), w2, rtol=1e-12)


def test_gp_linear_critic_kat():
    # D(x) = <a, x>  =>  slopes == ||a|| for every sample (SURVEY §8c KAT 5)
    a = rng.standard_normal((4, 4, 3))
    g = np.broadcast_to(a, (5, 4, 4, 3))
    gp, slopes = O.gp_from_grad(g)
    np.testing.assert_allclose(slopes, np.linalg.norm(a.ravel()) * np.ones(5), rtol=1e-12)
    assert abs(gp - max(0.0, np.linalg.norm(a.ravel()) - 1.0) ** 2) < 1e-12
    # analytic backward vs finite differences
    g = rng.standard_normal((3, 2, 2, 3))
    _, s = O.gp_from_grad(g)
    dg = O.gp_from_grad_bwd(g, s)
    d = rng.standard_normal(g.shape); h = 1e-6
    fd = (O.gp_from_grad(g + h * d)[0] - O.gp_from_grad(g - h * d)[0]) / (2 * h)
    assert abs(fd - (dg * d).sum()) < 1e-6 * max(1.0, abs(fd))


def _tiny():
    cfg = T.Cfg(z_dim=8, embed_dim=32, compressed=16, gf=8, df=8, batch=4)
    return cfg


def test_variable_registry_counts():
    V = T.variable_shapes(T.Cfg())
    g = sum(int(np.prod(s)) for n, (s, k, f) in V.items() if n.startswith('g_net') and T.is_trainable(n))
    d = sum(int(np.prod(s)) for n, (s, k, f) in V.items() if n.startswith('d_net') and T.is_trainable(n))
    assert (g, d) == (22643287, 28995329)          # SURVEY §8a totals
    assert V['d_net/Conv_7/weights'][0] == (3, 3, 1152, 1024)
    assert V['g_net/Conv2d_transpose/weights'][0] == (4, 4, 512, 1024)
    assert V['g_net/dense_2/kernel'][0] == (256, 16384)


def test_d_step_double_backward_finite_difference(golden_step):
    """Directional derivative of D_loss along a random direction in d-weight space: autograd (incl. the
    gradient-penalty double backward) vs central differences, float64."""
    cfg = _tiny()
    P = {k[len('param/'):]: torch.tensor(golden_step[k], dtype=torch.float64) for k in golden_step.files if k.startswith('param/')}
    feed = {k[len('feed/'):]: torch.tensor(golden_step[k], dtype=torch.float64) for k in golden_step.files if k.startswith('feed/')}
    d = T.d_step(P, cfg, feed, 0.7)
    g = torch.Generator().manual_seed(3)
    dirs = {n: torch.randn(P[n].shape, generator=g, dtype=torch.float64) for n in d['grads']}
    norm = sum(float((v ** 2).sum()) for v in dirs.values()) ** 0.5
    dirs = {n: v / norm for n, v in dirs.items()}     # unit step direction: few lrelu/hinge kinks crossed
    analytic = sum(float((d['grads'][n] * dirs[n]).sum()) for n in dirs)
    h = 1e-6
    Pp = dict(P); Pm = dict(P)
    for n in dirs:
        Pp[n] = P[n] + h * dirs[n]; Pm[n] = P[n] - h * dirs[n]
    fd = (T.d_step(Pp, cfg, feed, 0.7)['D_loss'] - T.d_step(Pm, cfg, feed, 0.7)['D_loss']) / (2 * h)
    assert abs(fd - analytic) < 1e-6 * abs(analytic), (fd, analytic)


def test_golden_fixtures_regenerate(golden_ops, golden_step):
    """The committed fixtures are exactly what the oracle produces today (guards silent oracle drift)."""
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    ops = mg.make_ops()
    assert sorted(ops) == sorted(golden_ops.files)
    for k in ops:
        np.testing.assert_allclose(ops[k], golden_ops[k], rtol=1e-12, atol=1e-12, err_msg=k)
    step = mg.make_step()
    assert sorted(step) == sorted(golden_step.files)
    for k in step:
        np.testing.assert_allclose(step[k], golden_step[k], rtol=1e-9, atol=1e-10, err_msg=k)


def test_golden_step_is_nontrivial(golden_step):
    assert golden_step['d/real_gp'] > 0 and golden_step['d/real_gp2'] > 0
    assert abs(float(golden_step['after/kt']) - float(golden_step['d/kt_new'])) < 1e-15
    # Adam with beta1=0 at t=1 moves every weight by ~lr*sign(g) (SURVEY §7 hard parts)
    k = 'd_net/Conv_3/weights'
    delta = golden_step['after/' + k] - golden_step['param/' + k].astype(np.float64)
    assert np.all(np.abs(delta) <= 1.0001e-4)


def test_mask_tape_replay_is_identity_and_pins_branches():
    """MaskTape (used by the B=64 GPU parity test): replaying a run's own branches reproduces it exactly; replaying a
    branch set with one unit forced onto the other slope changes that run's gradients (the tape really decides)."""
    cfg = T.Cfg(z_dim=8, embed_dim=32, compressed=16, gf=8, df=8, batch=2)
    P = {n: v.double() for n, v in T.init_variables(cfg, seed=3).items()}
    feed = {k: v.double() for k, v in T.synthetic_feed(cfg, seed=4).items()}
    own = {k: T.MaskTape() for k in ('G', 'Dg', 'Dx', 'Dxmi', 'Dxh')}
    with torch.no_grad():
        G, _, _ = T.generator(P, cfg, feed['z'], feed['cond'], feed['ca_noise_d'], train=True, tape=own['G'])
        T.discriminator(P, cfg, G, feed['cond'], own['Dg']); T.discriminator(P, cfg, feed['x'], feed['cond'], own['Dx'])
        T.discriminator(P, cfg, feed['x_mismatch'], feed['cond'], own['Dxmi'])
        T.discriminator(P, cfg, feed['eps'] * G + (1.0 - feed['eps']) * feed['x'], feed['cond'], own['Dxh'])
    masks = {k: [m.clone() for m in t.record] for k, t in own.items()}
    assert [len(masks[k]) for k in ('G', 'Dg', 'Dxh')] == [10, 9, 9]
    free, pinned = T.d_step(P, cfg, feed, 0.7), T.d_step(P, cfg, feed, 0.7, masks=masks)
    assert abs(free['D_loss'] - pinned['D_loss']) <= 1e-12 * abs(free['D_loss'])
    for n in free['grads']:          # x * slope(mask) vs F.leaky_relu: the same function, evaluated in a slightly different order
        assert torch.allclose(free['grads'][n], pinned['grads'][n], rtol=1e-10, atol=1e-12), n
    masks['Dxh'][1][:] = ~masks['Dxh'][1]            # force a whole layer of the x_hat pass onto the other slope
    forced = T.d_step(P, cfg, feed, 0.7, masks=masks)
    assert forced['real_gp'] != free['real_gp']
    gfree = T.g_step(P, cfg, feed)
    own = {k: T.MaskTape() for k in ('G', 'Dg')}
    with torch.no_grad():
        G, _, _ = T.generator(P, cfg, feed['z'], feed['cond'], feed['ca_noise_g'], train=True, tape=own['G'])
        T.discriminator(P, cfg, G, feed['cond'], own['Dg'])
    gpin = T.g_step(P, cfg, feed, masks={k: t.record for k, t in own.items()})
    assert abs(gfree['G_loss'] - gpin['G_loss']) <= 1e-12 * abs(gfree['G_loss'])
    for n in gfree['grads']:
        assert torch.allclose(gfree['grads'][n], gpin['grads'][n], rtol=1e-10, atol=1e-12), n
