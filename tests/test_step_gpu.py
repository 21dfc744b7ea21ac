"""GPU parity of the whole wgancls iteration (model + autograd composition + optimizers) against the oracle.

  * tiny model (GF=DF=8, B=4): every loss scalar, every gradient of both steps, kt', first Adam step and BN moving
    statistics against the committed float64 golden step (tests/golden/step_tiny.npz);
  * full-width model at B=8: losses and gradients against the torch-CPU fp32 oracle run on the same seeded inputs.
Tolerances for the tiny golden step = SURVEY.md 8(c)'s: forward tensors 1e-5, loss scalars rel <= 1e-5, gradients
max|d|/max|ref| <= 1e-4 per tensor (measured after the batch-norm statistics were made stable: 5.5e-6 / 1.2e-6 / 5.5e-6);
post-Adam weights within 2*lr (Adam with beta1=0 is sign-like at t=1).  The metric's own size (B = 64) is
tests/test_step_b64_gpu.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(z, e, c, gf, df, B):
    from t2i_amd.utils.config import AttrDict
    return AttrDict({'MODEL': {'Z_DIM': z, 'OUTPUT_SIZE': 64, 'EMBED_DIM': e, 'COMPRESSED_EMBED_DIM': c, 'GF_DIM': gf,
                               'DF_DIM': df, 'IMAGE_SHAPE': {'W': 64, 'H': 64, 'D': 3}},
                     'TRAIN': {'BATCH_SIZE': B, 'SAMPLE_NUM': 4, 'D_LR': 1e-4, 'G_LR': 1e-4, 'BETA1': 0.0, 'BETA2': 0.9,
                               'N_CRITIC': 1, 'SUMMARY_PERIOD': 10, 'MAX_STEPS': 10, 'COEFF': {'KL': 1.0, 'LAMBDA': 100.0}}})


def relerr(got, ref, floor=1e-3):
    """max|d| / max(max|ref|, floor): the floor keeps exactly-zero reference gradients (e.g. the logit bias, whose
    +1/-1 contributions cancel) from turning fp32 rounding residue into an infinite relative error."""
    got = got.detach().double().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), floor))


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import t2i_amd  # noqa: F401
    return torch.device('cuda')


def _feed(gs, dev):
    f = {k[len('feed/'):]: torch.tensor(gs[k], dtype=torch.float32, device=dev) for k in gs.files if k.startswith('feed/')}
    f['epsilon'] = f.pop('eps')
    f['learning_rate_d'] = 1e-4
    f['learning_rate_g'] = 1e-4
    return f


def test_tiny_step_matches_golden(gpu, golden_step):
    from t2i_amd.models.wgancls.model import WGanCls
    gs = golden_step
    m = WGanCls(_cfg(8, 32, 16, 8, 8, 4), device=gpu)
    m.store.load({k[len('param/'):]: gs[k] for k in gs.files if k.startswith('param/')})
    feed = _feed(gs, gpu)
    d = m.d_losses(feed)
    torch.cuda.synchronize()
    assert relerr(d['G'], gs['d/G']) <= 1e-5
    assert relerr(d['Dx_hat_logit'], gs['d/Dx_hat']) <= 1e-5
    assert relerr(d['grad_x_hat'], gs['d/grad_x_hat']) <= 1e-4
    assert relerr(d['grad_cond'], gs['d/grad_cond']) <= 1e-4
    for k in ('D_loss', 'D_loss_real', 'D_loss_fake', 'D_loss_mismatch', 'wdist', 'wdist2', 'real_gp', 'real_gp2',
              'reg_loss', 'balance_loss', 'kt_grad'):
        ref = float(gs['d/' + k])
        assert abs(float(d[k]) - ref) <= 1e-5 * max(abs(ref), 1.0), (k, float(d[k]), ref)
    for n in m.d_vars:
        _check_grad(m.d_arena.grad_of(n), gs['d/grad/' + n], n)
    print('tiny wgancls: worst critic gradient error %.2e; losses %s' % (
        max(relerr(m.d_arena.grad_of(n), gs['d/grad/' + n], floor=1e-30) for n in m.d_vars if np.abs(gs['d/grad/' + n]).max() >= 1e-9),
        {k: '%.1e' % (abs(float(d[k]) - float(gs['d/' + k])) / max(abs(float(gs['d/' + k])), 1.0)) for k in ('D_loss', 'wdist', 'real_gp', 'real_gp2')}))
    g = m.g_losses(feed)
    assert abs(float(g['G_loss']) - float(gs['g/G_loss'])) <= 1e-5 * max(abs(float(gs['g/G_loss'])), 1.0)
    assert abs(float(g['G_kl_loss']) - float(gs['g/G_kl_loss'])) <= 1e-5 * max(abs(float(gs['g/G_kl_loss'])), 1.0)
    assert relerr(g['G'], gs['g/G']) <= 1e-5
    for n in m.g_vars:
        _check_grad(m.g_arena.grad_of(n), gs['g/grad/' + n], n)
    print('tiny wgancls: worst generator gradient error %.2e; G_loss %.1e G %.1e' % (
        max(relerr(m.g_arena.grad_of(n), gs['g/grad/' + n], floor=1e-30) for n in m.g_vars if np.abs(gs['g/grad/' + n]).max() >= 1e-9),
        abs(float(g['G_loss']) - float(gs['g/G_loss'])) / max(abs(float(gs['g/G_loss'])), 1.0), relerr(g['G'], gs['g/G'])))


def _check_grad(got, ref, name, tol=1e-4):
    """Per-tensor max-norm check.  Tensors whose exact gradient is zero (a bias in front of a batch norm; the logit
    bias, whose +1/-1 terms cancel) only carry fp32 rounding residue: bound it absolutely instead."""
    ref = np.asarray(ref, np.float64)
    if np.abs(ref).max() < 1e-9:
        assert float(got.abs().max()) <= 1e-4, (name, float(got.abs().max()))
    else:
        e = relerr(got, ref, floor=1e-30)
        assert e <= tol, (name, e)


def _check_grad_kinks(got, ref, name, e_cpu):
    """Gradient check for the full-width nets, robust to lrelu/relu MASK FLIPS.  The fp32 MFMA chain accumulates
    K <= 10368 products sequentially: forward activations are ~5e-6 (relative) from float64, so out of ~4e5 units per
    layer a handful of pre-activations within that distance of zero take the other slope (1 vs 0.2).  Each flip moves
    one channel's gradient by O(its own size) — a sparse, legitimate effect of fp32 arithmetic on a piecewise-linear net
    (torch-CPU fp32 sits ~1e-7 from float64 and flips ~30x fewer).  So: either the max-norm error is at the level of
    torch-CPU fp32's own error, or the relative L2 error is <= 1e-2 with no element off by more than 10% of the
    tensor's scale (a wrong kernel gives O(1) errors in both)."""
    ref = np.asarray(ref, np.float64)
    got = got.detach().double().cpu().numpy()
    scale = np.abs(ref).max()
    diff = np.abs(got - ref)
    if scale < 1e-9:            # exactly-zero gradient (logit bias; biases in front of a batch norm): fp32 residue only
        assert diff.max() <= 1e-4, (name, diff.max())
        return
    if diff.max() <= max(3 * e_cpu, 1e-3) * scale:
        return
    l2 = float(np.linalg.norm(diff) / max(np.linalg.norm(ref), 1e-30))
    assert l2 <= 1e-2 and diff.max() <= 0.1 * scale, (name, diff.max() / scale, l2)


def test_tiny_full_iteration_state(gpu, golden_step):
    """D step (+kt) then G step with Adam and BN moving averages: post-update state vs the oracle trainer."""
    from t2i_amd.models.wgancls.model import WGanCls
    from t2i_amd.models.wgancls.trainer import WGanClsTrainer
    gs = golden_step
    cfg = _cfg(8, 32, 16, 8, 8, 4)
    m = WGanCls(cfg, device=gpu)
    m.store.load({k[len('param/'):]: gs[k] for k in gs.files if k.startswith('param/')})
    tr = WGanClsTrainer(None, m, None, cfg)
    tr.iteration(1, _feed(gs, gpu))
    torch.cuda.synchronize()
    assert abs(float(m.kt) - float(gs['after/kt'])) <= 1e-5
    lr = 1e-4
    for n, v in m.store.vars.items():
        ref = gs['after/' + n]
        if n.endswith('moving_mean') or n.endswith('moving_variance'):
            assert relerr(v, ref) <= 1e-4, n
        else:
            # Adam(beta1=0) at t=1 moves each weight by ~lr*sign(g): near-zero gradients may flip sign in fp32
            delta = np.abs(v.detach().double().cpu().numpy() - ref)
            assert delta.max() <= 2.0 * lr * 1.001, (n, delta.max())
            gkey = ('d/grad/' if n.startswith('d_net') else 'g/grad/') + n
            if np.abs(gs[gkey]).max() > 1e-9:      # exact-zero gradients: fp32 residue decides the sign, skip
                assert np.mean(delta <= 0.02 * lr) >= 0.98, (n, np.mean(delta <= 0.02 * lr))


def test_double_backward_of_conv_chain(gpu):
    """grad-of-grad through conv -> lrelu -> conv (+ residual add) against torch's own CPU double backward."""
    import torch.nn.functional as F
    from t2i_amd import autograd as A, kernels as K
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 8, 8, 8, generator=g); w1 = torch.randn(4, 4, 8, 16, generator=g) * 0.2
    w2 = torch.randn(3, 3, 16, 16, generator=g) * 0.2; b1 = torch.randn(16, generator=g) * 0.1

    def ref():
        xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
        W1, W2, B1 = w1.double().requires_grad_(True), w2.double().requires_grad_(True), b1.double().requires_grad_(True)
        h = F.leaky_relu(F.conv2d(xr, W1.permute(3, 2, 0, 1), B1, stride=2, padding=1), 0.2)
        y = F.leaky_relu(h + F.conv2d(h, W2.permute(3, 2, 0, 1), None, padding=1), 0.2)
        gx, = torch.autograd.grad(y.sum(), [xr], create_graph=True)
        pen = ((gx.reshape(3, -1) ** 2).sum(1).sqrt() - 1).clamp(min=0).pow(2).mean()
        return (float(pen),) + torch.autograd.grad(pen, [W1, W2, B1])

    def ours():
        # the penalty does not depend on b1 except through lrelu masks: torch reports a zero gradient, our graph has
        # no edge at all (allow_unused) — both mean dpen/db1 == 0
        xd = x.cuda().requires_grad_(True)
        W1, W2, B1 = w1.cuda().requires_grad_(True), w2.cuda().requires_grad_(True), b1.cuda().requires_grad_(True)
        g1 = K.conv_desc(3, 8, 8, 8, 16, 4, 4, 2, 2, 'SAME'); g2 = K.conv_desc(3, 4, 4, 16, 16, 3, 3, 1, 1, 'SAME')
        h = A.Conv2dFn.apply(xd, W1, B1, g1, K.ACT_LRELU, 0.2)
        y = A.AddActFn.apply(h, A.Conv2dFn.apply(h, W2, None, g2, K.ACT_NONE, 0.0), K.ACT_LRELU, 0.2)
        with A.input_grads_only():
            gx, = torch.autograd.grad(y.sum(), [xd], create_graph=True)
        s = A.GpSlopesFn.apply(gx)
        pen = (s - 1).clamp(min=0).pow(2).mean()
        gs_ = torch.autograd.grad(pen, [W1, W2, B1], allow_unused=True)
        return (float(pen),) + tuple(g_ if g_ is not None else torch.zeros_like(p_) for g_, p_ in zip(gs_, [W1, W2, B1]))

    r, o = ref(), ours()
    assert abs(r[0] - o[0]) <= 1e-5 * max(abs(r[0]), 1.0)
    for a, b in zip(r[1:], o[1:]):
        assert relerr(b, a.numpy(), floor=1e-6) <= 1e-4


def test_full_width_step_vs_cpu_oracle(gpu):
    """The benchmark's architecture (GF=DF=128, 1024-d text) at B=8 against the torch-CPU oracle run in float64 on
    the same (float32-representable) weights and inputs."""
    from oracle import torch_step as T
    from t2i_amd.models.wgancls.model import WGanCls
    B = 8
    ocfg = T.Cfg(batch=B)
    P = {n: v.double() for n, v in T.init_variables(ocfg, seed=0).items()}
    feed = {k: v.double() for k, v in T.synthetic_feed(ocfg, seed=1).items()}
    m = WGanCls(_cfg(128, 1024, 128, 128, 128, B), device=gpu)
    m.store.load({n: v.numpy() for n, v in P.items()})
    f = {k: v.float().to(gpu) for k, v in feed.items()}
    f['epsilon'] = f.pop('eps'); f['learning_rate_d'] = 1e-4; f['learning_rate_g'] = 1e-4
    # the same step by the torch-CPU oracle in fp32: its distance from float64 is the yardstick for "fp32-exact".
    # (At random init the 150x gradient-penalty term makes dD_loss/dw a sum of large cancelling parts: even torch-CPU fp32
    # is only ~2e-3 from float64 on some tensors, so a fixed 1e-4 bound would test conditioning, not the kernels.)
    P32 = {n: v.float() for n, v in P.items()}
    feed32 = {k: v.float() for k, v in feed.items()}
    d = m.d_losses(f)
    ref, ref32 = T.d_step(P, ocfg, feed, 0.7), T.d_step(P32, ocfg, feed32, 0.7)
    for k in ('D_loss', 'wdist', 'wdist2', 'real_gp', 'real_gp2'):
        tol = max(3 * abs(ref32[k] - ref[k]), 1e-4 * max(abs(ref[k]), 1.0))
        assert abs(float(d[k]) - ref[k]) <= tol, (k, float(d[k]), ref[k], ref32[k])
    worst = (0.0, 0.0, '')
    for n in m.d_vars:
        e_cpu = relerr(ref32['grads'][n], ref['grads'][n].numpy())
        e = relerr(m.d_arena.grad_of(n), ref['grads'][n].numpy())
        worst = max(worst, (e, e_cpu, n))
        _check_grad_kinks(m.d_arena.grad_of(n), ref['grads'][n].numpy(), n, e_cpu)
    print('critic grads: worst max-norm error vs f64 %.2e (torch-CPU fp32: %.2e) at %s' % worst)
    g = m.g_losses(f)
    gref, gref32 = T.g_step(P, ocfg, feed), T.g_step(P32, ocfg, feed32)
    assert abs(float(g['G_loss']) - gref['G_loss']) <= max(3 * abs(gref32['G_loss'] - gref['G_loss']), 1e-4 * max(abs(gref['G_loss']), 1.0))
    worst = (0.0, 0.0, '')
    for n in m.g_vars:
        if float(gref['grads'][n].abs().max()) < 1e-9:      # bias in front of a batch norm: exact zero gradient
            assert float(m.g_arena.grad_of(n).abs().max()) <= 1e-4, n
            continue
        e_cpu = relerr(gref32['grads'][n], gref['grads'][n].numpy())
        e = relerr(m.g_arena.grad_of(n), gref['grads'][n].numpy())
        worst = max(worst, (e, e_cpu, n))
        _check_grad_kinks(m.g_arena.grad_of(n), gref['grads'][n].numpy(), n, e_cpu)
    print('generator grads: worst max-norm error vs f64 %.2e (torch-CPU fp32: %.2e) at %s' % worst)


@pytest.mark.parametrize('n_critic,noise_in_feed', [(1, True), (2, True), (1, False), (2, False)])
def test_graph_replay_matches_eager(gpu, golden_step, n_critic, noise_in_feed):
    """d_step/g_step captured into hipGraphs and replayed == the eager launches, bit for bit, over 3 iterations
    (same kernels, same order; only the launch mechanism differs)."""
    from t2i_amd.models.wgancls.model import WGanCls
    from t2i_amd.models.wgancls.trainer import WGanClsTrainer
    gs = golden_step
    cfg = _cfg(8, 32, 16, 8, 8, 4)
    cfg.TRAIN.N_CRITIC = n_critic        # 2: critic-only iterations in between (d_step's own graph) next to the merged D+G launch
    params = {k[len('param/'):]: gs[k] for k in gs.files if k.startswith('param/')}
    feeds = []
    g = torch.Generator(device=gpu).manual_seed(5)
    for _ in range(4):
        f = _feed(gs, gpu)
        f['x'] = torch.rand(f['x'].shape, generator=g, device=gpu) * 2 - 1
        f['z'] = torch.randn(f['z'].shape, generator=g, device=gpu)
        if not noise_in_feed:                # the model draws the conditioning-augmentation noise itself, as the reference does:
            del f['ca_noise_d'], f['ca_noise_g']   # replays must re-draw it, in the eager step's order
        feeds.append(f)
    states = []
    for use_graphs in (False, True):
        torch.manual_seed(77); torch.cuda.manual_seed_all(77)
        m = WGanCls(cfg, device=gpu)
        m.store.load(params)
        tr = WGanClsTrainer(None, m, None, cfg)
        tr.iteration(1, feeds[0])
        if use_graphs:
            m.enable_graphs(feeds[0])
        outs = [tr.iteration(2 + i, feeds[1 + i]) for i in range(3)]
        torch.cuda.synchronize()
        states.append(({n: v.detach().clone() for n, v in m.store.vars.items()}, float(m.kt), float(outs[-1]['d']['D_loss']),
                       float(outs[-1]['g']['G_loss'])))
    (s0, kt0, d0, g0), (s1, kt1, d1, g1) = states
    assert kt0 == kt1 and d0 == d1 and g0 == g1
    for n in s0:
        assert torch.equal(s0[n], s1[n]), n


@pytest.mark.parametrize('n_critic', [1, 2])
def test_side_stream_and_dp_single_rank_match_plain(gpu, golden_step, n_critic):
    """Three launch schedules of the same iteration give the same bits: (a) one stream; (b) sunk filter gradients on the
    second HIP stream (autograd.SIDE); (c) = (b) under dp.DataParallel with an RCCL communicator of world size 1, where
    the gradient buckets are all-reduced on the communication stream as autograd.NOTIFY completes them (first step learns
    the counts, later steps overlap); (d) data parallelism with the iteration replayed from hipGraph segments cut at the
    exchange steps.  Same kernels and accumulation order in all four."""
    import socket
    import torch.distributed as dist
    from t2i_amd import autograd as A
    from t2i_amd.dp import DataParallel
    from t2i_amd.models.wgancls.model import WGanCls
    from t2i_amd.models.wgancls.trainer import WGanClsTrainer
    gs = golden_step
    cfg = _cfg(8, 32, 16, 8, 8, 4)
    cfg.TRAIN.N_CRITIC = n_critic        # 2: critic-only iterations (the d_step segments) around one D+G iteration
    params = {k[len('param/'):]: gs[k] for k in gs.files if k.startswith('param/')}
    feeds = []
    g = torch.Generator(device=gpu).manual_seed(6)
    for _ in range(4):
        f = _feed(gs, gpu)
        f['x'] = torch.rand(f['x'].shape, generator=g, device=gpu) * 2 - 1
        f['z'] = torch.randn(f['z'].shape, generator=g, device=gpu)
        feeds.append(f)

    def run(side, dp, graphs=False):
        A.enable_side_stream(side)
        try:
            m = WGanCls(cfg, device=gpu, dp=dp)
            m.pair_g = False        # (the data-parallel schedules run the generator's two evaluations as two passes: compare like with like)
            m.store.load(params)
            tr = WGanClsTrainer(None, m, None, cfg)
            outs = []
            for i in range(4):
                if graphs and i == 1:                # [losses+backward] | eager all-reduce | [Adam] graph segments
                    m.enable_graphs(feeds[0])
                outs.append(tr.iteration(1 + i, feeds[i]))
            torch.cuda.synchronize()
            return ({n: v.detach().clone() for n, v in m.store.vars.items()}, float(m.kt), float(outs[-1]['d']['D_loss']),
                    float(outs[-1]['g']['G_loss']))
        finally:
            A.enable_side_stream(False)
            A.NOTIFY[0] = None

    plain = run(False, None)
    side = run(True, None)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                            device_id=torch.device('cuda', torch.cuda.current_device()))
    try:
        dp = DataParallel(bucket_bytes=4096)          # several buckets even on the tiny model
        both = run(True, dp)
        st = next(iter(dp._arenas.values()))
        assert st['expect'] and len(st['buckets']) > 1      # counts were learned; the overlap path was live on steps 2-3
        cut = run(False, DataParallel(bucket_bytes=4096), graphs=True)
        # (e) the fp32 buckets through the explicit reduce-scatter + all-gather form (dp.DataParallel(f32_exchange='rs_ag')): over RCCL
        # with one rank both collectives are identities, so this pins the in-place call form (output chunk = a view of the input) and
        # the bucket plumbing — the sums themselves are covered by tests/test_dp_gloo.py[rs_ag] and the two-rank run of
        # tests/test_dp_exactness_gpu.py
        rsag = run(True, DataParallel(bucket_bytes=4096, f32_exchange='rs_ag'))
    finally:
        dist.destroy_process_group()
    for other in (side, both, cut, rsag):
        assert plain[1:] == other[1:]
        for n in plain[0]:
            assert torch.equal(plain[0][n], other[0][n]), n


def test_all_bf16_math_tiny_step_unpinned_envelope(gpu, golden_step):
    """KERNEL COVERAGE, not config 3's parity claim.  The ALL-bf16 arithmetic (every GEMM of both networks with bf16 MFMA operands,
    fp32 accumulation, fp32 tensors and master weights; no per-network scope) on the tiny golden step against the UN-pinned float64
    oracle — config 3's own arithmetic (kernels.CONFIG3_NET_MATH) and its 2e-2 claim are tests/test_step_b64_gpu.py::
    test_config3_bf16_steps_mask_pinned, at full width, B = 64 / 16 / 8.  Bounds here are the measured envelope of this arithmetic
    on an 8-channel model (measured values in brackets), two of them above 2e-2 and stated as such:
      * forward tensors (G, D(x_hat)): relative L2 <= 2e-2 [1.2e-2, 9.7e-3] — 2^-8 per product, ~sqrt(12 layers);
      * Wasserstein scalars: <= 2e-2 of max(|ref|, 1) [<= 6.5e-3]; the penalty terms 100*(slope-1)^2 and hence D_loss
        amplify the slope error: <= 5e-2 [3.1e-2] (ABOVE config 3's 2e-2: an 8-channel critic averages nothing);
      * gradients: cosine to the oracle's >= 0.95 [0.984 critic, 0.971 generator].  A per-element bound is not
        meaningful un-pinned: a 1e-2 forward perturbation flips ~0.4% of the lrelu masks per layer and each flip changes that
        unit's gradient by 80%, i.e. ~17% relative L2 over 12 layers — the same model property that makes
        test_full_width_step_vs_cpu_oracle use kink-robust criteria at fp32.
    The arithmetic itself is pinned to 1e-5 by test_every_tile_and_split_config[bf16] (oracle on bf16-rounded operands)."""
    from t2i_amd import kernels as K
    from t2i_amd.models.wgancls.model import WGanCls
    gs = golden_step

    def rel_l2(got, ref):
        got = got.detach().double().cpu().numpy()
        ref = np.asarray(ref, np.float64)
        return float(np.linalg.norm(got - ref) / np.linalg.norm(ref))

    K.set_math('bf16')
    try:
        m = WGanCls(_cfg(8, 32, 16, 8, 8, 4), device=gpu)
        m.store.load({k[len('param/'):]: gs[k] for k in gs.files if k.startswith('param/')})
        feed = _feed(gs, gpu)
        d = m.d_losses(feed)
        torch.cuda.synchronize()
        assert 1e-4 < rel_l2(d['G'], gs['d/G']) <= 2e-2                     # reduced precision is really in use
        assert rel_l2(d['Dx_hat_logit'], gs['d/Dx_hat']) <= 2e-2
        for k, tol in (('D_loss_real', 2e-2), ('D_loss_fake', 2e-2), ('D_loss_mismatch', 2e-2), ('wdist', 2e-2),
                       ('wdist2', 2e-2), ('real_gp', 5e-2), ('real_gp2', 5e-2), ('D_loss', 5e-2)):
            ref = float(gs['d/' + k])
            assert abs(float(d[k]) - ref) <= tol * max(abs(ref), 1.0), (k, float(d[k]), ref)

        def cosine(arena, names, prefix):
            got = torch.cat([arena.grad_of(n).reshape(-1).double().cpu() for n in names])
            ref = torch.cat([torch.from_numpy(np.asarray(gs[prefix + n], np.float64)).reshape(-1) for n in names])
            return float((got * ref).sum() / (got.norm() * ref.norm()))
        assert cosine(m.d_arena, list(m.d_vars), 'd/grad/') >= 0.95
        g = m.g_losses(feed)
        assert abs(float(g['G_loss']) - float(gs['g/G_loss'])) <= 2e-2 * max(abs(float(gs['g/G_loss'])), 1.0)
        assert rel_l2(g['G'], gs['g/G']) <= 2e-2
        assert cosine(m.g_arena, list(m.g_vars), 'g/grad/') >= 0.95
    finally:
        K.set_math('f32')


def test_bf16_operand_images_and_batched_refresh_bit_identical(gpu):
    """bf16 math at the benchmark's widths: (a) handing the convs caller-held bf16 images of their activation operands
    (kernels.bf16_image + t2i_conv_opts.a_image / b_image: one cast per tensor instead of one per conv that reads it) and (b) the
    batched regeneration of the cached filter images behind the optimizer steps and at the head of every captured graph
    (t2i_filter_cache_refresh) and (c) the bf16 twins the producing kernels write next to their fp32 outputs (y_h arguments / t2i_conv_opts.out_image) change no bit of three training iterations — eager and replayed from a graph."""
    from t2i_amd import kernels as K
    from t2i_amd.models.wgancls.model import WGanCls
    from t2i_amd.models.wgancls.trainer import WGanClsTrainer
    B = 8
    cfg = _cfg(128, 1024, 128, 128, 128, B)
    g = torch.Generator(device=gpu).manual_seed(5)
    feed = {'x': torch.rand(B, 64, 64, 3, generator=g, device=gpu) * 2 - 1, 'x_mismatch': torch.rand(B, 64, 64, 3, generator=g, device=gpu) * 2 - 1,
            'cond': torch.randn(B, 1024, generator=g, device=gpu), 'z': torch.randn(B, 128, generator=g, device=gpu),
            'epsilon': torch.rand(B, 1, 1, 1, generator=g, device=gpu), 'learning_rate_d': 1e-4, 'learning_rate_g': 1e-4,
            'ca_noise_d': torch.randn(B, 128, generator=g, device=gpu).clamp_(-2, 2),
            'ca_noise_g': torch.randn(B, 128, generator=g, device=gpu).clamp_(-2, 2)}

    def run(images, refresh, graphs, twins=False):
        prev_i = K.bf16_images(images)
        prev_t = K.bf16_twins(twins)
        real_refresh = K.filter_cache_refresh
        if not refresh:
            K.filter_cache_refresh = lambda t=None: None        # every image is then filled lazily, at its first use
        prev_c = K.filter_cache(True)
        try:
            m = WGanCls(cfg, device=gpu, seed=3)
            tr = WGanClsTrainer(None, m, None, cfg)
            tr.iteration(1, feed)
            if graphs:
                m.enable_graphs(feed)
            tr.iteration(2, feed)
            tr.iteration(3, feed)
            torch.cuda.synchronize()
            return {n: v.detach().clone() for n, v in m.store.vars.items()}, float(m.kt)
        finally:
            K.bf16_images(prev_i)
            K.bf16_twins(prev_t)
            K.filter_cache_refresh = real_refresh
            K.filter_cache(prev_c)
    K.set_math('bf16')
    try:
        base = run(False, False, False)
        for variant in ((True, False, False), (False, True, False), (True, True, False), (True, True, True), (True, True, False, True),
                        (True, True, True, True)):
            got = run(*variant)
            assert got[1] == base[1], variant
            for n in base[0]:
                assert torch.equal(got[0][n], base[0][n]), (variant, n)
    finally:
        K.set_math('f32')


def test_store_first_gradient_slots_equal_zero_filled_ones(gpu):
    """optim.Arena(store_first): the large filter slots of the gradient arenas are not zero-filled — the first contribution of a step is
    written as a plain store (accumulate = 0 in the filter-gradient epilogue), later ones add.  0 + x == x in fp32, so both steps' arenas
    must equal the zero-filled ones bit for bit (full width, B = 8; the critic's first layer and text projection take two contributions,
    every other filter one).  And a slot that gets NO contribution is zeroed by finish_step() before the optimizer reads it."""
    from t2i_amd.models.wgancls.model import WGanCls
    B = 8
    cfg = _cfg(128, 1024, 128, 128, 128, B)
    g = torch.Generator(device=gpu).manual_seed(13)
    feed = {'x': torch.rand(B, 64, 64, 3, generator=g, device=gpu) * 2 - 1, 'x_mismatch': torch.rand(B, 64, 64, 3, generator=g, device=gpu) * 2 - 1,
            'cond': torch.randn(B, 1024, generator=g, device=gpu), 'z': torch.randn(B, 128, generator=g, device=gpu),
            'epsilon': torch.rand(B, 1, 1, 1, generator=g, device=gpu), 'learning_rate_d': 1e-4, 'learning_rate_g': 1e-4,
            'ca_noise_d': torch.randn(B, 128, generator=g, device=gpu).clamp_(-2, 2),
            'ca_noise_g': torch.randn(B, 128, generator=g, device=gpu).clamp_(-2, 2)}
    res = {}
    for store in (False, True):
        m = WGanCls(cfg, device=gpu, seed=3)
        m.d_arena.enable_sinks(store_first=store)
        m.g_arena.enable_sinks(store_first=store)
        assert (m.d_arena._store_first is not None) == store
        m.d_arena.grad.fill_(7.0); m.g_arena.grad.fill_(-3.0)          # stale values: the step must not depend on what the arenas held
        m.d_losses(feed)
        m.g_losses(feed)
        torch.cuda.synchronize()
        res[store] = (m.d_arena.grad.clone(), m.g_arena.grad.clone())
        if store:
            assert len(m.d_arena._store_first) >= 8 and m.d_arena._touched >= m.d_arena._store_first
            # a step that skips a filter: its slot is zeroed before the optimizer reads the arena
            m.d_arena.zero_grad()
            name = 'd_net/Conv_7/weights'
            assert float(m.d_arena.grad_of(name).abs().max()) > 0          # (not zero-filled: last step's gradient is still there)
            m.d_arena._touched = set(m.d_arena._store_first) - {name}
            m.d_arena.finish_step()
            assert float(m.d_arena.grad_of(name).abs().max()) == 0.0
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])


def test_paired_generator_iteration_matches_two_passes(gpu):
    """Round 6: a single-GPU D + G iteration evaluates the generator ONCE on 2B rows (WGanCls._g_forward_pair, stacked.py) — the
    generator step's evaluation (gradient, UPDATE_OPS) in front, the critic step's (no gradient, its own conditioning noise) behind,
    per-evaluation batch-norm statistics — instead of twice on B.  Against the two-pass iteration (pair_g = False) at the benchmark's
    width, B = 8, same weights and feed: both generated images, every logged scalar, the moving averages (they must have moved ONCE,
    by the generator step's statistics only) to fp32 rounding; the weight gradients by the kink-tolerant criterion of
    _check_grad_kinks (a 2B-row GEMM rounds differently from a B-row one: a handful of relu units at their kink take the other
    branch — see tests/test_step_b64_gpu.py — so a per-element 1e-6 is not owed; a wrong row, statistic or noise is an O(1) error)."""
    from t2i_amd.models.wgancls.model import WGanCls
    B = 8
    cfg = _cfg(128, 1024, 128, 128, 128, B)
    g = torch.Generator(device=gpu).manual_seed(11)
    feed = {'x': torch.rand(B, 64, 64, 3, generator=g, device=gpu) * 2 - 1, 'x_mismatch': torch.rand(B, 64, 64, 3, generator=g, device=gpu) * 2 - 1,
            'cond': torch.randn(B, 1024, generator=g, device=gpu), 'z': torch.randn(B, 128, generator=g, device=gpu),
            'epsilon': torch.rand(B, 1, 1, 1, generator=g, device=gpu), 'learning_rate_d': 1e-4, 'learning_rate_g': 1e-4,
            'ca_noise_d': torch.randn(B, 128, generator=g, device=gpu).clamp_(-2, 2),
            'ca_noise_g': torch.randn(B, 128, generator=g, device=gpu).clamp_(-2, 2)}
    res = {}
    for pair in (False, True):
        m = WGanCls(cfg, device=gpu, seed=3)
        m.pair_g = pair
        assert m._pairing(feed) == pair
        d, gout = m.dg_step(feed)
        torch.cuda.synchronize()
        res[pair] = dict(Gd=d['G'].clone(), Gg=gout['G'].clone(), d={k: float(d[k]) for k in ('D_loss', 'wdist', 'wdist2', 'real_gp', 'real_gp2')},
                         g={k: float(gout[k]) for k in ('G_loss', 'G_kl_loss', 'D_loss_fake')}, dgrad={n: m.d_arena.grad_of(n).clone() for n in m.d_vars},
                         ggrad={n: m.g_arena.grad_of(n).clone() for n in m.g_vars},
                         moving={n: v.detach().clone() for n, v in m.store.vars.items() if 'moving' in n})
    a, b = res[False], res[True]
    # tanh of logits of magnitude ~20 behind ten batch norms over 8 samples: the yardstick of tests/test_step_b64_gpu.py is 1e-5 x max|logits|
    assert float((a['Gd'] - b['Gd']).abs().max()) <= 2e-4 and float((a['Gg'] - b['Gg']).abs().max()) <= 2e-4
    assert float((a['Gd'] - a['Gg']).abs().max()) > 1e-3                                                            # (the two evaluations do differ: their noise)
    for grp in ('d', 'g'):
        for k, v in a[grp].items():
            assert abs(b[grp][k] - v) <= 2e-4 * max(abs(v), 1.0), (k, v, b[grp][k])
    for n, v in a['moving'].items():
        w = b['moving'][n]
        assert float((v - w).abs().max()) <= 1e-5 * max(float(v.abs().max()), 0.1), n      # (some batch means are zero to rounding: absolute floor)
    for key in ('ggrad', 'dgrad'):
        top = max(float(v.abs().max()) for v in a[key].values())
        for n, v in a[key].items():
            if float(v.abs().max()) < 1e-5 * top:        # a bias in front of a batch norm: its gradient is zero up to fp32 residue on both sides
                assert float(b[key][n].abs().max()) < 1e-4 * top, n
                continue
            _check_grad_kinks(b[key][n], v.double().cpu().numpy(), n, 1e-5)


def test_shared_winograd_input_transform_bit_identical(gpu):
    """fp32 at the benchmark's widths (Winograd live): the filter gradient reading the input transform its forward conv left
    behind (t2i_conv2d_input_transform) changes no bit of three iterations — eager, and replayed from the one-graph iteration."""
    from t2i_amd import kernels as K
    from t2i_amd.models.wgancls.model import WGanCls
    from t2i_amd.models.wgancls.trainer import WGanClsTrainer
    B = 8
    cfg = _cfg(128, 1024, 128, 128, 128, B)
    g = torch.Generator(device=gpu).manual_seed(9)
    feed = {'x': torch.rand(B, 64, 64, 3, generator=g, device=gpu) * 2 - 1, 'x_mismatch': torch.rand(B, 64, 64, 3, generator=g, device=gpu) * 2 - 1,
            'cond': torch.randn(B, 1024, generator=g, device=gpu), 'z': torch.randn(B, 128, generator=g, device=gpu),
            'epsilon': torch.rand(B, 1, 1, 1, generator=g, device=gpu), 'learning_rate_d': 1e-4, 'learning_rate_g': 1e-4,
            'ca_noise_d': torch.randn(B, 128, generator=g, device=gpu).clamp_(-2, 2),
            'ca_noise_g': torch.randn(B, 128, generator=g, device=gpu).clamp_(-2, 2)}
    used = []
    real = K.conv_bwd_filter

    def spy(x, dy, d, ws_bytes, out=None, xform=None, **kw):
        used.append(xform is not None)
        return real(x, dy, d, ws_bytes, out=out, xform=xform, **kw)

    def run(share, graphs):
        prev = K.share_xform(share)
        K.conv_bwd_filter = spy
        try:
            m = WGanCls(cfg, device=gpu, seed=3)
            tr = WGanClsTrainer(None, m, None, cfg)
            tr.iteration(1, feed)
            if graphs:
                m.enable_graphs(feed)
            tr.iteration(2, feed)
            tr.iteration(3, feed)
            torch.cuda.synchronize()
            return {n: v.detach().clone() for n, v in m.store.vars.items()}, float(m.kt)
        finally:
            K.share_xform(prev)
            K.conv_bwd_filter = real
    base = run(False, False)
    assert not any(used)
    for variant in ((True, False), (True, True)):
        del used[:]
        got = run(*variant)
        assert sum(used) >= 10, sum(used)            # the Winograd layers of both nets really took the shared transform
        assert got[1] == base[1], variant
        for n in base[0]:
            assert torch.equal(got[0][n], base[0][n]), (variant, n)


@pytest.mark.parametrize('math', ['f32', 'bf16'])
def test_filter_cache_full_width_bit_identical(gpu, math):
    """Transformed-filter cache (include/t2i_hip.h) at the benchmark's widths, where the Winograd paths (fp32) / the bf16 filter
    images are live: three iterations with the cache on — eager, and replayed from one hipGraph — leave exactly the weights, Adam
    state and kt of the run without it; the cache held buffers, and checkpoint-style loads behind the optimizer's back are honoured
    (round 4: the one-graph iteration no longer regenerates the critic's images at its head — t2i_filter_cache_assume — and
    dg_step regenerates them when kernels.filter_epoch() moved: the load below rewrites a critic and a generator filter)."""
    from t2i_amd import kernels as K
    from t2i_amd.models.wgancls.model import WGanCls
    from t2i_amd.models.wgancls.trainer import WGanClsTrainer
    B = 8
    cfg = _cfg(128, 1024, 128, 128, 128, B)
    g = torch.Generator(device=gpu).manual_seed(11)
    feeds = []
    for _ in range(4):
        feeds.append({'x': torch.rand(B, 64, 64, 3, generator=g, device=gpu) * 2 - 1,
                      'x_mismatch': torch.rand(B, 64, 64, 3, generator=g, device=gpu) * 2 - 1,
                      'cond': torch.randn(B, 1024, generator=g, device=gpu), 'z': torch.randn(B, 128, generator=g, device=gpu),
                      'epsilon': torch.rand(B, 1, 1, 1, generator=g, device=gpu), 'learning_rate_d': 1e-4, 'learning_rate_g': 1e-4,
                      'ca_noise_d': torch.randn(B, 128, generator=g, device=gpu).clamp_(-2, 2),
                      'ca_noise_g': torch.randn(B, 128, generator=g, device=gpu).clamp_(-2, 2)})

    def run(cache, graphs):
        prev = K.filter_cache(cache)
        try:
            m = WGanCls(cfg, device=gpu, seed=3)
            tr = WGanClsTrainer(None, m, None, cfg)
            tr.iteration(1, feeds[0])
            if graphs:
                m.enable_graphs(feeds[0])
            for i in range(2):
                tr.iteration(2 + i, feeds[1 + i])
            # a load behind the optimizer's back (ParamStore.load invalidates), then one more step
            loaded = {n: (v.detach() * 0.5).cpu().numpy() for n, v in m.store.vars.items() if n.endswith('Conv_6/weights')}
            assert sorted(loaded) == ['d_net/Conv_6/weights', 'g_net/Conv_6/weights']
            m.store.load(loaded)
            tr.iteration(4, feeds[3])
            tr.iteration(5, feeds[0])
            torch.cuda.synchronize()
            held = K.filter_cache_bytes()
            return ({n: v.detach().clone() for n, v in m.store.vars.items()}, m.D_optim.v.clone(), m.G_optim.m.clone(), float(m.kt)), held
        finally:
            K.filter_cache(prev)

    K.set_math(math)
    try:
        (ref, _), (eager, held), (replay, _) = run(False, False), run(True, False), run(True, True)
    finally:
        K.set_math('f32')
    assert held > 0                                         # the Winograd layers / bf16 filter images went through the cache
    for other in (eager, replay):
        assert other[3] == ref[3]
        assert torch.equal(other[1], ref[1]) and torch.equal(other[2], ref[2])
        for n in ref[0]:
            assert torch.equal(other[0][n], ref[0][n]), n


_ALGO_CHILD = r'''
import json, sys, torch
sys.path.insert(0, sys.argv[1])
import bench, t2i_amd
from t2i_amd import kernels as K
from t2i_amd.models.wgancls.model import WGanCls
dev = torch.device('cuda', 0)
cfg = bench.make_cfg(16)
m = WGanCls(cfg, device=dev, seed=5)
feed = bench.synthetic_feed(cfg, dev, seed=9)
d = m.d_losses(feed)
out = {k: float(d[k]) for k in ('D_loss', 'wdist', 'wdist2', 'real_gp', 'real_gp2')}
dg = m.d_arena.grad.clone()
g = m.g_losses(feed)
out['G_loss'] = float(g['G_loss'])
torch.save({'d': dg.cpu(), 'g': m.g_arena.grad.cpu()}, sys.argv[2])
descs = [K.conv_desc(48, 4, 4, 1152, 1024, 3, 3, 1, 1, 'SAME')[0], K.conv_desc(48, 16, 16, 256, 512, 4, 4, 2, 2, 'SAME')[0]]
out['algos'] = [K.conv_algo(x, 'fwd') for x in descs]
print(json.dumps(out))
'''


def test_winograd_and_direct_paths_agree_on_the_full_width_step(gpu, tmp_path):
    """The same full-width critic + generator step (B=16, benchmark architecture, random init) computed twice in child
    processes: with the library's default algorithm choice (Winograd on the many-channel 3x3 and 4x4-stride-2 layers) and
    with T2I_WINOGRAD=0 T2I_WINOGRAD_K4S2=0 (implicit GEMM everywhere).  Two fp32 evaluation orders of the same
    mathematics: losses agree to 1e-4 relative, the critic's gradient arena to 1e-2 and the generator's to 2e-3 of its
    norm.  (Yardstick: at random init the 150x gradient penalty makes dD_loss/dw a sum of large cancelling terms, and even
    torch-CPU fp32 sits ~2e-3 from float64 on some critic tensors — see test_full_width_step_vs_cpu_oracle, which holds
    the float64 comparison; measured here: 2.9e-3 for the critic arena.)"""
    import json
    import subprocess
    import sys
    res = {}
    for name, env in (('default', {}), ('direct', {'T2I_WINOGRAD': '0', 'T2I_WINOGRAD_K4S2': '0'})):
        e = dict(os.environ); e.update(env)
        dump = str(tmp_path / (name + '.pt'))
        r = subprocess.run([sys.executable, '-c', _ALGO_CHILD, ROOT, dump], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        line = [l for l in r.stdout.decode().splitlines() if l.startswith('{')][-1]
        res[name] = (json.loads(line), torch.load(dump))
    assert res['default'][0]['algos'] == ['winograd_f2x2_3x3', 'winograd_f2x2_2x2']
    assert res['direct'][0]['algos'] == ['implicit_gemm', 'implicit_gemm']
    # un-pinned: the two runs take a few lrelu / hinge branches differently (tests/test_step_b64_gpu.py pins them and gets 1e-8
    # on the same scalars with EITHER algorithm choice); 5e-4 is the size of that effect on the 150x gradient-penalty term
    for k in ('D_loss', 'wdist', 'wdist2', 'real_gp', 'real_gp2', 'G_loss'):
        a, b = res['default'][0][k], res['direct'][0][k]
        assert abs(a - b) <= 5e-4 * max(abs(b), 1.0), (k, a, b)
    for k in ('d', 'g'):
        a, b = res['default'][1][k].double(), res['direct'][1][k].double()
        assert float((a - b).norm() / b.norm()) <= (1e-2 if k == 'd' else 2e-3), k


def test_trainer_train_with_graphs_matches_eager(gpu):
    """WGanClsTrainer.train(graphs=True) on the synthetic dataset (fresh batches, z, epsilon and conditioning noise every
    iteration) ends with exactly the state of the eager loop: the capture happens after the first generator step, inputs are
    copied into the static buffers and the noise is re-drawn in the eager order."""
    from t2i_amd.data import SyntheticTextDataset
    from t2i_amd.models.wgancls.model import WGanCls
    from t2i_amd.models.wgancls.trainer import WGanClsTrainer
    cfg = _cfg(8, 32, 16, 8, 8, 4)
    cfg.TRAIN.N_CRITIC = 2
    states = []
    for graphs in (False, True):
        torch.manual_seed(5); torch.cuda.manual_seed_all(5)
        m = WGanCls(cfg, device=gpu, seed=1)
        ds = SyntheticTextDataset(cfg, device=gpu, seed=3, num_examples=64)
        tr = WGanClsTrainer(None, m, ds, cfg)
        tr.train(max_steps=8, log=lambda s: None, graphs=graphs)
        torch.cuda.synchronize()
        assert (m._graphs is not None) == graphs
        states.append(({n: v.detach().clone() for n, v in m.store.vars.items()}, float(m.kt)))
    assert states[0][1] == states[1][1]
    for n in states[0][0]:
        assert torch.equal(states[0][0][n], states[1][0][n]), n


def test_adam_first_moment_survives_a_stray_zero_grad(gpu):
    """beta1 == 0: the optimizer neither reads nor writes its first moment; `m` is formed from the gradient arena on demand
    (optim.AdamTF).  A zero_grad / backward between the step and the checkpoint must not change what the checkpoint holds
    (ADVICE round 4): an eager Arena.zero_grad forms the stale moment before it clears its source; assigning `m` sticks."""
    from t2i_amd.models.wgancls.model import WGanCls
    from t2i_amd.models.wgancls.trainer import WGanClsTrainer
    B = 4
    cfg = _cfg(8, 32, 16, 8, 8, B)
    g = torch.Generator(device=gpu).manual_seed(5)
    feed = {'x': torch.rand(B, 64, 64, 3, generator=g, device=gpu) * 2 - 1, 'x_mismatch': torch.rand(B, 64, 64, 3, generator=g, device=gpu) * 2 - 1,
            'cond': torch.randn(B, 32, generator=g, device=gpu), 'z': torch.randn(B, 8, generator=g, device=gpu),
            'epsilon': torch.rand(B, 1, 1, 1, generator=g, device=gpu), 'learning_rate_d': 1e-4, 'learning_rate_g': 1e-4}
    m = WGanCls(cfg, device=gpu, seed=3)
    assert m.D_optim.skip_m and m.G_optim.skip_m
    tr = WGanClsTrainer(None, m, None, cfg)
    tr.iteration(1, feed)
    tr.iteration(2, feed)
    want_d, want_g = m.d_arena.grad.clone(), m.g_arena.grad.clone()          # m_t = g_t (beta1 = 0, grad_scale = 1)
    assert float(want_d.abs().max()) > 0 and float(want_g.abs().max()) > 0
    m.d_arena.zero_grad()                                                    # stray: nobody has asked for m yet
    m.g_losses(feed)                                                         # a backward that refills the generator arena with other gradients
    assert not torch.equal(m.g_arena.grad, want_g)
    state = tr.make_saver().state()
    name = 'd_net/Conv_3/weights'
    o, k = m.d_arena.offsets[name]
    assert np.array_equal(state['D_optim/%s/Adam' % name].reshape(-1), want_d[o:o + k].cpu().numpy())
    assert torch.equal(m.D_optim.m, want_d) and torch.equal(m.G_optim.m, want_g)
    m.D_optim.m = torch.full_like(want_d, 0.25)                              # assignment sticks (no moments_loaded() needed)
    m.d_arena.zero_grad()
    assert float(m.D_optim.m.min()) == 0.25 == float(m.D_optim.m.max())
