"""Data parallelism of the "next" rows on a real device: PGGAN and StackGAN Stage-I iterate under dp.DataParallel with an
RCCL communicator of world size 1 — eager (bucketed all-reduces launched from the backward as autograd.NOTIFY completes
them) and replayed from hipGraph segments cut at the exchange steps — and must end with exactly the bits of the plain
single-stream run: same kernels, same accumulation order, the exchange of one rank is the identity.  (Two ranks: the
gloo tests of tests/test_dp_models_gloo.py on CPU and tests/test_dp_exactness_gpu.py for wgancls.)"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_golden():
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    return mg


@pytest.fixture(scope='module')
def group():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import torch.distributed as dist
    import t2i_amd  # noqa: F401
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                            device_id=torch.device('cuda', torch.cuda.current_device()))
    yield dist
    dist.destroy_process_group()
    from t2i_amd import autograd as A
    A.NOTIFY[0] = None


def _same(a, b):
    assert a[1:] == b[1:]
    for n in a[0]:
        assert torch.equal(a[0][n], b[0][n]), n


def test_pggan_dp_eager_and_segments_match_plain(group):
    from t2i_amd import autograd as A
    from t2i_amd.dp import DataParallel
    from t2i_amd.models.pggan.pggan import PGGAN
    mg = _make_golden()
    gs = np.load(os.path.join(ROOT, 'tests', 'golden', 'pggan_tiny.npz'))
    t = mg.PGGAN_TINY
    dev = torch.device('cuda')
    f = {k[len('feed/'):]: torch.tensor(gs[k], dtype=torch.float32, device=dev) for k in gs.files if k.startswith('feed/')}
    g = torch.Generator(device=dev).manual_seed(4)
    feeds = [{'x': torch.rand(f['x'].shape, generator=g, device=dev) * 2 - 1, 'x_mismatch': f['x_mismatch'], 'cond': f['cond'],
              'z': torch.randn(f['z'].shape, generator=g, device=dev), 'eps_graph': torch.rand(f['eps'].shape, generator=g, device=dev),
              'ca_noise_d': torch.randn(f['ca_noise_d'].shape, generator=g, device=dev).clamp(-2, 2),
              'ca_noise_g': torch.randn(f['ca_noise_g'].shape, generator=g, device=dev).clamp(-2, 2)} for _ in range(4)]

    def run(dp, graphs):
        try:
            m = PGGAN(t['batch'], mg.PGGAN_STEPS, None, None, None, None, None, mg.PGGAN_STAGE, True, device=dev, fmap_base=t['base'],
                      fmap_max=t['cap'], z_dim=t['z_dim'], embed_dim=t['embed_dim'], compr_embed_dim=t['compressed'], dp=dp)
            m.store.load({k[len('param/'):]: gs[k] for k in gs.files if k.startswith('param/')})
            outs = []
            for i in range(4):
                if graphs and i == 1:
                    m.enable_graphs(feeds[0])
                outs.append(m.iteration(1 + 2 * i, feeds[i]))
            torch.cuda.synchronize()
            return ({n: v.detach().clone() for n, v in m.store.vars.items()}, float(outs[-1]['d']['D_loss']), float(outs[-1]['g']['G_loss']))
        finally:
            A.NOTIFY[0] = None

    plain = run(None, False)
    dp = DataParallel(bucket_bytes=4096)
    eager = run(dp, False)
    st = next(iter(dp._arenas.values()))
    assert st['expect'] and len(st['buckets']) > 1          # counts were learned; the overlap path was live from step 2 on
    segments = run(DataParallel(bucket_bytes=4096), True)
    _same(plain, eager)
    _same(plain, segments)


def test_stackgan1_dp_eager_and_segments_match_plain(group):
    from t2i_amd import autograd as A
    from t2i_amd.dp import DataParallel
    from t2i_amd.models.stackgan.stageI.model import ConditionalGan
    from t2i_amd.models.stackgan.stageI.trainer import ConditionalGanTrainer
    from t2i_amd.utils.config import AttrDict
    mg = _make_golden()
    gs = np.load(os.path.join(ROOT, 'tests', 'golden', 'stackgan1_tiny.npz'))
    t1 = mg.STACKGAN1_TINY
    dev = torch.device('cuda')
    cfg = AttrDict({'MODEL': {'Z_DIM': t1['z_dim'], 'OUTPUT_SIZE': 64, 'EMBED_DIM': t1['embed_dim'], 'COMPRESSED_EMBED_DIM': t1['compressed'],
                              'GF_DIM': t1['gf'], 'DF_DIM': t1['df'], 'IMAGE_SHAPE': {'W': 64, 'H': 64, 'D': 3}},
                    'TRAIN': {'BATCH_SIZE': t1['batch'], 'SAMPLE_NUM': 4, 'EPOCH': 1, 'D_LR': 2e-4, 'D_BETA_DECAY': 0.5, 'G_LR': 2e-4,
                              'G_BETA_DECAY': 0.5, 'COEFF': {'ALPHA_MISMATCH_LOSS': 0.5, 'KL': 2.0}}})
    f = {k[len('feed/'):]: torch.tensor(gs[k], dtype=torch.float32, device=dev) for k in gs.files if k.startswith('feed/')}
    g = torch.Generator(device=dev).manual_seed(8)
    feeds = []
    for _ in range(4):
        fd = {'inputs': torch.rand(f['x'].shape, generator=g, device=dev) * 2 - 1, 'wrong_inputs': f['x_mismatch'],
              'phi_inputs': torch.randn(f['cond'].shape, generator=g, device=dev), 'z': torch.randn(f['z'].shape, generator=g, device=dev)}
        for k in f:
            if k.startswith('ca_noise'):
                fd[k] = torch.randn(f[k].shape, generator=g, device=dev).clamp(-2, 2)
        feeds.append(fd)

    def run(dp, graphs):
        try:
            m = ConditionalGan(cfg, device=dev, dp=dp)
            m.store.load({k[len('param/'):]: gs[k] for k in gs.files if k.startswith('param/')})
            tr = ConditionalGanTrainer(None, m, None, cfg)
            outs = []
            for i in range(4):
                if graphs and i == 1:
                    tr.enable_graphs(feeds[0])
                outs.append(tr.iteration(feeds[i], epoch=100 * i))
            torch.cuda.synchronize()
            return ({n: v.detach().clone() for n, v in m.store.vars.items()}, float(outs[-1]['d']['D_loss']), float(outs[-1]['g']['G_loss']))
        finally:
            A.NOTIFY[0] = None

    plain = run(None, False)
    eager = run(DataParallel(bucket_bytes=4096), False)
    segments = run(DataParallel(bucket_bytes=4096), True)
    _same(plain, eager)
    _same(plain, segments)


def test_gancls_dp_eager_and_segments_match_plain(group):
    """gancls (the sigmoid-cross-entropy variant) under dp.DataParallel with a 1-rank RCCL communicator: eager with the
    bucketed exchange, and replayed from graph segments cut at the two exchange steps, leave the bits of the plain run."""
    from t2i_amd import autograd as A
    from t2i_amd.dp import DataParallel
    from t2i_amd.models.gancls.model import GanCls
    from t2i_amd.models.gancls.trainer import GanClsTrainer
    from t2i_amd.utils.config import AttrDict
    mg = _make_golden()
    gs = np.load(os.path.join(ROOT, 'tests', 'golden', 'gancls_tiny.npz'))
    t = mg.GANCLS_TINY
    dev = torch.device('cuda')
    cfg = AttrDict({'MODEL': {'Z_DIM': t['z_dim'], 'OUTPUT_SIZE': 64, 'EMBED_DIM': t['embed_dim'], 'COMPRESSED_EMBED_DIM': t['compressed'],
                              'GF_DIM': t['gf'], 'DF_DIM': t['df'], 'IMAGE_SHAPE': {'W': 64, 'H': 64, 'D': 3}},
                    'TRAIN': {'BATCH_SIZE': t['batch'], 'SAMPLE_NUM': 4, 'EPOCH': 1, 'D_LR': 2e-4, 'D_BETA_DECAY': 0.5, 'G_LR': 2e-4,
                              'G_BETA_DECAY': 0.5, 'COEFF': {'ALPHA_MISMATCH_LOSS': 0.5}}})
    f = {k[len('feed/'):]: torch.tensor(gs[k], dtype=torch.float32, device=dev) for k in gs.files if k.startswith('feed/')}
    g = torch.Generator(device=dev).manual_seed(12)
    feeds = [{'inputs': torch.rand(f['x'].shape, generator=g, device=dev) * 2 - 1, 'wrong_inputs': f['x_mismatch'],
              'phi_inputs': torch.randn(f['cond'].shape, generator=g, device=dev), 'z': torch.randn(f['z'].shape, generator=g, device=dev)}
             for _ in range(4)]

    def run(dp, graphs):
        try:
            m = GanCls(cfg, device=dev, dp=dp)
            m.store.load({k[len('param/'):]: gs[k] for k in gs.files if k.startswith('param/')})
            tr = GanClsTrainer(None, m, None, cfg)
            outs = []
            for i in range(4):
                if graphs and i == 1:
                    tr.enable_graphs(feeds[0])
                outs.append(tr.iteration(feeds[i]))
            torch.cuda.synchronize()
            return ({n: v.detach().clone() for n, v in m.store.vars.items()}, float(outs[-1]['d']['D_loss']), float(outs[-1]['g']['G_loss']))
        finally:
            A.NOTIFY[0] = None

    plain = run(None, False)
    eager = run(DataParallel(bucket_bytes=4096), False)
    segments = run(DataParallel(bucket_bytes=4096), True)
    plain_graphs = run(None, True)
    _same(plain, eager)
    _same(plain, segments)
    _same(plain, plain_graphs)
