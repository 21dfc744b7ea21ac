"""world_size-2 gloo runs (CPU) of the REAL models' data-parallel iteration: wgancls, StackGAN Stage-I and PGGAN.

There is no CPU arithmetic in this package, so the iteration runs in `kernels.dry_run()`: every launch is skipped and every
kernel output is zero-filled (torch.empty is replaced by torch.zeros in the worker), while everything the exchange depends
on is real — the autograd graph of the model, the gradient sinks and their NOTIFY announcements, the learned per-parameter
contribution counts, bucket completion, the all-reduce of the two gradient arenas (and of the kt means for wgancls) over a
real process group, and the hand-over to Adam.  `Arena.zero_grad` is replaced by "fill with rank + 1": since no kernel
writes gradients in a dry run, every element of both arenas must read 1 + 2 = 3 after an iteration — a bucket that was
never exchanged reads rank + 1, one exchanged twice reads 6, one exchanged before a later zero_grad reads rank + 1."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _build(which, dp):
    """-> (iterate(i), [arenas], device-free feed)"""
    from t2i_amd.utils.config import AttrDict
    B = 2
    if which in ('wgancls', 'wgancls_cut'):
        from t2i_amd.models.wgancls.model import WGanCls
        from t2i_amd.models.wgancls.trainer import WGanClsTrainer
        cfg = AttrDict({'MODEL': {'Z_DIM': 8, 'OUTPUT_SIZE': 64, 'EMBED_DIM': 32, 'COMPRESSED_EMBED_DIM': 16, 'GF_DIM': 8, 'DF_DIM': 8,
                                  'IMAGE_SHAPE': {'W': 64, 'H': 64, 'D': 3}},
                        'TRAIN': {'BATCH_SIZE': B, 'SAMPLE_NUM': 4, 'D_LR': 1e-4, 'G_LR': 1e-4, 'BETA1': 0.0, 'BETA2': 0.9, 'N_CRITIC': 1,
                                  'SUMMARY_PERIOD': 10, 'MAX_STEPS': 10, 'COEFF': {'KL': 1.0, 'LAMBDA': 100.0}}})
        m = WGanCls(cfg, device='cpu', dp=dp)
        m.dp_cut_eager = which == 'wgancls_cut'      # the segment sequence of the graph schedule: each backward cut once
        tr = WGanClsTrainer(None, m, None, cfg)
        feed = {'x': torch.zeros(B, 64, 64, 3), 'x_mismatch': torch.zeros(B, 64, 64, 3), 'cond': torch.zeros(B, 32), 'z': torch.zeros(B, 8),
                'epsilon': torch.zeros(B, 1, 1, 1), 'ca_noise_d': torch.zeros(B, 16), 'ca_noise_g': torch.zeros(B, 16),
                'learning_rate_d': 1e-4, 'learning_rate_g': 1e-4}
        return (lambda i: tr.iteration(i, feed)), [m.d_arena, m.g_arena], m
    if which == 'stackgan1':
        from t2i_amd.models.stackgan.stageI.model import ConditionalGan
        from t2i_amd.models.stackgan.stageI.trainer import ConditionalGanTrainer
        cfg = AttrDict({'MODEL': {'Z_DIM': 8, 'OUTPUT_SIZE': 64, 'EMBED_DIM': 32, 'COMPRESSED_EMBED_DIM': 16, 'GF_DIM': 8, 'DF_DIM': 8,
                                  'IMAGE_SHAPE': {'W': 64, 'H': 64, 'D': 3}},
                        'TRAIN': {'BATCH_SIZE': B, 'SAMPLE_NUM': 4, 'EPOCH': 1, 'D_LR': 2e-4, 'D_BETA_DECAY': 0.5, 'G_LR': 2e-4,
                                  'G_BETA_DECAY': 0.5, 'COEFF': {'ALPHA_MISMATCH_LOSS': 0.5, 'KL': 2.0}}})
        m = ConditionalGan(cfg, device='cpu', dp=dp)
        tr = ConditionalGanTrainer(None, m, None, cfg)
        feed = {'inputs': torch.zeros(B, 64, 64, 3), 'wrong_inputs': torch.zeros(B, 64, 64, 3), 'phi_inputs': torch.zeros(B, 32),
                'z': torch.zeros(B, 8), 'ca_noise_d': torch.zeros(B, 16), 'ca_noise_g': torch.zeros(B, 16)}
        return (lambda i: tr.iteration(feed)), [m.d_arena, m.g_arena], m
    from t2i_amd.models.pggan.pggan import PGGAN
    m = PGGAN(B, 100, None, None, None, None, None, stage=3, trans=True, device='cpu', fmap_base=32, fmap_max=16, z_dim=8, embed_dim=32,
              compr_embed_dim=16, dp=dp)
    feed = {'x': torch.zeros(B, 16, 16, 3), 'x_mismatch': torch.zeros(B, 16, 16, 3), 'cond': torch.zeros(B, 32), 'z': torch.zeros(B, 8),
            'eps_graph': torch.zeros(B), 'ca_noise_d': torch.zeros(B, 16), 'ca_noise_g': torch.zeros(B, 16)}
    return (lambda i: m.iteration(i, feed)), [m.d_arena, m.g_arena], m


def _worker(rank, world, port, which, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
        import t2i_amd  # noqa: F401
        from t2i_amd import kernels as K, optim
        from t2i_amd.dp import DataParallel
        torch.empty, torch.empty_like = torch.zeros, torch.zeros_like          # dry-run outputs: deterministic zeros
        optim.Arena.zero_grad = lambda self: self.grad.fill_(float(rank + 1))
        dp = DataParallel(bucket_bytes=2048)                                   # several buckets per arena even at these widths
        launches = []
        real_launch = dp._launch_range

        def spy(st, start, end):
            launches.append((id(st['arena']), start, end, sum(st['seen'].values())))
            return real_launch(st, start, end)
        dp._launch_range = spy
        with K.dry_run():
            step, arenas, model = _build(which, dp)
            total = sum(range(1, world + 1))
            for i in range(1, 4):
                del launches[:]
                step(i)
                for a in arenas:
                    assert bool((a.grad == float(total)).all()), (which, i, a.grad.unique().tolist()[:8])
                per_arena = {}
                for aid, s, e, seen in launches:
                    per_arena.setdefault(aid, []).append((s, e, seen))
                for a in arenas:
                    got = sorted((s, e) for s, e, _ in per_arena[id(a)])
                    assert got[0][0] == 0 and got[-1][1] == a.numel and all(x[1] == y[0] for x, y in zip(got, got[1:])), (which, i, got)
                    st = dp._arenas[id(a)]
                    if which == 'wgancls_cut':
                        # two exchanges per arena, the FIRST being the tail of the arena (the layers whose gradients are final
                        # after the first part of the cut backward: Conv_3.. of the critic, Conv2d_transpose.. of the generator)
                        first_var = model._CUT_D if a is model.d_arena else model._CUT_G
                        cut = a.offsets[first_var][0]
                        order = [(s_, e_) for s_, e_, _ in per_arena[id(a)]]
                        assert order == [(cut, a.numel), (0, cut)], (which, i, order, cut)
                        continue
                    assert st['expect'] and len(st['buckets']) > 1
                    if i > 1:        # counts learned on step 1: from then on the first bucket leaves before the backward has ended
                        seen = [n for _, _, n in per_arena[id(a)]]
                        assert min(seen) < sum(st['expect'].values()), (which, i, seen)
            if which in ('wgancls', 'wgancls_cut'):   # the kt means travelled as `extra`: both ranks contributed (zeros in a dry run) and kt is finite
                assert torch.isfinite(model.kt).all()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception as e:      # surface the failure in the parent
        import traceback
        q.put((rank, 'FAILED: %s\n%s' % (e, traceback.format_exc())))


@pytest.mark.parametrize('which,world', [('wgancls', 2), ('wgancls_cut', 2), ('stackgan1', 2), ('pggan', 2), ('wgancls', 3), ('wgancls_cut', 3)])
def test_model_iteration_exchanges_every_gradient_once(which, world):
    """(world 3, round 6: an odd number of ranks through both wgancls schedules — every check had run with two)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, which, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == 'ok' for r in results), results
