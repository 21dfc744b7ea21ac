#!/usr/bin/env python
"""One measured row per model of SURVEY 8(f) / BASELINE configs 4-5 (and the gancls variant of 8(a)): images/s under hipGraph
replay, the ALGORITHMIC work per image (direct-convolution FLOPs of the conv / deconv / dense calls one iteration launches,
2 FLOP per multiply-add, padded taps included — the same count SURVEY 8(d) states for wgancls, taken here from the launched
descriptors of one eager iteration), and that work against the matrix peak on the replayed iteration's clock.

    python tools/next_rows.py [--math f32|bf16] [--rows gancls stage1 stage2 pggan7 wgancls_b8] [--budget-s 1.0]

bench.py imports `measure_rows` for its `next_rows` / `b8_per_gpu` blocks; run directly it prints one line per row
(profiles/r04_next_rows_throughput.txt).

reference: models/gancls/model.py:54-192, models/stackgan/stageI/model.py:76-171, models/stackgan/stageII/model.py:78-201,
models/pggan/pggan.py:251-316 (stage 7 = 256x256, batch 8: train_pggan.py:26-28), models/wgancls/cfg/flowers.yml:24 (BATCH_SIZE 8)."""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK = {'f32': 157.3, 'bf16': 2500.0}
PKG = os.path.join(ROOT, 'text-to-image_amd', 'models')


class _FlopCounter(object):
    """kernels.set_conv_timer target that only adds up the algorithmic FLOPs of the conv entry-point calls."""

    class _Ev(object):
        def record(self):
            pass

    def __init__(self):
        self.flop, self.calls = 0.0, 0

    def begin(self, flops, algo='implicit_gemm'):
        self.flop += flops
        self.calls += 1
        return self._Ev()


def _time_replays(step, budget_s, min_iters=3, max_iters=200):
    """step(): one replayed iteration.  Returns seconds per iteration over at least min_iters iterations / ~budget_s seconds."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-6)
    n = int(max(min_iters, min(max_iters, budget_s / one)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, n


def _row(name, workload, batch, math, dt, n, flop_per_iter, calls, extra=None):
    r = {'row': name, 'workload': workload, 'batch': batch, 'dtype': math, 'launch': 'hipGraph replay', 'iterations_timed': n,
         'ms_per_iteration': dt * 1e3, 'images_per_sec': batch / dt,
         'algorithmic_gflop_per_image': flop_per_iter / batch / 1e9, 'conv_calls_per_iteration': calls,
         'achieved_tflops': flop_per_iter / dt / 1e12, 'peak_tflops': PEAK[math],
         'frac_vs_driver_ms': flop_per_iter / dt / 1e12 / PEAK[math]}
    if extra:
        r.update(extra)
    return r


def _count_eager(K, run_eager):
    fc = _FlopCounter()
    K.set_conv_timer(fc)
    try:
        run_eager()
        torch.cuda.synchronize()
    finally:
        K.set_conv_timer(None)
    return fc.flop, fc.calls


def _gancls(K, dev, math, batch, budget_s):
    from t2i_amd.models.gancls.model import GanCls
    from t2i_amd.models.gancls.trainer import GanClsTrainer
    from t2i_amd.utils.config import config_from_yaml
    cfg = config_from_yaml(os.path.join(PKG, 'gancls', 'cfg', 'flowers.yml'))
    cfg.TRAIN.BATCH_SIZE = batch
    m = GanCls(cfg, device=dev)
    tr = GanClsTrainer(None, m, None, cfg)
    g = torch.Generator(device=dev).manual_seed(1)
    feed = {'inputs': torch.rand((batch, 64, 64, 3), generator=g, device=dev) * 2 - 1,
            'wrong_inputs': torch.rand((batch, 64, 64, 3), generator=g, device=dev) * 2 - 1,
            'phi_inputs': torch.randn((batch, cfg.MODEL.EMBED_DIM), generator=g, device=dev),
            'z': torch.randn((batch, cfg.MODEL.Z_DIM), generator=g, device=dev)}
    tr.iteration(feed)
    flop, calls = _count_eager(K, lambda: tr.iteration(feed))
    # the reference graph evaluates the generator in BOTH runs of an iteration (identical numbers: same feed, no noise, weights unchanged);
    # GanClsTrainer evaluates it once.  Both FLOP counts are reported: what is launched, and what the reference's two runs contain.
    shared, tr.share_g = tr.share_g, False
    ref_flop, ref_calls = _count_eager(K, lambda: tr.iteration(feed))
    tr.share_g = shared
    tr.enable_graphs(feed)
    tr.iteration(feed)
    dt, n = _time_replays(lambda: tr.iteration(feed), budget_s)
    tr._graphs = None
    extra = {'generator_evaluations_per_iteration': 1 if shared and m.dp is None else 2,
             'reference_graph_gflop_per_image': ref_flop / batch / 1e9, 'reference_graph_conv_calls': ref_calls,
             'frac_vs_reference_graph_flops': ref_flop / dt / 1e12 / PEAK[math],
             'note': 'algorithmic_gflop_per_image / frac_vs_driver_ms count the LAUNCHED convolutions (one generator forward per iteration); '
                     'reference_graph_* count the two identical generator forwards the reference\'s D run and G run evaluate'}
    if math == 'bf16':
        extra['arithmetic'] = {'mode': 'all_bf16', 'note': 'every GEMM of both networks in bf16 math: no tolerance is claimed and NO parity test runs gancls in this arithmetic '
                                                           '(its batch-normalised critic is the case DESIGN.md section 8 measures far outside 2e-2 on Stage-II): kernel throughput only'}
    return _row('gancls', 'gancls 64x64 (reference dims: z 100, GF 128, DF 64), D + G update, both under UPDATE_OPS', batch, math, dt, n, flop, calls, extra)


def _stackgan(K, dev, math, stage, batch, budget_s):
    from t2i_amd.models.stackgan.run import build
    from t2i_amd.utils.config import config_from_yaml
    cfg1 = config_from_yaml(os.path.join(PKG, 'stackgan', 'stageI', 'cfg', 'flowers.yml'))
    cfg = config_from_yaml(os.path.join(PKG, 'stackgan', 'stageI' if stage == 1 else 'stageII', 'cfg', 'flowers.yml'))
    cfg.TRAIN.BATCH_SIZE = cfg1.TRAIN.BATCH_SIZE = batch
    model, tr = build(stage, cfg, cfg1, device=dev)
    compliant = math == 'bf16' and stage == 2
    if compliant:
        # every forward GEMM of the critic, the Stage-II generator and the frozen Stage-I generator in fp32 math, every input- / filter-gradient GEMM in
        # bf16 math (kernels.FWD_F32_BWD_BF16): the arithmetic tests/test_fullsize_gpu.py::test_stackgan_stage2_fwd_f32_bwd_bf16_within_2e2 holds to 2e-2
        nm = {'g_net': K.FWD_F32_BWD_BF16, 'd_net': K.FWD_F32_BWD_BF16}
        model.net_math = dict(nm)
        model.stagei.net_math = dict(nm)
    feed = tr.make_feed()
    tr.iteration(feed)
    flop, calls = _count_eager(K, lambda: tr.iteration(feed))
    tr.enable_graphs(feed)
    tr.iteration(feed)
    dt, n = _time_replays(lambda: tr.iteration(feed), budget_s)
    tr._graphs = None
    what = ('StackGAN Stage-I 64x64' if stage == 1 else 'StackGAN Stage-II 256x256 (frozen Stage-I generator in training mode inside)')
    extra = None
    if compliant:
        extra = {'arithmetic': {'mode': 'fwd_f32_bwd_bf16', 'net_math': {k: list(v) for k, v in model.net_math.items()},
                                'note': 'every forward GEMM of the three networks in fp32 math, every input- and filter-gradient GEMM in bf16 math (bf16 operand '
                                        'images, fp32 accumulate), fp32 tensors',
                                'parity': 'tests/test_fullsize_gpu.py::test_stackgan_stage2_fwd_f32_bwd_bf16_within_2e2: every loss and every gradient tensor of both '
                                          'steps <= 2e-2 (relative L2, mask-pinned; measured <= 1.27e-2), the image exact'}}
    elif math == 'bf16':
        # every GEMM of every network in bf16 math: NOT a 2e-2 claim, and no parity test runs Stage-I in this arithmetic (the batch-normalised StackGAN step
        # at batch 2 sits 1.5e-1 .. 4.8e-1 from the oracle in it: DESIGN.md section 8): kernel throughput only
        extra = {'arithmetic': {'mode': 'all_bf16', 'note': 'every GEMM of both networks in bf16 math (bf16 MFMA operands, fp32 accumulate), %s activation '
                                'tensors: OUTSIDE 2e-2, kernel throughput only, no parity test in this arithmetic' % K.get_storage()}}
    return _row('stackgan_stage%d' % stage, what + ', D + G update', batch, math, dt, n, flop, calls, extra)


def _pggan(K, dev, math, stage, trans, budget_s):
    from t2i_amd.models.pggan.pggan import PGGAN
    from t2i_amd.models.pggan.train_pggan import dataset_for
    batch = 8 if stage >= 6 else 16
    size = 4 * 2 ** (stage - 1)
    p = PGGAN(batch_size=batch, steps=600000 // batch, check_dir_write=None, check_dir_read=None, dataset=dataset_for(size, dev),
              sample_path=None, log_dir=None, stage=stage, trans=trans, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    feed = p.make_feed(gen)
    p.iteration(1, feed)
    flop, calls = _count_eager(K, lambda: p.iteration(2, feed))
    p.enable_graphs(feed)
    p.iteration(3, feed)
    it = [4]

    def step():
        p.iteration(it[0], feed)
        it[0] += 1
    dt, n = _time_replays(step, budget_s)
    p._graphs = None
    return _row('pggan_stage%d%s' % (stage, 't' if trans else ''), 'PGGAN stage %d%s %dx%d (WGAN-GP, layer norm, pool / upscale), D + G update' % (
        stage, ' transition' if trans else '', size, size), batch, math, dt, n, flop, calls)


def _wgancls(K, dev, math, batch, budget_s):
    import bench
    from t2i_amd.models.wgancls.model import WGanCls
    from t2i_amd.models.wgancls.trainer import WGanClsTrainer
    cfg = bench.make_cfg(batch)
    m = WGanCls(cfg, device=dev, seed=0)
    arith = None
    if math == 'bf16':
        # BASELINE configs[2]'s COMPLIANT arithmetic (kernels.CONFIG3_NET_MATH: critic and the generator's backward GEMMs in bf16 math, the
        # generator's forward GEMMs in fp32 math) — the one tests/test_step_b64_gpu.py::test_config3_bf16_steps_mask_pinned[B8] holds to 2e-2
        m.net_math = dict(K.CONFIG3_NET_MATH)
        arith = {'mode': 'config3', 'net_math': {k: list(v) for k, v in K.CONFIG3_NET_MATH.items()},
                 'parity': 'tests/test_step_b64_gpu.py::test_config3_bf16_steps_mask_pinned[B%d]: <= 2e-2 on every tensor, mask-pinned, plus un-pinned forward values' % batch}
    tr = WGanClsTrainer(None, m, None, cfg)
    feed = bench.synthetic_feed(cfg, dev, seed=1, with_noise=False)
    tr.iteration(1, feed)
    flop, calls = _count_eager(K, lambda: tr.iteration(2, feed))
    m.enable_graphs(feed)
    feed.update({k: v for k, v in m.static_inputs().items() if feed.get(k) is not None})
    tr.iteration(3, feed)
    it = [4]

    def step():
        tr.iteration(it[0], feed)
        it[0] += 1
    dt, n = _time_replays(step, budget_s)
    m._graphs = None
    return _row('wgancls_b%d' % batch, 'wgancls 64x64 at batch %d per GPU = the strong-scaling share of global batch 64 on %d GPUs%s' % (
        batch, 64 // batch, ' (= the yml\'s BATCH_SIZE)' if batch == 8 else ''), batch, math, dt, n, flop, calls,
        {'arithmetic': arith} if arith else None)


ROWS = {
    'gancls': lambda K, dev, math, b: _gancls(K, dev, math, 64, b),
    'stage1': lambda K, dev, math, b: _stackgan(K, dev, math, 1, 64, b),
    'stage2': lambda K, dev, math, b: _stackgan(K, dev, math, 2, 32, b),
    'pggan7': lambda K, dev, math, b: _pggan(K, dev, math, 7, False, b),
    'wgancls_b8': lambda K, dev, math, b: _wgancls(K, dev, math, 8, b),
}


def measure_rows(names, math='f32', budget_s=1.0, device=None, storage='bf16'):
    """Measure the named rows one after the other in arithmetic `math`; everything a row creates is released before the next.
    A row that fails reports the exception instead of a number (the headline line must never die on a side block)."""
    import t2i_amd  # noqa: F401
    from t2i_amd import kernels as K
    dev = device if device is not None else torch.device('cuda')
    out = []
    for name in names:
        K.set_storage('f32')
        K.set_math(math)
        if math == 'bf16' and storage == 'bf16':
            K.set_storage('bf16')
        t0 = time.time()
        side = os.environ.get('T2I_SIDE_STREAM') == '1'        # diagnostics: sunk filter gradients on a second stream (autograd.enable_side_stream)
        if side:
            from t2i_amd import autograd as A
            A.enable_side_stream(True)
        try:
            r = ROWS[name](K, dev, math, budget_s)
            r['storage'] = K.get_storage()
        except Exception as e:          # noqa: BLE001
            r = {'row': name, 'dtype': math, 'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        r['wall_s'] = time.time() - t0
        if side:
            A.enable_side_stream(False)
            r['side_stream'] = True
        out.append(r)
        gc.collect()
        torch.cuda.synchronize()
        K.filter_cache_reset()
        K.set_storage('f32')
        K.set_math('f32')
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--math', choices=['f32', 'bf16'], default='f32')
    ap.add_argument('--rows', nargs='*', default=['gancls', 'stage1', 'stage2', 'pggan7', 'wgancls_b8'])
    ap.add_argument('--budget-s', type=float, default=1.5)
    ap.add_argument('--json', action='store_true')
    a = ap.parse_args()
    import t2i_amd  # noqa: F401
    from t2i_amd import kernels as K
    K.filter_cache(True)
    rows = measure_rows(a.rows, a.math, a.budget_s)
    for r in rows:
        if a.json:
            print(json.dumps(r))
        elif 'error' in r:
            print('%-16s %s  FAILED: %s' % (r['row'], r['dtype'], r['error']))
        else:
            print('%-16s %-4s B=%-3d %8.2f ms/iteration %9.1f img/s | %7.2f GFLOP/img algorithmic (%d conv calls) | %7.1f TFLOP/s = %.3f of the %s matrix peak '
                  '(frac_vs_driver_ms) | %s' % (r['row'], r['dtype'], r['batch'], r['ms_per_iteration'], r['images_per_sec'], r['algorithmic_gflop_per_image'],
                                                r['conv_calls_per_iteration'], r['achieved_tflops'], r['frac_vs_driver_ms'], r['dtype'], r['workload']))
            if r.get('arithmetic'):
                ar = r['arithmetic']
                print('%-16s      arithmetic: %s — %s' % ('', ar.get('mode'), ar.get('parity') or ar.get('note')))
            if 'frac_vs_reference_graph_flops' in r:
                print('%-16s      (the reference graph\'s %.2f GFLOP/img incl. its second, identical generator forward: %.3f of the peak)' % (
                    '', r['reference_graph_gflop_per_image'], r['frac_vs_reference_graph_flops']))


if __name__ == '__main__':
    main()
