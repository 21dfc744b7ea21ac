#!/usr/bin/env python
"""Throughput and conv-roofline row of the gancls variant (reference models/gancls/model.py:54-192, trainer.py:19-51) at the
reference's own dimensions (cfg/flowers.yml: z 100, GF 128, DF 64), synthetic inputs, one MI355X.

    python tools/bench_gancls.py [--batch 64] [--steps 20] [--math f32|bf16]

Prints one line per configuration: images/s under hipGraph replay (one D + one G update per iteration, both under UPDATE_OPS,
i.e. 4 generator-side and 11 critic-side conv-class passes: SURVEY 8d "gancls iteration = 4 G + 11 D"), and the conv roofline
measured like bench.py's (HIP events around every conv entry point of 3 eager iterations: algorithmic direct-convolution FLOPs /
time, against the fp32 matrix peak of 157.3 TFLOP/s resp. the 2.5 PFLOP/s bf16 peak)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, nargs='*', default=[64, 8])
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--math', choices=['f32', 'bf16'], default='f32')
    args = ap.parse_args()
    import bench
    import t2i_amd  # noqa: F401
    from t2i_amd import kernels as K
    from t2i_amd.models.gancls.model import GanCls
    from t2i_amd.models.gancls.trainer import GanClsTrainer
    from t2i_amd.utils.config import config_from_yaml
    K.set_math(args.math)
    K.filter_cache(True)
    dev = torch.device('cuda')
    peak = bench.FP32_MATRIX_PEAK_TFLOPS if args.math == 'f32' else bench.BF16_MATRIX_PEAK_TFLOPS
    for B in args.batch:
        cfg = config_from_yaml(os.path.join(ROOT, 'text-to-image_amd', 'models', 'gancls', 'cfg', 'flowers.yml'))
        cfg.TRAIN.BATCH_SIZE = B
        m = GanCls(cfg, device=dev)
        tr = GanClsTrainer(None, m, None, cfg)
        g = torch.Generator(device=dev).manual_seed(1)
        feed = {'inputs': torch.rand((B, 64, 64, 3), generator=g, device=dev) * 2 - 1,
                'wrong_inputs': torch.rand((B, 64, 64, 3), generator=g, device=dev) * 2 - 1,
                'phi_inputs': torch.randn((B, cfg.MODEL.EMBED_DIM), generator=g, device=dev),
                'z': torch.randn((B, cfg.MODEL.Z_DIM), generator=g, device=dev)}
        for i in range(2):
            tr.iteration(feed)
        tr.enable_graphs(feed)
        for i in range(3):
            tr.iteration(feed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = tr.iteration(feed)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        # conv roofline: eager launches with per-launch events (as bench.py --instrument after)
        saved, tr._graphs = tr._graphs, None
        timer = bench.ConvTimer()
        K.set_conv_timer(timer)
        for i in range(3):
            torch.cuda._sleep(50000000)
            tr.iteration(feed)
        torch.cuda.synchronize()
        K.set_conv_timer(None)
        tr._graphs = saved
        s = timer.summary()
        tf = s['flop'] / (s['ms'] * 1e-3) / 1e12
        by = ', '.join('%s %d calls %.2f ms %.0f TF/s' % (a, r[0] // 3, r[1] / 3, r[2] / (r[1] * 1e-3) / 1e12) for a, r in sorted(s['by_algo'].items()))
        print('gancls  batch %3d  %s  graphs  %7.3f ms/iteration  %8.1f images/s  | conv entry points %.2f ms/iteration, %.1f GFLOP algorithmic, '
              '%.1f TFLOP/s = %.3f of the %s matrix peak (whole iteration: %.3f) | %s | d_loss %.4f g_loss %.4f' % (
                  B, args.math, dt * 1e3, B / dt, s['ms'] / 3, s['flop'] / 3 / 1e9, tf, tf / peak, args.math, s['flop'] / 3 / dt / 1e12 / peak, by,
                  float(out['d']['D_loss']), float(out['g']['G_loss'])))
        tr._graphs = None
        del tr, m
        torch.cuda.synchronize()
        K.filter_cache_reset()


if __name__ == '__main__':
    main()
