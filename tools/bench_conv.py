#!/usr/bin/env python
"""Per-layer micro-benchmark of the three conv primitives at the wgancls shapes (tuning tool, not a test).
usage: python tools/bench_conv.py [--batch 64] [--filter D10] [--reps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import t2i_amd  # noqa: E402,F401
from t2i_amd import kernels as K  # noqa: E402

# name, H, W, Cin, Cout, k, s, pad, batch multiplier (3 = the batched critic pass)
LAYERS = [
    ('D1', 64, 64, 3, 128, 4, 2, 'SAME'), ('D2', 32, 32, 128, 256, 4, 2, 'SAME'), ('D3', 16, 16, 256, 512, 4, 2, 'SAME'),
    ('D4', 8, 8, 512, 1024, 4, 2, 'SAME'), ('D5', 4, 4, 1024, 256, 1, 1, 'VALID'), ('D6', 4, 4, 256, 512, 3, 1, 'SAME'),
    ('D7', 4, 4, 512, 1024, 3, 1, 'SAME'), ('D8fc', 1, 1, 1024, 128, 1, 1, 'VALID'), ('D10', 4, 4, 1152, 1024, 3, 1, 'SAME'),
    ('D11', 4, 4, 1024, 1024, 1, 1, 'VALID'), ('D12', 4, 4, 1024, 1, 4, 4, 'VALID'),
    ('G3fc', 1, 1, 256, 16384, 1, 1, 'VALID'), ('G4a', 4, 4, 1024, 256, 1, 1, 'VALID'), ('G4b', 4, 4, 256, 256, 3, 1, 'SAME'),
    ('G4c', 4, 4, 256, 1024, 3, 1, 'SAME'), ('G5dc', 8, 8, 512, 1024, 4, 2, 'SAME'), ('G5c', 8, 8, 512, 512, 3, 1, 'SAME'),
    ('G6a', 8, 8, 512, 128, 1, 1, 'VALID'), ('G6b', 8, 8, 128, 128, 3, 1, 'SAME'), ('G6c', 8, 8, 128, 512, 3, 1, 'SAME'),
    ('G7dc', 16, 16, 256, 512, 4, 2, 'SAME'), ('G7c', 16, 16, 256, 256, 3, 1, 'SAME'), ('G8dc', 32, 32, 128, 256, 4, 2, 'SAME'),
    ('G8c', 32, 32, 128, 128, 3, 1, 'SAME'), ('G9dc', 64, 64, 3, 128, 4, 2, 'SAME'), ('G9c', 64, 64, 3, 3, 3, 1, 'SAME'),
]


def timeit(fn, reps):
    """Device time per call: `reps` back-to-back calls captured into ONE hipGraph and replayed (no host launch cost between
    them — a bf16 conv of 20 us would otherwise be timed at the host's ~25 us issue rate).  T2I_BENCH_EAGER=1: eager launches."""
    if os.environ.get('T2I_BENCH_EAGER') != '1':
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        s.record()
        for _ in range(n):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        del g
        return s.elapsed_time(e) / (reps * n) * 1e-3
    return timeit_eager(fn, reps)


def timeit_eager(fn, reps):
    import time
    t0 = time.perf_counter()
    n = 0
    while n < 3 or time.perf_counter() - t0 < 0.02:      # >= 20 ms of back-to-back launches: clocks ramped, caches warm
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--filter', default='')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--math', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--cache', action='store_true', help='transformed-filter cache on, as in the training step (filter transforms leave the timed calls)')
    ap.add_argument('--storage', default='f32', choices=['f32', 'bf16'], help='bf16 (with --math bf16): the activation tensors handed to the entry points are bf16 (config 3 as it runs: no cast launches inside the timed calls)')
    a = ap.parse_args()
    K.set_math(a.math)
    if a.storage == 'bf16':
        K.set_storage('bf16')
    if a.cache:
        K.filter_cache(True)
    tot = {'fwd': [0, 0, 0], 'bwd_data': [0, 0, 0], 'bwd_filter': [0, 0, 0]}
    # TF/s = direct-convolution FLOPs / time (what bench.py's roofline block counts); algo: G = implicit GEMM, W3 = Winograd
    # F(2x2,3x3) (executes 1/2.25 of those FLOPs), W2 = Winograd F(2x2,2x2) on 4x4 stride 2 (9/16), S = small direct kernel
    print('%-6s %5s %22s %10s | %8s %6s | %8s %6s | %8s %6s | %s' % ('layer', 'B', 'shape', 'GFLOP', 'fwd us', 'TF/s', 'bwdD us', 'TF/s', 'bwdF us', 'TF/s', 'algo f/d/w'))
    for name, H, W, Ci, Co, k, s, pad in LAYERS:
        if a.filter and a.filter not in name:
            continue
        B = a.batch
        d, ws = K.conv_desc(B, H, W, Ci, Co, k, k, s, s, pad)
        x = torch.randn(B, H, W, Ci, device='cuda'); w = torch.randn(k, k, Ci, Co, device='cuda') * 0.05
        dy = torch.randn(B, d.Ho, d.Wo, Co, device='cuda')
        if a.storage == 'bf16':           # what config 3 hands the entry points: bf16 activations wherever the channel count allows
            if Ci % 64 == 0:
                x = x.bfloat16()
            if Co % 64 == 0:
                dy = dy.bfloat16()
        fl = K.conv_flops(d)
        t1 = timeit(lambda: K.conv_fwd(x, w, None, d, ws), a.reps)
        t2 = timeit(lambda: K.conv_bwd_data(dy, w, None, d, ws), a.reps)
        t3 = timeit(lambda: K.conv_bwd_filter(x, dy, d, ws), a.reps)
        for key, t in (('fwd', t1), ('bwd_data', t2), ('bwd_filter', t3)):
            tot[key][0] += fl; tot[key][1] += t
        short = {'implicit_gemm': 'G', 'winograd_f2x2_3x3': 'W3', 'winograd_f2x2_2x2': 'W2', 'direct_small': 'S', 'implicit_gemm_bf16_operands': 'H'}
        algos = '/'.join(short[K.conv_algo(d, m)] for m in ('fwd', 'bwd_data', 'bwd_filter'))
        for key in ('fwd', 'bwd_data', 'bwd_filter'):
            tot[key][2] += fl * K.ALGO_MAC_RATIO[K.ALGO_NAMES.index(K.conv_algo(d, key))]
        print('%-6s %5d %22s %10.2f | %8.1f %6.1f | %8.1f %6.1f | %8.1f %6.1f | %s' % (
            name, B, '%dx%dx%d->%d k%ds%d' % (H, W, Ci, Co, k, s), fl / 1e9, t1 * 1e6, fl / t1 / 1e12, t2 * 1e6, fl / t2 / 1e12,
            t3 * 1e6, fl / t3 / 1e12, algos))
    for key, (f, t, fx) in tot.items():
        if t:
            print('TOTAL %-10s %8.2f GFLOP %8.1f us  %6.1f TF/s   (executed on the matrix cores: %.2f GFLOP, %.1f TF/s)' % (
                key, f / 1e9, t * 1e6, f / t / 1e12, fx / 1e9, fx / t / 1e12))


if __name__ == '__main__':
    main()
