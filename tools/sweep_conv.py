#!/usr/bin/env python
"""Tile / split-K sweep of the implicit-GEMM kernel through the t2i_tuning_set force_tile / force_splitk hooks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import t2i_amd  # noqa: E402,F401
from t2i_amd import kernels as K
from t2i_amd._lib import lib  # noqa: E402
from tools.bench_conv import LAYERS, timeit  # noqa: E402

WANT = [('D2', 64), ('D3', 64), ('D4', 64), ('D7', 64), ('D10', 64), ('G5c', 64), ('G8c', 64), ('G4c', 64), ('D4', 192),
        ('D10', 192), ('D2', 192)]
if os.environ.get('T2I_SWEEP_MATH'):
    K.set_math(os.environ['T2I_SWEEP_MATH'])
if len(sys.argv) > 1:
    WANT = [(a.split(':')[0], int(a.split(':')[1])) for a in sys.argv[1:]]
L = {l[0]: l for l in LAYERS}
for name, B in WANT:
    _, H, W, Ci, Co, k, s, pad = L[name]
    d, ws = K.conv_desc(B, H, W, Ci, Co, k, k, s, s, pad)
    K.workspace(torch.device('cuda', 0), 1 << 30)
    x = torch.randn(B, H, W, Ci, device='cuda'); w = torch.randn(k, k, Ci, Co, device='cuda') * 0.05
    dy = torch.randn(B, d.Ho, d.Wo, Co, device='cuda')
    fl = K.conv_flops(d)
    big = 1 << 30
    fns = {'fwd': lambda: K.conv_fwd(x, w, None, d, big), 'bwdD': lambda: K.conv_bwd_data(dy, w, None, d, big),
           'bwdF': lambda: K.conv_bwd_filter(x, dy, d, big)}
    for mode, fn in fns.items():
        lib.t2i_tuning_set(b'force_tile', 0.0); lib.t2i_tuning_set(b'force_splitk', 0.0)
        t0 = timeit(fn, 10)
        res = []
        for tile in (22, 21, 12, 11):
            for sk in (1, 2, 3, 4, 6, 8):
                lib.t2i_tuning_set(b'force_tile', float(tile)); lib.t2i_tuning_set(b'force_splitk', float(sk))
                try:
                    res.append((timeit(fn, 10), tile, sk))
                except Exception as e:
                    pass
        if os.environ.get('T2I_SWEEP_DUMP'):
            M, N, Kd = {'fwd': (B * d.Ho * d.Wo, Co, k * k * Ci), 'bwdD': (B * ((H + s - 1) // s) * ((W + s - 1) // s), Ci, (k // s) * (k // s) * Co if s > 1 else k * k * Co),
                        'bwdF': (k * k * Ci, Co, B * d.Ho * d.Wo)}[mode]
            print('DUMP %s %d %s M=%d N=%d K=%d nphase=%d out=%d auto=%.2f %s' % (name, B, mode, M, N, Kd, s * s if mode == 'bwdD' else 1,
                  {'fwd': B * d.Ho * d.Wo * Co, 'bwdD': B * H * W * Ci, 'bwdF': k * k * Ci * Co}[mode], t0 * 1e6,
                  ' '.join('%d/%d:%.2f' % (tl, sk, t * 1e6) for t, tl, sk in res)))
        res.sort()
        print('%-5s B=%-3d %-5s auto %7.1f us (%5.1f TF) | best: %s' % (
            name, B, mode, t0 * 1e6, fl / t0 / 1e12,
            '  '.join('t%d/s%d %.1f' % (tl, sk, t * 1e6) for t, tl, sk in res[:5])))
