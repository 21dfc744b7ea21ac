#!/usr/bin/env python
"""bf16 GEMM (1x1 conv on bf16 tensors) timing against K-tiles and tile count, forced tile / split: what is fixed per launch,
what scales with the K loop, what a second workgroup per CU buys.  usage: gemm_sweep.py [f32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import t2i_amd  # noqa: E402,F401
from t2i_amd import kernels as K  # noqa: E402
from t2i_amd._lib import lib  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402

K.set_math('bf16'); K.set_storage('bf16')
K.workspace(torch.device('cuda', 0), 1 << 30)


def run(M, N, Kd, tile, sk, mode='fwd'):
    B = M // 256
    d, ws = K.conv_desc(B, 16, 16, Kd, N, 1, 1, 1, 1, 'VALID')
    x = torch.randn(B, 16, 16, Kd, device='cuda').bfloat16()
    w = torch.randn(1, 1, Kd, N, device='cuda') * 0.05
    dy = torch.randn(B, 16, 16, N, device='cuda').bfloat16()
    lib.t2i_tuning_set(b'force_tile', float(tile)); lib.t2i_tuning_set(b'force_splitk', float(sk))
    fn = {'fwd': lambda: K.conv_fwd(x, w, None, d, 1 << 30), 'bwdD': lambda: K.conv_bwd_data(dy, w, None, d, 1 << 30),
          'bwdF': lambda: K.conv_bwd_filter(x, dy, d, 1 << 30)}[mode]
    t = timeit(fn, 10)
    lib.t2i_tuning_set(b'force_tile', 0.0); lib.t2i_tuning_set(b'force_splitk', 0.0)
    return t * 1e6


for mode in ('fwd', 'bwdF'):
    print('== %s: K sweep, M=16384 N=256 (256 tiles of 128x128, 512 of 64x128), unsplit' % mode)
    for Kd in (128, 256, 512, 1024, 2048, 4096, 8192):
        ts = [run(16384, 256, Kd, tile, 1, mode) for tile in (22, 12, 21, 11)]
        fl = 2.0 * 16384 * 256 * Kd
        print('K=%5d  t22 %6.1f us (%6.1f TF)  t12 %6.1f  t21 %6.1f  t11 %6.1f' % (Kd, ts[0], fl / ts[0] / 1e6, ts[1], ts[2], ts[3]))
    print('== %s: M sweep at K=2048 N=256, tile 128x128 unsplit (tiles = M/64)' % mode)
    for M in (2048, 4096, 8192, 16384, 32768, 65536, 131072):
        ts = [run(M, 256, 2048, tile, 1, mode) for tile in (22, 12)]
        fl = 2.0 * M * 256 * 2048
        print('M=%6d tiles22=%4d  t22 %6.1f us (%6.1f TF)  t12 %6.1f (%6.1f TF)' % (M, M // 128 * 2, ts[0], fl / ts[0] / 1e6, ts[1], fl / ts[1] / 1e6))
print('== fwd split-K at M=4096 N=512 K=4096 (128 tiles of 128x128)')
for sk in (1, 2, 4, 8):
    ts = [run(4096, 512, 4096, tile, sk) for tile in (22, 12, 11)]
    print('split %d: t22 %6.1f  t12 %6.1f  t11 %6.1f' % (sk, ts[0], ts[1], ts[2]))
