#!/usr/bin/env python
"""Fixed cost of a bf16 GEMM launch: K sweep at 256 tiles of 128x128, also with an fp32 output and an empty kernel for reference."""
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


def run(M, N, Kd, tile, sk, out_dtype=None):
    B = M // 256
    d, ws = K.conv_desc(B, 16, 16, Kd, N, 1, 1, 1, 1, 'VALID')
    x = torch.randn(B, 16, 16, Kd, device='cuda').bfloat16()
    w = torch.randn(1, 1, Kd, N, device='cuda') * 0.05
    lib.t2i_tuning_set(b'force_tile', float(tile)); lib.t2i_tuning_set(b'force_splitk', float(sk))
    t = timeit(lambda: K.conv_fwd(x, w, None, d, 1 << 30, out_dtype=out_dtype), 20)
    lib.t2i_tuning_set(b'force_tile', 0.0); lib.t2i_tuning_set(b'force_splitk', 0.0)
    return t * 1e6


print('lib', os.environ.get('T2I_HIP_LIB', 'default'))
for M in (2048, 16384, 32768):
    print('M=%d N=256 tile 128x128: ' % M + '  '.join('K%d %.1f' % (Kd, run(M, 256, Kd, 22, 1)) for Kd in (64, 128, 512, 2048)))
print('M=16384 fp32 out: ' + '  '.join('K%d %.1f' % (Kd, run(16384, 256, Kd, 22, 1, torch.float32)) for Kd in (64, 128, 512, 2048)))
a = torch.zeros(1024, device='cuda')
print('tiny torch kernel back-to-back: %.2f us' % (timeit(lambda: a.add_(1.0), 20) * 1e6))
