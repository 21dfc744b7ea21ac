#!/usr/bin/env python
"""A layer's backward pair (input gradient + filter gradient on bf16 tensors): two launches vs t2i_conv2d_bwd_pair's one."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import t2i_amd  # noqa: E402,F401
from t2i_amd import kernels as K  # noqa: E402
from t2i_amd._lib import lib  # noqa: E402
from tools.bench_conv import LAYERS, timeit  # noqa: E402

K.set_math('bf16'); K.set_storage('bf16')
dev = torch.device('cuda')
K.workspace(dev, 1 << 30)
L = {l[0]: l for l in LAYERS}
tot = [0.0, 0.0, 0.0]
for name, B in (('D2', 64), ('D3', 64), ('D4', 64), ('D7', 64), ('D10', 64), ('G5c', 64), ('G7c', 64), ('G8c', 64), ('G4c', 64), ('G6c', 64), ('D2', 192), ('D3', 192), ('D4', 192), ('D10', 192)):
    _, H, W, Ci, Co, k, s, pad = L[name]
    d, ws = K.conv_desc(B, H, W, Ci, Co, k, k, s, s, pad)
    x = torch.randn(B, H, W, Ci, device=dev).bfloat16()
    w = torch.randn(k, k, Ci, Co, device=dev) * 0.05
    dy = torch.randn(B, d.Ho, d.Wo, Co, device=dev).bfloat16()
    dw = torch.zeros(k * k * Ci * Co, device=dev)

    def sep():
        K.conv_bwd_data(dy, w, None, d, 1 << 30, out_dtype=torch.bfloat16)
        K.conv_bwd_filter(x, dy, d, 1 << 30, out=dw)

    def pair():
        K.conv_bwd_pair(K.PAIR_BWD_DATA, dy, w, x, dy, d, 1 << 30, dw, out_dtype=torch.bfloat16)
    K.tuning_set('debug_plan', 1)
    n0 = lib.t2i_stat(b'pair_fused'); pair(); fused = lib.t2i_stat(b'pair_fused') - n0
    K.tuning_set('debug_plan', 0)
    t_sep = timeit(sep, 10) * 1e6
    K.tuning_set('pair', 1); t_f = timeit(pair, 10) * 1e6
    K.tuning_set('pair', 0); t_u = timeit(pair, 10) * 1e6
    K.tuning_set('pair', 1)
    if B == 64:
        tot[0] += t_sep; tot[1] += t_f; tot[2] += t_u
    print('%-4s B=%-3d two calls %6.1f  pair fused(%d) %6.1f  pair unfused %6.1f us' % (name, B, t_sep, fused, t_f, t_u))
print('B=64 totals: two calls %.1f  fused %.1f  unfused %.1f' % tuple(tot))
