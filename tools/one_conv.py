#!/usr/bin/env python
"""Run one conv primitive a few times (for rocprofv3 --pmc runs).  usage: one_conv.py LAYER BATCH fwd|bwdD|bwdF [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import t2i_amd  # noqa: E402,F401
from t2i_amd import kernels as K  # noqa: E402
from tools.bench_conv import LAYERS  # noqa: E402

K.set_math(os.environ.get('T2I_ONE_MATH', 'f32'))
name, B, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
if os.environ.get('T2I_ONE_SHAPE'):           # "H,W,Ci,Co,k,s,pad" instead of a named layer (K sweeps)
    f = os.environ['T2I_ONE_SHAPE'].split(',')
    H, W, Ci, Co, k, s, pad = [int(v) for v in f[:6]] + [f[6]]
else:
    _, H, W, Ci, Co, k, s, pad = {l[0]: l for l in LAYERS}[name]
d, ws = K.conv_desc(B, H, W, Ci, Co, k, k, s, s, pad)
x = torch.randn(B, H, W, Ci, device='cuda'); w = torch.randn(k, k, Ci, Co, device='cuda') * 0.05
dy = torch.randn(B, d.Ho, d.Wo, Co, device='cuda')
fn = {'fwd': lambda: K.conv_fwd(x, w, None, d, ws), 'bwdD': lambda: K.conv_bwd_data(dy, w, None, d, ws),
      'bwdF': lambda: K.conv_bwd_filter(x, dy, d, ws)}[mode]
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print('flops', K.conv_flops(d))
