#!/usr/bin/env python
"""The persistent batched fp32 GEMM of the Winograd paths (t2i_bgemm.hip) against the vendor library on the same shapes: every
`[t2i plan] batched ...` launch of one fp32 wgancls iteration at B = 64 is re-timed as torch.bmm / torch.baddbmm-free strided batched
SGEMM (rocBLAS / hipBLASLt behind torch, exact fp32: allow_tf32 is off) under the same graph-replay timer.
    python tools/probe/bgemm_vs_rocblas.py"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CHILD = r'''
import os, sys
os.environ['T2I_DEBUG_PLAN'] = '1'
sys.path.insert(0, %r)
import torch, bench, t2i_amd
from t2i_amd import kernels as K
from t2i_amd.models.wgancls.model import WGanCls
from t2i_amd.models.wgancls.trainer import WGanClsTrainer
K.filter_cache(True)
dev = torch.device('cuda'); cfg = bench.make_cfg(64)
m = WGanCls(cfg, device=dev, seed=0); tr = WGanClsTrainer(None, m, None, cfg)
feed = bench.synthetic_feed(cfg, dev, seed=1, with_noise=False)
tr.iteration(1, feed); torch.cuda.synchronize()
sys.stderr.write('==== iteration 2\n'); sys.stderr.flush()
tr.iteration(2, feed); torch.cuda.synchronize()
''' % ROOT
r = subprocess.run([sys.executable, '-c', CHILD], capture_output=True, text=True)
log = r.stderr.split('==== iteration 2')[-1]
shapes = collections.Counter()
for mm in re.finditer(r'batched x(\d+) M=(\d+) N=(\d+) K=(\d+) mode (\d+)', log):
    shapes[tuple(int(x) for x in mm.groups())] += 1
import torch
import t2i_amd  # noqa
from t2i_amd import kernels as K
from t2i_amd._lib import lib
from tools.bench_conv import timeit
torch.backends.cuda.matmul.allow_tf32 = False
tot_mine = tot_lib = 0.0
print('%-44s %5s %10s %10s %8s %8s' % ('shape (batch x M x N x K, mode)', 'calls', 'rocBLAS us', 'TF/s', '', ''))
rows = []
for (nb, M, N, Kd, mode), cnt in sorted(shapes.items()):
    # mode 0: C = A[M,K] B[K,N]; 1: C = A[M,K] B[N,K]^T; 2: C[M,N] = A[K,M]^T B[K,N]  (layouts of t2i_bgemm.hip)
    if mode == 0:
        a = torch.randn(nb, M, Kd, device='cuda'); b = torch.randn(nb, Kd, N, device='cuda'); f = lambda: torch.bmm(a, b)
    elif mode == 1:
        a = torch.randn(nb, M, Kd, device='cuda'); b = torch.randn(nb, N, Kd, device='cuda'); f = lambda: torch.bmm(a, b.transpose(1, 2))
    else:
        a = torch.randn(nb, Kd, M, device='cuda'); b = torch.randn(nb, Kd, N, device='cuda'); f = lambda: torch.bmm(a.transpose(1, 2), b)
    out = torch.empty(nb, M, N, device='cuda')
    if mode == 0:
        g = lambda: torch.bmm(a, b, out=out)
    elif mode == 1:
        g = lambda: torch.bmm(a, b.transpose(1, 2), out=out)
    else:
        g = lambda: torch.bmm(a.transpose(1, 2), b, out=out)
    t = timeit(g, 10) * 1e6
    fl = 2.0 * nb * M * N * Kd
    rows.append((nb, M, N, Kd, mode, cnt, t, fl))
    print('%-44s %5d %10.1f %10.1f' % ('%d x %d x %d x %d, mode %d' % (nb, M, N, Kd, mode), cnt, t, fl / t / 1e6))
    tot_lib += cnt * t
fl_all = sum(c * f for (_, _, _, _, _, c, _, f) in rows)
print('one iteration: %d batched launches, %.1f GFLOP executed; vendor library %.0f us = %.1f TF/s' % (sum(shapes.values()), fl_all / 1e9, tot_lib, fl_all / tot_lib / 1e6))
print('(t2i bgemm_kernel on the same launches: profiles/r04_kernel_stats_summary.txt — 8.09 ms per iteration)')
