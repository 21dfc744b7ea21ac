#!/usr/bin/env python
"""How long does the host take to ISSUE one iteration vs how long the GPU takes to run it? (launch-bound check)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import t2i_amd  # noqa
from t2i_amd.models.wgancls.model import WGanCls
from t2i_amd.models.wgancls.trainer import WGanClsTrainer
dev = torch.device('cuda', 0)
cfg = bench.make_cfg(64)
m = WGanCls(cfg, device=dev)
tr = WGanClsTrainer(None, m, None, cfg)
feed = bench.synthetic_feed(cfg, dev, 1)
for i in range(5):
    tr.iteration(1 + i, feed)
torch.cuda.synchronize()
iss, tot = [], []
for i in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.iteration(10 + i, feed)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    iss.append(t1 - t0); tot.append(t2 - t0)
print('issue ms: %.2f  total ms: %.2f' % (1e3 * sorted(iss)[5], 1e3 * sorted(tot)[5]))
# split D / G
for name, fn in (('d_step', m.d_step), ('g_step', m.g_step)):
    ts = []
    for i in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(feed); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts.sort()
    print('%s issue %.2f ms total %.2f ms' % (name, ts[3][0] * 1e3, ts[3][1] * 1e3))
