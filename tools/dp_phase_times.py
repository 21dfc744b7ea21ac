#!/usr/bin/env python
"""Where a data-parallel graph-segment iteration spends its time on one rank (RCCL world size 1): GPU time of each graph
segment and exchange step (HIP events) next to the host time to issue them.  usage: dp_phase_times.py [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
import t2i_amd  # noqa: E402,F401
from t2i_amd.dp import DataParallel  # noqa: E402
from t2i_amd.models.wgancls.model import WGanCls  # noqa: E402
from t2i_amd.models.wgancls.trainer import WGanClsTrainer  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
use_dp = os.environ.get('DP', '1') == '1'
dp = None
if use_dp:
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29544', rank=0, world_size=1, device_id=dev)
    dp = DataParallel()
cfg = bench.make_cfg(64)
m = WGanCls(cfg, device=dev, seed=0, dp=dp)
tr = WGanClsTrainer(None, m, None, cfg)
feed = bench.synthetic_feed(cfg, dev, seed=1)
for i in range(3):
    tr.iteration(1 + i, feed)
m.enable_graphs(feed)
for i in range(3):
    tr.iteration(4 + i, feed)
g = m._graphs
if not use_dp:
    steps = [('d graph', lambda: g['d'].replay()), ('g graph', lambda: g['g'].replay())]
else:
  steps = [('d graph', lambda: g['d'].replay()), ('d exchange', lambda: dp.allreduce_arena(m.d_arena, extra=g['d_out']['wd_sums'])),
           ('d update', lambda: g['d_upd'].replay()), ('g graph', lambda: g['g'].replay()),
           ('g exchange', lambda: dp.allreduce_arena(m.g_arena)), ('g update', lambda: g['g_upd'].replay())]
gpu = {n: 0.0 for n, _ in steps}
host = {n: 0.0 for n, _ in steps}
torch.cuda.synchronize()
t_all = time.perf_counter()
for _ in range(iters):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(steps) + 1)]
    evs[0].record()
    for i, (n, fn) in enumerate(steps):
        t0 = time.perf_counter()
        fn()
        host[n] += time.perf_counter() - t0
        evs[i + 1].record()
    torch.cuda.synchronize()
    for i, (n, _) in enumerate(steps):
        gpu[n] += evs[i].elapsed_time(evs[i + 1])
wall = (time.perf_counter() - t_all) / iters * 1e3
for n, _ in steps:
    print('%-12s gpu %7.3f ms   host %7.3f ms' % (n, gpu[n] / iters, host[n] / iters * 1e3))
print('sum gpu %.3f ms, wall %.3f ms/iter (with a sync per iteration)' % (sum(gpu.values()) / iters, wall))
if use_dp:
    dist.destroy_process_group()
