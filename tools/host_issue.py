import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench, t2i_amd
from t2i_amd.dp import DataParallel
from t2i_amd.models.wgancls.model import WGanCls
from t2i_amd.models.wgancls.trainer import WGanClsTrainer
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
use_dp = os.environ.get('DP') == '1'
dp = None
if use_dp:
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29545', rank=0, world_size=1, device_id=dev)
    dp = DataParallel()
cfg = bench.make_cfg(64)
m = WGanCls(cfg, device=dev, seed=0, dp=dp)
tr = WGanClsTrainer(None, m, None, cfg)
feed = bench.synthetic_feed(cfg, dev, seed=1)
for i in range(4): tr.iteration(1 + i, feed)
torch.cuda.synchronize()
for n in (1, 2, 4):
    t0 = time.perf_counter()
    for i in range(n): tr.iteration(10 + i, feed)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print('dp=%s iters %d: host issue %.2f ms/iter, total %.2f ms/iter' % (use_dp, n, th / n * 1e3, tt / n * 1e3))
