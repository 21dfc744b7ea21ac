import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench, t2i_amd
from t2i_amd import kernels as K
from t2i_amd.models.wgancls.model import WGanCls
from t2i_amd.models.wgancls.trainer import WGanClsTrainer
dev = torch.device('cuda', 0)
K.filter_cache(True)
cfg = bench.make_cfg(64)
m = WGanCls(cfg, device=dev, seed=0)
tr = WGanClsTrainer(None, m, None, cfg)
feed = bench.synthetic_feed(cfg, dev, seed=1)
for i in range(2): tr.iteration(1 + i, feed)
torch.cuda.synchronize()
print('eager: max allocated %.2f GB, reserved %.2f GB, filter cache %.2f GB' % (torch.cuda.max_memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, K.filter_cache_bytes() / 2**30))
m.enable_graphs(feed)
for i in range(3): tr.iteration(3 + i, feed)
torch.cuda.synchronize()
free, total = torch.cuda.mem_get_info()
print('graphs: reserved %.2f GB, device in use %.2f GB of %.0f GB' % (torch.cuda.memory_reserved() / 2**30, (total - free) / 2**30, total / 2**30))
