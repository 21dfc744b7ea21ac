"""Which torch-native ops does one eager wgancls iteration launch, and from where?  (torch.profiler with Python stacks)"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import t2i_amd  # noqa
from t2i_amd import kernels as K
from t2i_amd.models.wgancls.model import WGanCls
from t2i_amd.models.wgancls.trainer import WGanClsTrainer
math = sys.argv[1] if len(sys.argv) > 1 else 'f32'
K.filter_cache(True)
K.set_math(math)
if math == 'bf16':
    K.set_storage('bf16')
dev = torch.device('cuda')
cfg = bench.make_cfg(64)
m = WGanCls(cfg, device=dev, seed=0)
tr = WGanClsTrainer(None, m, None, cfg)
feed = bench.synthetic_feed(cfg, dev, seed=1, with_noise=False)
tr.iteration(1, feed); tr.iteration(2, feed)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.iteration(3, feed)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith('aten::') and e.device_type.name == 'CPU' and e.cuda_time_total > 0 and not any(c.name.startswith('aten::') and c.cuda_time_total > 0 for c in e.cpu_children):
        st = [s for s in (e.stack or []) if 'text-to-image_amd' in s or 'bench.py' in s]
        where = st[0].split('text-to-image_amd/')[-1] if st else '(autograd engine / no python frame)'
        cnt[(e.name, where)] += 1
for (name, where), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print('%3d  %-28s %s' % (n, name, where))
