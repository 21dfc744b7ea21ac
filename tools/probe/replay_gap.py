"""What does the host-side part of an iteration cost on the device clock?  N x dg graph replay alone against N x trainer.iteration()
(= two lr_t fills + replay + bookkeeping)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench, t2i_amd
from t2i_amd import kernels as K
from t2i_amd.models.wgancls.model import WGanCls
from t2i_amd.models.wgancls.trainer import WGanClsTrainer
math = sys.argv[1] if len(sys.argv) > 1 else 'f32'
K.filter_cache(True); K.set_math(math)
if math == 'bf16':
    K.set_storage('bf16')
dev = torch.device('cuda'); cfg = bench.make_cfg(64)
m = WGanCls(cfg, device=dev, seed=0); tr = WGanClsTrainer(None, m, None, cfg)
feed = bench.synthetic_feed(cfg, dev, seed=1, with_noise=False)
tr.iteration(1, feed); tr.iteration(2, feed)
m.enable_graphs(feed)
feed.update({k: v for k, v in m.static_inputs().items() if feed.get(k) is not None})
for i in range(5):
    tr.iteration(3 + i, feed)
def timed(fn, n=60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    a = timed(lambda i: tr.iteration(10 + i, feed))
    b = timed(lambda i: m._graphs['dg'].replay())
    print('%s: trainer.iteration %.4f ms | bare graph replay %.4f ms | difference %.1f us' % (math, a, b, (a - b) * 1e3))
