import sys, os; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, ctypes
import t2i_amd
from t2i_amd import kernels as K
from oracle import torch_step as T
from t2i_amd.models.wgancls.model import WGanCls
from test_step_gpu import _cfg
gpu = torch.device('cuda')
B = 8
ocfg = T.Cfg(batch=B)
P = T.init_variables(ocfg, seed=0)
feed = T.synthetic_feed(ocfg, seed=1)
m = WGanCls(_cfg(128, 1024, 128, 128, 128, B), device=gpu)
m.store.load({n: v.numpy() for n, v in P.items()})
f = {k: v.float().to(gpu) for k, v in feed.items()}
f['epsilon'] = f.pop('eps'); f['learning_rate_d'] = 1e-4; f['learning_rate_g'] = 1e-4
REC = []
def wrap(name):
    orig = getattr(K, name)
    def w(*a, **k):
        out = orig(*a, **k)
        o = out[0] if isinstance(out, tuple) else out
        shp = tuple(a[0].shape)
        REC.append((name, shp, tuple(o.shape), o.detach().clone()))
        return out
    setattr(K, name, w)
for nm in ('conv_fwd', 'conv_bwd_data', 'conv_bwd_filter', 'col_reduce', 'act_bwd', 'add_act', 'concat_tile_bwd', 'concat_tile_fwd'):
    wrap(nm)
runs = {}
for tag, env in (('thin', {}), ('nothin', {'T2I_NO_THIN': '1'})):
    for k in ('T2I_NO_THIN',):
        os.environ.pop(k, None)
    os.environ.update(env)
    REC.clear()
    m.d_losses(f)
    torch.cuda.synchronize()
    runs[tag] = list(REC)
a, b = runs['thin'], runs['nothin']
print(len(a), len(b))
for i, (ra, rb) in enumerate(zip(a, b)):
    same = ra[0] == rb[0] and ra[2] == rb[2]
    d = float((ra[3].double() - rb[3].double()).abs().max()) if same else -1
    mx = float(rb[3].abs().max())
    if d != 0.0:
        print(i, ra[0], ra[1], ra[2], 'maxdiff %.3e (max %.3e)' % (d, mx))
