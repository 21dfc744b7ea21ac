import sys, os; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import t2i_amd
from oracle import torch_step as T
from t2i_amd.models.wgancls.model import WGanCls
from test_step_gpu import _cfg
gpu = torch.device('cuda'); B = 8
ocfg = T.Cfg(batch=B)
P = {n: v.double() for n, v in T.init_variables(ocfg, seed=0).items()}
feed = {k: v.double() for k, v in T.synthetic_feed(ocfg, seed=1).items()}
m = WGanCls(_cfg(128, 1024, 128, 128, 128, B), device=gpu)
m.store.load({n: v.numpy() for n, v in P.items()})
f = {k: v.float().to(gpu) for k, v in feed.items()}
f['epsilon'] = f.pop('eps'); f['learning_rate_d'] = 1e-4; f['learning_rate_g'] = 1e-4
d = m.d_losses(f)
ref = T.d_step(P, ocfg, feed, 0.7)
ref32 = T.d_step({n: v.float() for n, v in P.items()}, ocfg, {k: v.float() for k, v in feed.items()}, 0.7)
sc = T.d_step_term_scales(P, ocfg, feed, 0.7)
for n in m.d_vars:
    r = ref['grads'][n].numpy(); o = m.d_arena.grad_of(n).double().cpu().numpy(); c = ref32['grads'][n].double().numpy()
    print('%-26s max|g| %.2e scale %.2e | err/scale ours %.2e cpu32 %.2e' % (n, np.abs(r).max(), sc[n], np.abs(o - r).max() / sc[n], np.abs(c - r).max() / sc[n]))
