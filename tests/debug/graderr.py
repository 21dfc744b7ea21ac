import sys, os; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import t2i_amd
from oracle import torch_step as T
from t2i_amd.models.wgancls.model import WGanCls
from test_step_gpu import _cfg, relerr
gpu = torch.device('cuda')
B = 8
ocfg = T.Cfg(batch=B)
P = {n: v.double() for n, v in T.init_variables(ocfg, seed=0).items()}
feed = {k: v.double() for k, v in T.synthetic_feed(ocfg, seed=1).items()}
m = WGanCls(_cfg(128, 1024, 128, 128, 128, B), device=gpu)
m.store.load({n: v.numpy() for n, v in P.items()})
f = {k: v.float().to(gpu) for k, v in feed.items()}
f['epsilon'] = f.pop('eps'); f['learning_rate_d'] = 1e-4; f['learning_rate_g'] = 1e-4
P32 = {n: v.float() for n, v in P.items()}; feed32 = {k: v.float() for k, v in feed.items()}
d = m.d_losses(f)
ref, ref32 = T.d_step(P, ocfg, feed, 0.7), T.d_step(P32, ocfg, feed32, 0.7)
print('env', {k: v for k, v in os.environ.items() if k.startswith('T2I')})
for k in ('D_loss', 'wdist', 'real_gp', 'real_gp2'):
    print(k, float(d[k]), ref[k], ref32[k])
print('Dx_hat', relerr(d['Dx_hat_logit'], ref['Dx_hat'].numpy()), 'gx', relerr(d['grad_x_hat'], ref['grad_x_hat'].numpy()), 'gc', relerr(d['grad_cond'], ref['grad_cond'].numpy()))
for n in m.d_vars:
    print('%-28s ours %.2e cpu32 %.2e' % (n, relerr(m.d_arena.grad_of(n), ref['grads'][n].numpy()), relerr(ref32['grads'][n], ref['grads'][n].numpy())))
