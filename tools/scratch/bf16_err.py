import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import t2i_amd
from t2i_amd import kernels as K
from t2i_amd.models.wgancls.model import WGanCls
sys.path.insert(0, 'tests')
from test_step_gpu import _cfg, _feed
gs = np.load('tests/golden/step_tiny.npz')
gpu = torch.device('cuda')
K.set_math('bf16')
m = WGanCls(_cfg(8, 32, 16, 8, 8, 4), device=gpu)
m.store.load({k[len('param/'):]: gs[k] for k in gs.files if k.startswith('param/')})
feed = _feed(gs, gpu)
d = m.d_losses(feed); torch.cuda.synchronize()
def l2(got, ref):
    got = got.detach().double().cpu().numpy(); ref = np.asarray(ref, np.float64)
    return np.linalg.norm(got - ref) / np.linalg.norm(ref), np.abs(got - ref).max() / np.abs(ref).max()
for k, r in (('G', 'd/G'), ('Dx_hat_logit', 'd/Dx_hat'), ('grad_x_hat', 'd/grad_x_hat'), ('grad_cond', 'd/grad_cond')):
    print(k, 'relL2 %.3e  relmax %.3e' % l2(d[k], gs[r]))
for k in ('D_loss', 'D_loss_real', 'D_loss_fake', 'D_loss_mismatch', 'wdist', 'wdist2', 'real_gp', 'real_gp2'):
    print(k, float(d[k]), float(gs['d/' + k]), abs(float(d[k]) - float(gs['d/' + k])) / max(abs(float(gs['d/' + k])), 1.0))
def cosine(arena, names, prefix):
    got = torch.cat([arena.grad_of(n).reshape(-1).double().cpu() for n in names])
    ref = torch.cat([torch.from_numpy(np.asarray(gs[prefix + n], np.float64)).reshape(-1) for n in names])
    return float((got * ref).sum() / (got.norm() * ref.norm())), float((got - ref).norm() / ref.norm())
print('d grads cos, relL2', cosine(m.d_arena, list(m.d_vars), 'd/grad/'))
g = m.g_losses(feed)
print('G_loss', float(g['G_loss']), float(gs['g/G_loss']))
print('g.G', l2(g['G'], gs['g/G']))
print('g grads cos, relL2', cosine(m.g_arena, list(m.g_vars), 'g/grad/'))
