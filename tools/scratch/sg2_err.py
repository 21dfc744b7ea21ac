import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_stackgan as TS
from t2i_amd.models.stackgan.stageII.trainer import ConditionalGanTrainer as T2
gs = np.load(os.path.join(ROOT, 'tests/golden/stackgan2_tiny.npz'))
dev = torch.device('cuda')
m = TS._models(2, dev)
m.store.load({k[6:]: gs[k] for k in gs.files if k.startswith('param/')})
f = {k[5:]: torch.tensor(gs[k], dtype=torch.float32, device=dev) for k in gs.files if k.startswith('feed/')}
feed = {'inputs': f['x'], 'wrong_inputs': f['x_mismatch'], 'phi_inputs': f['cond'], 'z': f['z']}
feed.update({k: v for k, v in f.items() if k.startswith('ca_noise')})
tr = T2(None, m, None, m.cfg)
g = tr.g_losses(feed)
rows = []
for n in m.g_vars:
    ref = gs['g/grad/' + n].astype(np.float64); got = m.g_arena.grad_of(n).double().cpu().numpy()
    if np.abs(ref).max() < 1e-9: continue
    rows.append((np.linalg.norm(got - ref) / np.linalg.norm(ref), np.abs(got - ref).max() / np.abs(ref).max(), n))
rows.sort(reverse=True)
for r in rows[:12]: print('%.3e %.3e %s' % r)
got = torch.cat([m.g_arena.grad_of(n).reshape(-1).double().cpu() for n in m.g_vars]); ref = torch.cat([torch.from_numpy(gs['g/grad/' + n].astype(np.float64)).reshape(-1) for n in m.g_vars])
print('overall relL2 %.3e cos %.6f' % (float((got - ref).norm() / ref.norm()), float((got * ref).sum() / got.norm() / ref.norm())))
