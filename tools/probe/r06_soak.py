"""Loss trajectory of N wgancls iterations (B = 16, full width, hipGraph replay from iteration 3) — run once per setting of the round-6
toggles (environment) and compare: same seeds, so the first iterations agree to rounding and the later ones stay close (diagnostic)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import t2i_amd  # noqa
from t2i_amd import kernels as K
from t2i_amd.models.wgancls.model import WGanCls
from t2i_amd.models.wgancls.trainer import WGanClsTrainer
N = int(os.environ.get('SOAK_ITERS', '60'))
B = int(os.environ.get('SOAK_B', '16'))
dev = torch.device('cuda')
K.filter_cache(True)
torch.manual_seed(7); torch.cuda.manual_seed_all(7)
cfg = bench.make_cfg(B)
m = WGanCls(cfg, device=dev, seed=0)
tr = WGanClsTrainer(None, m, None, cfg)
rows = []
g = torch.Generator(device=dev).manual_seed(1)
for it in range(1, N + 1):
    feed = {'x': torch.rand(B, 64, 64, 3, generator=g, device=dev) * 2 - 1, 'x_mismatch': torch.rand(B, 64, 64, 3, generator=g, device=dev) * 2 - 1,
            'cond': torch.randn(B, 1024, generator=g, device=dev), 'z': torch.randn(B, 128, generator=g, device=dev),
            'epsilon': torch.rand(B, 1, 1, 1, generator=g, device=dev), 'learning_rate_d': 1e-4, 'learning_rate_g': 1e-4,
            'ca_noise_d': torch.randn(B, 128, generator=g, device=dev).clamp_(-2, 2), 'ca_noise_g': torch.randn(B, 128, generator=g, device=dev).clamp_(-2, 2)}
    out = tr.iteration(it, feed)
    if it == 2:
        m.enable_graphs(feed)
    rows.append([float(out['d']['D_loss']), float(out['d']['wdist']), float(out['d']['real_gp']), float(out['g']['G_loss']), float(m.kt)])
ok = all(all(x == x and abs(x) < 1e9 for x in r) for r in rows)
w = torch.cat([m.d_arena.flat, m.g_arena.flat])
print(json.dumps({'finite': ok and bool(torch.isfinite(w).all()), 'rows': rows, 'wnorm': float(w.norm())}))
