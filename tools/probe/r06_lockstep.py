"""New form (stacked critic step, paired generator, store-first slots, hipGraph replay) against the round-5 form (two critic passes, two
generator passes, zero-filled arenas, eager) in LOCKSTEP: before every iteration the round-5 model takes the new model's weights, Adam
state, kt and moving averages, both run the iteration on the same feed, and every logged scalar plus the gradient arenas are compared.
Differences cannot compound, so anything beyond the one-iteration rounding / kink level is a state bug of the new form (buffers or
bookkeeping that survive an iteration: image slots, seeds, kept transforms, touched sets, replayed graphs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import t2i_amd  # noqa
from t2i_amd import kernels as K
from t2i_amd import scope as S
from t2i_amd.models.wgancls.model import WGanCls
from t2i_amd.models.wgancls.trainer import WGanClsTrainer
B, N = int(os.environ.get('LS_B', '16')), int(os.environ.get('LS_ITERS', '10'))
dev = torch.device('cuda')
K.filter_cache(True)
cfg = bench.make_cfg(B)
new = WGanCls(cfg, device=dev, seed=0)
old = WGanCls(cfg, device=dev, seed=0)
old.stack_xhat = old.pair_g = False
old.d_arena.enable_sinks(store_first=False); old.g_arena.enable_sinks(store_first=False)
tn, to = WGanClsTrainer(None, new, None, cfg), WGanClsTrainer(None, old, None, cfg)
g = torch.Generator(device=dev).manual_seed(1)
worst = 0.0
for it in range(1, N + 1):
    feed = {'x': torch.rand(B, 64, 64, 3, generator=g, device=dev) * 2 - 1, 'x_mismatch': torch.rand(B, 64, 64, 3, generator=g, device=dev) * 2 - 1,
            'cond': torch.randn(B, 1024, generator=g, device=dev), 'z': torch.randn(B, 128, generator=g, device=dev),
            'epsilon': torch.rand(B, 1, 1, 1, generator=g, device=dev), 'learning_rate_d': 1e-4, 'learning_rate_g': 1e-4,
            'ca_noise_d': torch.randn(B, 128, generator=g, device=dev).clamp_(-2, 2), 'ca_noise_g': torch.randn(B, 128, generator=g, device=dev).clamp_(-2, 2)}
    with torch.no_grad():                      # the round-5 model starts the iteration from the new model's state
        for n, v in new.store.vars.items():
            old.store.vars[n].copy_(v)
        old.D_optim.v.copy_(new.D_optim.v); old.G_optim.v.copy_(new.G_optim.v)
        old.D_optim.t, old.G_optim.t = new.D_optim.t, new.G_optim.t
        old.kt.copy_(new.kt)
    K.filter_cache_invalidate()
    S.set_default_store(new.store)             # (utils/ops.py finds its variables in the process-wide default store: one model at a time)
    on = tn.iteration(it, feed)
    S.set_default_store(old.store)
    oo = to.iteration(it, feed)
    if it == 2:
        S.set_default_store(new.store)
        new.enable_graphs(feed)
    torch.cuda.synchronize()
    line = []
    for grp, keys in (('d', ('D_loss', 'wdist', 'wdist2', 'real_gp', 'real_gp2')), ('g', ('G_loss', 'G_kl_loss', 'D_loss_fake'))):
        for k in keys:
            a, b = float(on[grp][k]), float(oo[grp][k])
            e = abs(a - b) / max(abs(b), 1.0)
            worst = max(worst, e)
            line.append('%s %.2e' % (k, e))
    gd = float((new.d_arena.grad - old.d_arena.grad).norm() / old.d_arena.grad.norm())
    gg = float((new.g_arena.grad - old.g_arena.grad).norm() / old.g_arena.grad.norm())
    wd = float((new.d_arena.flat - old.d_arena.flat).abs().max())
    if it <= 2:
        for n in list(new.d_vars)[:4] + list(new.g_vars)[:3]:
            ar_n, ar_o = (new.d_arena, old.d_arena) if n in new.d_vars else (new.g_arena, old.g_arena)
            a, b = ar_n.grad_of(n), ar_o.grad_of(n)
            print('     %-30s |new| %.3e |old| %.3e rel %.2e' % (n, float(a.norm()), float(b.norm()), float((a - b).norm() / b.norm().clamp_min(1e-30))))
    print('it %2d  %s | grad arenas rel L2: critic %.2e generator %.2e | max |dw| after the update %.2e' % (it, '  '.join(line), gd, gg, wd))
print('worst scalar difference %.2e' % worst)
