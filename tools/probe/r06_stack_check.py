"""stacked vs two-pass critic step, per tensor (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'workers'))
import torch
import t2i_amd  # noqa
from t2i_amd import stacked as ST
from t2i_amd.models.wgancls.model import WGanCls
import dp_diffdata_worker as W
dev = torch.device('cuda', 0)
width = int(os.environ.get('WIDTH', '32'))
for b in (4, 8, 16):
    cfg = W.make_cfg(b); cfg.MODEL.GF_DIM = cfg.MODEL.DF_DIM = width
    feed = W.part(W.full_feed(W.make_cfg(16), dev), 0, b)
    res = {}
    for mode in ('two_pass', 'stackA', 'stackB'):
        m = WGanCls(cfg, device=dev, seed=0)
        m.stack_xhat = mode != 'two_pass'
        prev = ST.defer_filter_gradients(mode == 'stackB')
        out = W.critic_only(m, feed)
        ST.defer_filter_gradients(prev)
        torch.cuda.synchronize()
        res[mode] = ({n: m.d_arena.grad_of(n).clone() for n in m.d_vars}, {k: float(out[k]) for k in ('D_loss', 'real_gp', 'wdist')})
    print('b=%d' % b, res['two_pass'][1], res['stackB'][1])
    for n in res['two_pass'][0]:
        a, sa, sb = res['two_pass'][0][n], res['stackA'][0][n], res['stackB'][0][n]
        print('  %-24s A %.2e  B %.2e' % (n, W.rel(sa, a), W.rel(sb, a)))
