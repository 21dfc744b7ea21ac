"""How often does one wgancls iteration call each kernels.* entry (eager, B = 64)?  python tools/probe/k_calls.py [f32|bf16]"""
import collections, os, sys, types
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
tr.iteration(1, feed); torch.cuda.synchronize()
cnt = collections.Counter(); shapes = collections.defaultdict(collections.Counter)
for name in dir(K):
    f = getattr(K, name)
    if isinstance(f, types.FunctionType) and f.__module__ == K.__name__ and not name.startswith('_'):
        def wrap(f=f, name=name):
            def g(*a, **k):
                cnt[name] += 1
                t = next((x for x in a if isinstance(x, torch.Tensor)), None)
                if t is not None:
                    shapes[name][tuple(t.shape)] += 1
                return f(*a, **k)
            return g
        setattr(K, name, wrap())
tr.iteration(2, feed); torch.cuda.synchronize()
for n, c in cnt.most_common(40):
    print('%4d  %-24s %s' % (c, n, dict(shapes[n].most_common(6)) if n in ('add_act', 'axpby', 'act_bwd', 'act_fwd', 'bn_apply', 'col_reduce', 'cast_f32', 'bf16_image', 'cast_bf16') else ''))
