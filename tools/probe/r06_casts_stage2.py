"""cast_bf16 launches of one StackGAN Stage-II iteration in the compliant arithmetic (kernels.FWD_F32_BWD_BF16): who asks for them.  GPU box."""
import collections, inspect, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import t2i_amd  # noqa
from t2i_amd import kernels as K
from t2i_amd.models.stackgan.run import build
from t2i_amd.utils.config import config_from_yaml
PKG = os.path.join(ROOT, 'text-to-image_amd', 'models')
dev = torch.device('cuda')
K.filter_cache(True)
K.set_math('bf16')
cfg1 = config_from_yaml(os.path.join(PKG, 'stackgan', 'stageI', 'cfg', 'flowers.yml'))
cfg = config_from_yaml(os.path.join(PKG, 'stackgan', 'stageII', 'cfg', 'flowers.yml'))
cfg.TRAIN.BATCH_SIZE = cfg1.TRAIN.BATCH_SIZE = 32
model, tr = build(2, cfg, cfg1, device=dev)
nm = {'g_net': K.FWD_F32_BWD_BF16, 'd_net': K.FWD_F32_BWD_BF16}
model.net_math = dict(nm); model.stagei.net_math = dict(nm)
feed = tr.make_feed()
tr.iteration(feed); tr.iteration(feed)
log = collections.Counter(); elems = collections.Counter()
orig = K.cast_bf16
def f(t, *a, **k):
    which, fn = '?', '?'
    for fi in inspect.stack()[1:8]:
        if fi.function == '_operand_images':
            loc = fi.frame.f_locals
            which = 'a' if loc.get('a') is t else ('b' if loc.get('b') is t else '?')
        if fi.function in ('conv_fwd', 'conv_bwd_data', 'conv_bwd_filter', 'conv_bwd_pair', 'conv_fwd_stats'):
            fn = fi.function; break
    st = [fr for fr in traceback.extract_stack()[:-1] if not fr.filename.endswith('kernels.py')][-2:]
    key = ('operand %s of %s' % (which, fn), ' < '.join('%s:%d %s' % (os.path.basename(fr.filename), fr.lineno, fr.name) for fr in reversed(st)))
    log[key] += 1; elems[key] += t.numel()
    return orig(t, *a, **k)
K.cast_bf16 = f
tr.iteration(feed)
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -elems[kv[0]]):
    print('%3d casts %8.1f M elements  %s' % (v, elems[k] / 1e6, k))
print('total', sum(log.values()), 'casts,', sum(elems.values()) / 1e6, 'M elements (6 bytes each)')
