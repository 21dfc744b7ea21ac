"""Which tensors does one config-3 iteration cast to bf16 in a launch of its own (K.cast_bf16 / K.cast_f32)?  Prints shape + the two nearest
callers outside kernels.py for every call of the third eager iteration.  Run on the GPU box."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import t2i_amd  # noqa
from t2i_amd import kernels as K
from t2i_amd.models.wgancls.model import WGanCls
from t2i_amd.models.wgancls.trainer import WGanClsTrainer
dev = torch.device('cuda')
K.filter_cache(True)
K.set_math('bf16'); K.set_storage('bf16')
cfg = bench.make_cfg(64)
m = WGanCls(cfg, device=dev, seed=0)
m.net_math = dict(K.CONFIG3_NET_MATH)
tr = WGanClsTrainer(None, m, None, cfg)
feed = bench.synthetic_feed(cfg, dev, seed=1, with_noise=False)
for it in (1, 2):
    tr.iteration(it, feed)
log = collections.Counter()
def wrap(name):
    orig = getattr(K, name)
    def f(t, *a, **k):
        import inspect
        which, fn = '?', '?'
        for fi in inspect.stack()[1:8]:
            if fi.function == '_operand_images':
                loc = fi.frame.f_locals
                which = 'a' if loc.get('a') is t else ('b' if loc.get('b') is t else '?')
            if fi.function in ('conv_fwd', 'conv_bwd_data', 'conv_bwd_filter', 'conv_bwd_pair', 'conv_fwd_stats'):
                fn = fi.function; break
        producer = getattr(t, 'grad_fn', None)
        st = [fr for fr in traceback.extract_stack()[:-1] if not fr.filename.endswith('kernels.py')][-3:]
        tag = '%s operand %s of %s base=%s' % (name, which, fn, None if t._base is None else tuple(t._base.shape))
        log[(tag, tuple(t.shape), str(t.dtype).replace('torch.', ''), ' < '.join('%s:%d %s' % (os.path.basename(fr.filename), fr.lineno, fr.name) for fr in reversed(st)))] += 1
        return orig(t, *a, **k)
    setattr(K, name, f)
for n in ('cast_bf16', 'cast_f32'):
    if hasattr(K, n):
        wrap(n)
tw = collections.Counter()
_orig_twin = K._twin_for
def _twin_for(out, *inputs):
    r = _orig_twin(out, *inputs)
    import inspect
    fn = inspect.stack()[1].function
    if out.dtype == torch.float32 and out.dim() == 4:
        tw[(fn, tuple(out.shape), r is not None, 'math=%s mixed=%s bwd=%s store=%s grad=%s' % (K._MATH[0], K._MIXED[0], K._BWD_MATH[0], str(K._STORE[0]).replace('torch.', ''), torch.is_grad_enabled()))] += 1
    return r
K._twin_for = _twin_for
tr.iteration(3, feed)
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print(v, k)
print('total', sum(log.values()))
print('---- _twin_for calls on fp32 rank-4 outputs (producer, shape, twin made, state)')
for k, v in sorted(tw.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    print(v, k)
