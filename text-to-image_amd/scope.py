"""TF-1.x style variable scopes for an eager runtime.

The reference's operator wrappers (utils/ops.py) create their parameters implicitly through TF variable scopes:
``tf.contrib.layers.conv2d`` names its scope ``Conv``, ``Conv_1``, ... in call order inside the enclosing
``tf.variable_scope("d_net", reuse=...)`` (reference models/wgancls/model.py:134,167) and ``reuse=True`` re-enters the
same names.  This module reproduces exactly that naming so that a parameter dictionary keyed by TF variable names
(``d_net/Conv_3/weights``, ``g_net/BatchNorm_4/moving_mean`` — SURVEY.md appendix A) is interchangeable with the
reference's checkpoints' key space, and so that ``trainable_variables('d_net')`` (model.py:59-60) has an equivalent.
"""
import contextlib
import math
from collections import OrderedDict

import torch


class VariableStore(object):
    def __init__(self, device=None, seed=0):
        self.device = torch.device(device) if device is not None else torch.device(
            'cuda' if torch.cuda.is_available() else 'cpu')
        self.vars = OrderedDict()        # full name -> tensor (leaf; requires_grad iff trainable)
        self.trainable = OrderedDict()   # full name -> bool
        self.gen = torch.Generator(device='cpu').manual_seed(seed)
        self._stack = []                 # [(name, reuse, op-name counters)]
        self._frozen = set()             # scope prefixes whose variables are handed out detached

    # ---- scopes -----------------------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name, reuse=False):
        self._stack.append((name, reuse, {}))
        try:
            yield
        finally:
            self._stack.pop()

    @contextlib.contextmanager
    def frozen(self, prefix):
        """Variables under `prefix` are returned detached: no parameter gradients are produced for them (the
        generator step differentiates through the critic wrt its input only, reference model.py:102-106)."""
        self._frozen.add(prefix)
        try:
            yield
        finally:
            self._frozen.discard(prefix)

    def prefix(self):
        return '/'.join(n for n, _, _ in self._stack)

    def reuse(self):
        return any(r for _, r, _ in self._stack)

    def unique_op_name(self, base):
        """TF auto-naming: 'Conv', 'Conv_1', ... per enclosing scope entry."""
        if not self._stack:
            raise RuntimeError('ops must be called inside a variable_scope')
        counters = self._stack[-1][2]
        k = counters.get(base, 0)
        counters[base] = k + 1
        return base if k == 0 else '%s_%d' % (base, k)

    # ---- variables ----------------------------------------------------------------------------------------------------
    def get_variable(self, name, shape, initializer, trainable=True):
        full = self.prefix() + '/' + name
        v = self.vars.get(full)
        if v is None:
            if self.reuse():
                raise ValueError('Variable %s does not exist, or was not created with reuse=False' % full)
            data = initializer(tuple(shape), self.gen).to(dtype=torch.float32)
            v = data.to(self.device).requires_grad_(trainable)
            self.vars[full] = v
            self.trainable[full] = trainable
        else:
            if not self.reuse():
                raise ValueError('Variable %s already exists; did you mean reuse=True?' % full)
            if tuple(v.shape) != tuple(shape):
                raise ValueError('Variable %s has shape %s, requested %s' % (full, tuple(v.shape), tuple(shape)))
        if any(full.startswith(p + '/') for p in self._frozen):
            return v.detach()
        return v

    def trainable_variables(self, prefix):
        """tf.trainable_variables(scope) (reference models/wgancls/model.py:59-60): name -> tensor, creation order."""
        return OrderedDict((n, v) for n, v in self.vars.items() if n.startswith(prefix) and self.trainable[n])

    def global_variables(self, prefix=''):
        return OrderedDict((n, v) for n, v in self.vars.items() if n.startswith(prefix))

    def load(self, values, strict=True):
        """values: name -> array-like (TF layouts).  Copies into the existing storage (arena views stay valid)."""
        with torch.no_grad():
            for n, a in values.items():
                if n not in self.vars:
                    if strict:
                        raise KeyError('unknown variable %s' % n)
                    continue
                t = torch.as_tensor(a, dtype=torch.float32)
                if tuple(t.shape) != tuple(self.vars[n].shape):
                    raise ValueError('%s: shape %s != %s' % (n, tuple(t.shape), tuple(self.vars[n].shape)))
                self.vars[n].copy_(t.to(self.vars[n].device))
        from . import kernels as K
        K.filter_cache_invalidate()                  # filters changed behind the optimizer's back

    def state(self):
        return OrderedDict((n, v.detach().cpu().numpy().copy()) for n, v in self.vars.items())


# ---- initializers (host side, run once at variable creation) -----------------------------------------------------------
def truncated_normal_init(std, mean=0.0):
    """tf.truncated_normal: N(mean, std) re-drawn until within 2 std."""
    def init(shape, gen):
        t = torch.empty(shape, dtype=torch.float32)
        torch.nn.init.trunc_normal_(t, mean=mean, std=std, a=mean - 2 * std, b=mean + 2 * std, generator=gen)
        return t
    return init


def he_init(fan_in):
    """variance_scaling_initializer(factor=2.0, mode='FAN_IN', uniform=False) (reference utils/ops.py:60,68,86):
    truncated normal with stddev sqrt(1.3 * 2 / fan_in)."""
    return truncated_normal_init(math.sqrt(1.3 * 2.0 / fan_in))


def normal_init(std, mean=0.0):
    """tf.random_normal_initializer (gancls: reference models/gancls/model.py:28-31)."""
    def init(shape, gen):
        return torch.randn(shape, generator=gen, dtype=torch.float32) * std + mean
    return init


def constant_init(value):
    def init(shape, gen):
        return torch.full(shape, float(value), dtype=torch.float32)
    return init


# ---- the default store (what `from utils.ops import *` style code uses implicitly, like TF's default graph) -------------
_DEFAULT = [None]


def default_store():
    if _DEFAULT[0] is None:
        _DEFAULT[0] = VariableStore()
    return _DEFAULT[0]


def set_default_store(store):
    _DEFAULT[0] = store
    return store


def variable_scope(name, reuse=False):
    return default_store().variable_scope(name, reuse)


def trainable_variables(prefix):
    return default_store().trainable_variables(prefix)
