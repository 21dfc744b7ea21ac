"""Flat parameter arenas + tf.train.AdamOptimizer semantics (reference models/wgancls/model.py:94-106).

All trainable variables of one optimizer live back-to-back in ONE device buffer (16-byte aligned slots), with
gradients, first and second moments in three more buffers of the same shape.  A step is then a single kernel launch
over ~29 M (critic) / ~23 M (generator) floats instead of one launch per tensor, and data-parallel gradient exchange
works on contiguous slices of the gradient arena (dp.py)."""
import math
import os
from collections import OrderedDict

import torch

from . import kernels as K


class Arena(object):
    def __init__(self, variables):
        """variables: OrderedDict name -> leaf tensor.  Re-points every variable's storage (and .grad) into the arena."""
        self.names = list(variables.keys())
        self.vars = variables
        self.offsets = OrderedDict()
        off = 0
        for n, v in variables.items():
            self.offsets[n] = (off, v.numel())
            off += (v.numel() + 3) // 4 * 4          # 16-byte slots: the kernels' float4 paths need aligned bases
        self.numel = off
        dev = next(iter(variables.values())).device
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for n, v in variables.items():
                o, k = self.offsets[n]
                self.flat[o:o + k].copy_(v.reshape(-1))
                v.data = self.flat[o:o + k].view(v.shape)
                v.grad = self.grad[o:o + k].view(v.shape)
        K.filter_cache_invalidate()                  # the variables moved (possibly onto recycled addresses)

    def zero_grad(self):
        # an optimizer that derives its first moment from this arena on demand (AdamTF with beta1 == 0) is about to lose its source:
        # let it form the moment first.  (Not inside a capture: a replayed iteration leaves the last step's gradients behind, so the
        # on-demand path stays valid between replays, and the host does not run here on replay anyway.)
        if self.before_zero and not (self.grad.is_cuda and torch.cuda.is_current_stream_capturing()):
            dead = False
            for ref in self.before_zero:           # weak references to the optimizers: a discarded optimizer is not kept alive by its arena
                opt = ref()
                if opt is None:
                    dead = True
                else:
                    opt._materialize_m()
            if dead:
                self.before_zero = tuple(r for r in self.before_zero if r() is not None)
        if self._store_first is not None:
            # the large filter slots are not zeroed: their first contribution of the step is a plain store (first_touch); one launch zeroes
            # the small slots between them
            K.zero_ranges(self.grad, self._small_ranges)
            self._touched = set()
            self._pending_check = True
        else:
            self.grad.zero_()

    before_zero = ()
    _store_first = None
    _pending_check = False

    def enable_sinks(self, store_first=None):
        """Let the filter-gradient GEMMs accumulate directly into this arena (autograd.SINKS).
        store_first (default: T2I_STORE_FIRST != 0): the slots of the large filters (>= 2^16 elements; conv / deconv / dense kernels) are
        never zero-filled — the first contribution a step writes into such a slot is a plain store (accumulate = 0 in the filter-gradient
        epilogue), later ones add.  Saves the fill (116 + 91 MB per wgancls iteration) and the epilogues' read of the slot.  A slot that
        receives NO contribution in a step would keep the previous step's gradient: finish_step() — called by the optimizer before it
        reads the arena — zeroes exactly those."""
        from . import autograd as A
        if store_first is None:
            store_first = os.environ.get('T2I_STORE_FIRST', '1') != '0'
        big = set()
        self._store_first, self._pending_check = None, False          # (a repeated call re-decides)
        if store_first and self.grad.is_cuda:
            big = {n for n, v in self.vars.items() if v.numel() >= (1 << 16) and v.dim() >= 2}
        import weakref
        me = weakref.ref(self)             # (the sink registry must not keep the arena — and through it the parameters — alive)

        def first_touch(name):
            a = me()
            return a._first_touch(name) if a is not None else False
        for n, v in self.vars.items():     # kernels, biases, BN gamma/beta: every gradient is summed in place by its kernel
            A.register_sink(v, self.grad_of(n).view(-1), (lambda name=n: first_touch(name)) if n in big else None)
        if big:
            self._store_first = big
            self._touched = set(big)           # until the first zero_grad every slot holds zeros already: accumulate
            runs, start = [], None
            for n in self.names:               # maximal runs of small slots (padding included), in arena order
                o, k = self.offsets[n]
                end = o + (k + 3) // 4 * 4
                if n in big:
                    if start is not None:
                        runs.append((start, o - start)); start = None
                else:
                    if start is None:
                        start = o
                    last_end = end
            if start is not None:
                runs.append((start, self.numel - start))
            self._small_ranges = torch.tensor(runs if runs else [[0, 0]], dtype=torch.int64, device=self.grad.device).reshape(-1, 2)

    def _first_touch(self, name):
        if name in self._touched:
            return False
        self._touched.add(name)
        return True

    def finish_step(self):
        """Before the optimizer (or a gradient exchange) reads the arena: large slots that received no contribution since zero_grad still
        hold the previous step's gradient — zero them (none on the models here: every filter gets its gradient every step)."""
        if self._store_first is not None and self._pending_check:
            for n in self._store_first - self._touched:
                self.grad_of(n).zero_()
                self._touched.add(n)
            self._pending_check = False

    def grad_of(self, name):
        o, k = self.offsets[name]
        return self.grad[o:o + k].view(self.vars[name].shape)


class AdamTF(object):
    """lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v updates; w -= lr_t*m/(sqrt(v)+eps) — epsilon OUTSIDE the bias correction
    (SURVEY.md §8a M5).  With beta1=0, t=1 this is ~lr*sign(g)."""

    def __init__(self, arena, beta1=0.9, beta2=0.999, eps=1e-8):
        self.arena, self.beta1, self.beta2, self.eps = arena, beta1, beta2, eps
        self._m = torch.zeros_like(arena.flat)
        self.v = torch.zeros_like(arena.flat)
        # beta1 == 0 (both wgancls optimizers, PGGAN): m_t = g_t * grad_scale whatever m_{t-1} was, so the step neither reads nor
        # writes it (4 bytes per parameter less of the 24 the update streams); `m` is formed from the gradient arena when somebody
        # asks for it — a checkpoint between two iterations — which is valid until the arena is zeroed for the next backward
        self.skip_m = beta1 == 0.0 and os.environ.get('T2I_ADAM_SKIP_M', '1') != '0'
        # An EAGER Arena.zero_grad after a step forms the lagging moment before it clears its source (one read + one write of the
        # arena: 8 bytes per parameter, more than the 4 the fast path saves on that iteration; replayed graphs never run it).  A
        # loop that checkpoints only right after a step — every trainer here — may switch it off: `m` stays valid until the next
        # zero_grad either way.  T2I_ADAM_KEEP_M=0 sets the default.
        self.keep_m_across_zero_grad = os.environ.get('T2I_ADAM_KEEP_M', '1') != '0'
        self._m_stale, self._last_scale = False, 1.0    # `_m` lags the last step (it is re-formed from the gradient arena on demand)
        self.t = 0
        self.lr_t_dev = torch.zeros(4, dtype=torch.float32, device=arena.flat.device)   # [0] = this step's lr_t
        if self.skip_m:
            import weakref
            arena.before_zero = tuple(arena.before_zero) + (weakref.ref(self),)

    @property
    def m(self):
        """First moment.  With the beta1 == 0 fast path it is (re)built from the gradient arena on access (see __init__): the arena
        holds the last step's gradients until the next zero_grad, and an EAGER Arena.zero_grad forms a stale `m` before it clears
        them (_materialize_m), so a checkpoint taken after a stray zero_grad / backward still holds the last step's moment."""
        if self.skip_m and self._m_stale:
            with torch.no_grad():
                torch.mul(self.arena.grad, self._last_scale, out=self._m)
            self._m_stale = False
        return self._m

    @m.setter
    def m(self, value):
        """Assigning the first moment (a checkpoint restore, a test) keeps the assigned values until the next step."""
        with torch.no_grad():
            self._m.copy_(value)
        self._m_stale = False

    def _materialize_m(self):
        """Arena.zero_grad (eager) is about to clear the gradients the on-demand first moment is formed from."""
        if self.skip_m and self._m_stale and self.keep_m_across_zero_grad:
            self.m

    def moments_loaded(self):
        """The caller has just written m (and t) from a checkpoint: keep those values until the next step."""
        self._m_stale = False

    def prepare(self, lr):
        """Host half of a step: advance t and publish lr_t to the device scalar (outside any captured graph)."""
        self.t += 1
        self._m_stale = self.skip_m          # every step (eager or replayed: this is its host half) leaves `_m` behind
        lr_t = lr * math.sqrt(1.0 - self.beta2 ** self.t) / (1.0 - self.beta1 ** self.t)
        self.lr_t_dev.fill_(lr_t)
        return lr_t

    def apply(self, grad_scale=1.0, refresh=None):
        """Device half: one kernel over the arena, step size read from the device scalar (graph-capturable).
        refresh: regenerate the cached filter images of this arena behind the update, in one launch
        (kernels.filter_cache_refresh).  Default: yes for eager launches; not inside a capture, whose successor graph starts
        with a refresh of everything — pass True where the same capture goes on to use these filters (the critic's update in
        a one-graph iteration)."""
        self.arena.finish_step()
        K.adam_tf(self.arena.flat, self.arena.grad, None if self.skip_m else self._m, self.v, 0.0, self.beta1, self.beta2, self.eps, grad_scale,
                  lr_t_dev=self.lr_t_dev)
        if self.skip_m:
            self._last_scale = float(grad_scale)
            self._m_stale = True             # (an eager zero_grad between prepare() and here has formed the PREVIOUS step's moment)
        if refresh is None:
            refresh = self.arena.flat.is_cuda and not torch.cuda.is_current_stream_capturing()
        if refresh and self.arena.flat.is_cuda:
            K.filter_cache_refresh(self.arena.flat)

    def step(self, lr, grad_scale=1.0):
        self.prepare(lr)
        self.apply(grad_scale)
