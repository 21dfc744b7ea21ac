"""hipGraph capture of training-step bodies (HIP graphs instead of a tracing compiler).

A step body is a function of a dict of device tensors that issues only device work (kernels of libt2i_hip.so, the
Adam launch with its step size in device memory, tensor-library scalar math) — no host synchronisation, no allocation
that survives the call, no host-side random draws.  `StepGraphs` keeps one set of static input buffers, captures each
body once and replays it: at the reference's batch sizes (8 or 16) a GAN iteration is ~1000 launches of a few
microseconds each and is bound by the host's launch rate, not by the GPU.  Replay is bit-identical to the eager
launches (same kernels, same order)."""
import torch

from . import kernels as K


def capture_mode(requested='global'):
    """The capture_error_mode a capture in this process should use.  As soon as a process group exists, its watchdog thread
    polls events on the HIP runtime at any time; in 'global' mode such a call from another thread while a capture is open is an
    error that tears the process down (seen once in the full GPU suite: a single-GPU capture after a data-parallel test in the same
    process).  'thread_local' confines the checks to the capturing thread, which is all these captures need."""
    if requested == 'global':
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return 'thread_local'
        except Exception:          # noqa: BLE001 — a build without torch.distributed has no watchdog either
            pass
    return requested


class StepGraphs(object):
    def __init__(self, example_feed, keys, filters=()):
        """example_feed: name -> device tensor (shapes/dtypes of every later feed); keys: the entries the bodies read;
        filters: the tensors holding the convolution filters the bodies use (the optimizer arenas) — see capture(refresh=)."""
        self.filters = list(filters)
        self.static = {k: example_feed[k].clone() for k in keys if example_feed.get(k) is not None}
        self.graphs, self.outs, self._pool = {}, {}, None

    def load(self, feed):
        """Copy this iteration's inputs into the static buffers (device-to-device, outside the graphs)."""
        for k, buf in self.static.items():
            v = feed.get(k)
            if v is not None and v is not buf:
                buf.copy_(v, non_blocking=True)

    def capture(self, name, body, capture_error_mode='global', refresh=True):
        """body(static_feed) -> anything holding device tensors (kept alive and returned by replay).
        capture_error_mode='thread_local' when another thread touches the HIP runtime during the capture (the process
        group's watchdog under data parallelism).  refresh: start the graph with ONE batched regeneration of every cached filter
        image (kernels.filter_cache_refresh) — a graph must contain every transform it depends on, and filled lazily they are
        one small launch per filter; pass False for a segment that runs no convolution."""
        dev = next(iter(self.static.values())).device
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self._pool, capture_error_mode=capture_mode(capture_error_mode)):
            if refresh:
                for t in self.filters:               # one batched launch per arena; the convs of this graph then find every
                    K.filter_cache_refresh(t)        # known image of these filters filled
            out = body(self.static)
        if self._pool is None:
            self._pool = g.pool()
        self.graphs[name], self.outs[name] = g, out
        return out

    def replay(self, name):
        self.graphs[name].replay()
        K.filter_cache_invalidate()                  # a replayed optimizer step rewrote filters behind the host's back
        return self.outs[name]
