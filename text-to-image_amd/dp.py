"""Data parallelism for the wgancls step: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference is single-device; replicas are new functionality: every rank holds identical weights,
kt and Adam state, sees its own slice of the global batch, and gradients are averaged before the optimizer — so N
replicas at local batch b behave as the reference at BATCH_SIZE = N*b with per-replica batch-norm statistics.

Exchange step: the gradient arena (optim.Arena.grad, one flat buffer per optimizer) is cut into contiguous buckets in
REVERSE creation order (the order backward produces them).  A post-accumulate hook counts finished parameters; when a
bucket is complete its all-reduce is issued on a side HIP stream while the main stream keeps running the remaining
backward kernels.  xGMI is point-to-point (7 links x ~153 GB/s), ring all-reduce is per-link bound, so buckets are
large (default 32 MB) — few, big collectives.  The sum is turned into a mean inside the Adam kernel (grad_scale).
"""
import torch
import torch.distributed as dist


class DataParallel(object):
    def __init__(self, bucket_bytes=32 << 20, process_group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed must be initialised (init_process_group) before DataParallel')
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._arenas = {}
        self._side = None

    # ---- bucket plan ---------------------------------------------------------------------------------------------------
    def _plan(self, arena):
        """Contiguous [start, end) ranges of the arena, walking parameters last-created-first."""
        buckets, cur_names, cur_end, cur_start = [], [], None, None
        for name in reversed(arena.names):
            off, n = arena.offsets[name]
            end = off + (n + 3) // 4 * 4
            if cur_end is None:
                cur_end = end
            cur_start = off
            cur_names.append(name)
            if cur_end - cur_start >= self.bucket_elems:
                buckets.append((cur_start, cur_end, cur_names))
                cur_names, cur_end = [], None
        if cur_names:
            buckets.append((cur_start, cur_end, cur_names))
        return buckets

    def attach(self, arena):
        """Install the overlap hooks on an arena's parameters (idempotent)."""
        key = id(arena)
        if key in self._arenas:
            return self._arenas[key]
        st = {'buckets': self._plan(arena), 'pending': None, 'works': [], 'armed': False, 'arena': arena}
        owner = {}
        for bi, (_, _, names) in enumerate(st['buckets']):
            for n in names:
                owner[n] = bi
        for n, v in arena.vars.items():
            v.register_post_accumulate_grad_hook(self._make_hook(st, owner[n]))
        self._arenas[key] = st
        return st

    def _make_hook(self, st, bi):
        def hook(_param):
            if not st['armed']:
                return
            st['pending'][bi] -= 1
            if st['pending'][bi] == 0:
                self._launch(st, bi)
        return hook

    def arm(self, arena):
        """Call right before backward: the hooks of this arena start counting."""
        st = self.attach(arena)
        st['pending'] = [len(names) for _, _, names in st['buckets']]
        st['works'] = []
        st['armed'] = True

    def _launch(self, st, bi):
        start, end, _ = st['buckets'][bi]
        buf = st['arena'].grad[start:end]
        if buf.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=buf.device)
            self._side.wait_stream(torch.cuda.current_stream(buf.device))   # gradients of this bucket are final
            with torch.cuda.stream(self._side):
                st['works'].append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            st['works'].append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def allreduce_arena(self, arena, extra=None):
        """Finish the exchange for `arena` (launch whatever the hooks did not, wait) and return the factor that turns the
        summed gradients into the mean.  `extra`: a small tensor (the kt gradient) summed in place alongside."""
        st = self.attach(arena)
        if st['armed']:
            for bi, left in enumerate(st['pending']):
                if left > 0:                       # parameter unused this step: bucket never completed by hooks
                    self._launch(st, bi)
        else:                                      # hooks were not armed: plain bucketed all-reduce after backward
            st['works'] = []
            for bi in range(len(st['buckets'])):
                self._launch(st, bi)
        if extra is not None:
            st['works'].append(dist.all_reduce(extra, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in st['works']:
            w.wait()
        if self._side is not None and arena.grad.is_cuda:
            torch.cuda.current_stream(arena.grad.device).wait_stream(self._side)
        st['armed'] = False
        st['works'] = []
        return 1.0 / self.world

    def broadcast_variables(self, store, src=0):
        """Make every rank start from rank `src`'s variables (weights, BN moving statistics)."""
        for v in store.vars.values():
            dist.broadcast(v.data, src=src, group=self.group)
