"""Data parallelism for the wgancls step: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference is single-device; replicas are new functionality: every rank holds identical weights,
kt and Adam state, sees its own slice of the global batch, and gradients are averaged before the optimizer — so N
replicas at local batch b behave as the reference at BATCH_SIZE = N*b with per-replica batch-norm statistics.  The kt
step is not a gradient average: balance_loss = (kt*wdist2 - wdist)^2 is quadratic in two batch means, so the ranks
exchange those means (summed next to the arena as `extra`) and each evaluates the gradient of the GLOBAL-batch loss
(t2i_kt_sgd).  tests/test_dp_gloo.py[kt] checks the exchange of the means and the arithmetic on distinct per-rank data (in Python
floats, not through the kernel); tests/test_kernels_gpu.py checks t2i_kt_sgd itself; the two-rank runs of
tests/test_dp_exactness_gpu.py go through WGanCls._d_update on identical data.

Exchange step: the gradient arena (optim.Arena.grad, one flat buffer per optimizer) is cut into contiguous buckets in
REVERSE creation order (the order backward produces them).  Finished parameters are counted — by a post-accumulate
hook for gradients that go through autograd's AccumulateGrad, by autograd.NOTIFY for gradients that the kernels sum
straight into the arena (gradient sinks: no tensor gradient reaches AccumulateGrad for them — although the engine still
VISITS a leaf named in backward(inputs=...) with an undefined gradient, so the hook only counts visits whose incoming
gradient is defined; how many contributions each parameter gets per backward is learned on the first armed step, which
therefore exchanges after the backward).  When a bucket is
complete its all-reduce is issued on a side HIP stream while the main stream keeps running the remaining backward
kernels.  xGMI is point-to-point (7 links x ~153 GB/s), ring all-reduce is per-link bound, so buckets are
large (default 32 MB) — few, big collectives.  The sum is turned into a mean inside the Adam kernel (grad_scale).
"""
import os

import torch
import torch.distributed as dist

from .utils import roctx as _roctx


class DataParallel(object):
    def __init__(self, bucket_bytes=32 << 20, process_group=None, grad_dtype='f32', f32_exchange=None):
        """f32_exchange (grad_dtype 'f32' only): 'allreduce' (default) = one library all-reduce per bucket, whatever algorithm RCCL
        picks (a ring is per-link bound: SURVEY section 5 prices the 116 MB critic arena at ~1.33 ms on 8 GPUs); 'rs_ag' = the
        explicit reduce-scatter + all-gather over the same bucket, in place (every rank reduces ONE chunk with all of its 7 links
        busy, then the chunks are gathered: ~0.19 ms by the same arithmetic) — the fp32 mirror of _exchange_bf16.  Selected by the
        argument or T2I_DP_F32_EXCHANGE=rs_ag, so that an 8-GPU run can measure one against the other (bench.py reports which
        ran in `gradient_exchange.f32_exchange`).  Each chunk's sum is formed by ONE rank and gathered, so all ranks leave with the
        same bits, as with the library all-reduce; the summation ORDER may differ from the library's, so for N > 2 the two
        forms agree to rounding, not bit for bit (N = 2: a + b is commutative, bit-identical — tests/test_dp_gloo.py[rs_ag]).
        grad_dtype: 'f32' (exchange the fp32 arena in place) or 'bf16' (BASELINE config 3: bf16 on the wire — half the bytes
        on every xGMI link — with the SUM taken in fp32: a bucket is rounded to bf16 once (RNE), the ranks exchange chunks
        (all-to-all: rank j receives everybody's j-th chunk), each sums the N chunks it received in fp32, rounds that sum to
        bf16 once and the sums are all-gathered; the fp32 arena receives the result.  A ring all-reduce in bf16 would round the
        running sum at every one of its N-1 hops instead.  Same bytes per link as the ring: 2 (N-1)/N of the bf16 bucket.
        For N a power of two, N identical contributions come back EXACTLY (N x is representable whenever x is), which is what
        bench.dp_preflight demands.  The optimizer and its moments stay fp32)."""
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed must be initialised (init_process_group) before DataParallel')
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._arenas = {}
        self._by_ptr = {}
        self._side = None
        if grad_dtype not in ('f32', 'bf16'):
            raise ValueError("grad_dtype must be 'f32' or 'bf16', got %r" % (grad_dtype,))
        self.grad_dtype = grad_dtype
        f32_exchange = f32_exchange or os.environ.get('T2I_DP_F32_EXCHANGE', 'allreduce')
        if f32_exchange not in ('allreduce', 'rs_ag'):
            raise ValueError("f32_exchange must be 'allreduce' or 'rs_ag', got %r" % (f32_exchange,))
        self.f32_exchange = f32_exchange
        self._stage = {}                      # id(arena) -> (send, recv, [offset]) bf16 staging buffers
        # bf16 buckets: all-to-all + all-gather (RCCL) or the all-gather-everything fallback (gloo, test transports).  Decided
        # ONCE per device type from the backend's name — never by catching an exception around a collective: an OOM or a transient
        # RCCL error on one rank would otherwise switch that rank alone to a collective its peers are not in.
        self._a2a = {}
        # early bucket launches during the backward; T2I_DP_NO_OVERLAP=1 (or overlap = False) exchanges after it instead
        self.overlap = os.environ.get('T2I_DP_NO_OVERLAP') != '1'
        self._stats = None                    # begin_stats(): per-exchange HIP events + byte counts (bench.py's gradient_exchange block)

    # ---- measurement of the exchange itself ------------------------------------------------------------------------------
    def begin_stats(self):
        """From now on every collective is bracketed by events on the communication stream and every finish_allreduce by events on
        the calling stream (the time that stream stalls = the exposed part of the exchange).  Costs a host wait per collective in
        fp32 mode (the async work is waited for on the communication stream so that the closing event sees its end): use on a few
        extra iterations, not inside a timed region."""
        self._stats = {'payload_bytes': 0, 'collectives': 0, 'comm': [], 'stall': [], 'exchanges': 0}

    def end_stats(self, steps=1):
        """-> dict (per step): payload bytes, collectives, ms inside collectives (communication stream), ms the compute stream stalled
        in finish_allreduce, overlap fraction = 1 - stalled / in-collectives."""
        st, self._stats = self._stats, None
        if st is None:
            return None
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        comm = sum(a.elapsed_time(b) for a, b in st['comm'])
        stall = sum(a.elapsed_time(b) for a, b in st['stall'])
        n = max(1, steps)
        ring = 2.0 * (self.world - 1) / max(self.world, 1)
        return {'dtype_on_wire': self.grad_dtype, 'f32_exchange': self.f32_exchange if self.grad_dtype == 'f32' else None, 'ranks': self.world, 'payload_bytes_per_step': st['payload_bytes'] / n,
                'bytes_sent_per_rank_per_step': st['payload_bytes'] * ring / n, 'collectives_per_step': st['collectives'] / float(n),
                'exchanges_per_step': st['exchanges'] / float(n), 'ms_in_collectives_per_step': comm / n,
                'ms_compute_stream_stalled_per_step': stall / n,
                'overlap_fraction': (1.0 - stall / comm) if comm > 0 else None,
                'algorithm_bandwidth_GBps': (st['payload_bytes'] / 1e9) / (comm * 1e-3) if comm > 0 else None,
                'steps_measured': n}

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
        st = {'buckets': self._plan(arena), 'pending': None, 'works': [], 'armed': False, 'arena': arena,
              'expect': None, 'seen': {}, 'owner_ptr': {}, 'launched': set(), 'defined': set()}
        owner = {}
        for bi, (_, _, names) in enumerate(st['buckets']):
            for n in names:
                owner[n] = bi
        for n, v in arena.vars.items():
            # The engine visits a leaf named in backward(inputs=...) even when every path handed it an UNDEFINED gradient
            # (all its contributions were summed into sinks): its hooks then fire with nothing accumulated.  The tensor
            # pre-hook sees the incoming gradient and tells the post-accumulate hook whether anything real arrived.
            v.register_hook(self._make_pre_hook(st, v.data_ptr()))
            v.register_post_accumulate_grad_hook(self._make_hook(st, owner[n]))
            st['owner_ptr'][v.data_ptr()] = owner[n]
            self._by_ptr[v.data_ptr()] = st
        self._arenas[key] = st
        from . import autograd as A
        A.NOTIFY[0] = self.notify
        return st

    def notify(self, ptr):
        """autograd.NOTIFY target: one more contribution to the parameter at `ptr` has been issued into its sink."""
        st = self._by_ptr.get(ptr)
        if st is None or not st['armed']:
            return
        c = st['seen'].get(ptr, 0) + 1
        st['seen'][ptr] = c
        exp = st['expect']
        if exp is None:
            return                                  # first armed step: only learning the counts
        want = exp.get(ptr, 0)
        if c == want:
            bi = st['owner_ptr'][ptr]
            st['pending'][bi] -= 1
            if st['pending'][bi] == 0:
                self._launch(st, bi)
        elif c > want:
            raise RuntimeError('data-parallel overlap: parameter received %d sunk gradient contributions, %d were learned '
                               'on the first step (its bucket may already be in flight); the backward structure changed' %
                               (c, want))

    def _make_pre_hook(self, st, ptr):
        def pre(g):
            if st['armed'] and g is not None:
                st['defined'].add(ptr)
            return g
        return pre

    def _make_hook(self, st, bi):
        def hook(param):
            if not st['armed']:
                return
            ptr = param.data_ptr()
            if ptr not in st['defined']:
                return                              # visited with an undefined gradient: nothing was accumulated
            st['defined'].discard(ptr)
            from . import autograd as A
            if A.sink_at(ptr) is not None:          # a sunk parameter that ALSO received a tensor gradient: one more contribution
                self.notify(ptr)
                return
            st['pending'][bi] -= 1                  # ordinary leaf: exactly one AccumulateGrad per backward
            if st['pending'][bi] == 0:
                self._launch(st, bi)
        return hook

    def arm(self, arena):
        """Call right before backward: the hooks of this arena start counting."""
        st = self.attach(arena)
        if not self.overlap:                 # exchange after the backward (allreduce_arena launches every bucket then)
            return
        st['pending'] = [len(names) for _, _, names in st['buckets']]
        if id(arena) in self._stage:
            self._stage[id(arena)][2][0] = 0     # a new exchange: its buckets take fresh slices of the staging buffers
        st['works'] = []
        st['seen'] = {}
        st['launched'] = set()
        st['defined'] = set()
        st['armed'] = True

    def _launch(self, st, bi):
        start, end, _ = st['buckets'][bi]
        if bi in st['launched']:
            return
        st['launched'].add(bi)
        if st['armed'] and os.environ.get('T2I_DP_SNAPSHOT') == '1' and not st.get('in_start'):
            # diagnostics: instead of exchanging now, remember the bucket as it is at its "last contribution" ...
            torch.cuda.synchronize()
            st['launched'].discard(bi)
            st.setdefault('snap', {})[bi] = st['arena'].grad[start:end].clone()
            return
        self._launch_range(st, start, end)

    def _stage_buffers(self, st, n):
        """Two bf16 staging buffers per arena, each holding the arena padded to world x (chunk of a multiple of 8 elements)."""
        key = id(st['arena'])
        bufs = self._stage.get(key)
        if bufs is None:
            cap = -(-st['arena'].numel // (8 * self.world)) * 8 * self.world + 8 * self.world * len(st['buckets'])
            dev = st['arena'].grad.device
            bufs = self._stage[key] = (torch.zeros(cap, dtype=torch.bfloat16, device=dev), torch.zeros(cap, dtype=torch.bfloat16, device=dev), [0])
        return bufs

    def _exchange_bf16(self, st, buf):
        """bf16 on the wire, fp32 accumulation (see __init__): all-to-all of chunks -> local fp32 sum -> all-gather of the sums.
        Runs on the calling stream (the communication stream); every rank leaves with the same bits in `buf`."""
        n, N = buf.numel(), self.world
        chunk = -(-n // (8 * N)) * 8
        send_all, recv_all, used = self._stage_buffers(st, n)
        if used[0] + N * chunk > send_all.numel():      # buckets of one exchange use disjoint slices (they may be in flight together)
            used[0] = 0
        send = send_all[used[0]:used[0] + N * chunk]
        recv = recv_all[used[0]:used[0] + N * chunk]
        used[0] += N * chunk
        send[:n].copy_(buf)                             # fp32 -> bf16, round to nearest even
        if N * chunk > n:
            send[n:].zero_()
        if N == 1:
            buf.copy_(send[:n])
            return
        if self._has_all_to_all(buf):
            dist.all_to_all_single(recv, send, group=self.group)            # recv[j] = rank j's chunk number `rank`
            mine = recv.view(N, chunk).float().sum(0).to(torch.bfloat16)   # the sum in fp32, rounded once
            dist.all_gather_into_tensor(send, mine, group=self.group)       # send now holds every rank's reduced chunk
            buf.copy_(send[:n])
            return
        # same arithmetic without all-to-all: everybody gathers everybody's bf16 bucket and sums all of it in fp32 (N times the
        # bytes — test transports only; RCCL takes the branch above)
        parts = [torch.empty_like(send) for _ in range(N)]
        dist.all_gather(parts, send, group=self.group)
        buf.copy_(torch.stack(parts).float().sum(0).to(torch.bfloat16)[:n])

    def _exchange_f32_rs_ag(self, st, buf):
        """fp32 reduce-scatter + all-gather, in place on `buf` (see __init__).  Runs on the calling stream (the communication
        stream).  The bucket's first N * c elements (c = n // N) go through the two collectives — rank r's output chunk is the
        view buf[r*c:(r+1)*c] of the input itself, the in-place form RCCL supports — and the < N leftover elements through one
        tiny all-reduce.  Test transports without reduce_scatter_tensor (gloo) gather every rank's bucket and sum the ranks in rank
        order: the same bits on every rank there too."""
        n, N = buf.numel(), self.world
        if not self._has_all_to_all(buf):
            if N == 1:
                return
            parts = [torch.empty_like(buf) for _ in range(N)]
            dist.all_gather(parts, buf, group=self.group)
            acc = parts[0].clone()
            for p_ in parts[1:]:
                acc.add_(p_)
            buf.copy_(acc)
            return
        c = n // N
        if c > 0:
            main = buf[:N * c]
            mine = main[self.rank * c:(self.rank + 1) * c]
            dist.reduce_scatter_tensor(mine, main, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(main, mine, group=self.group)
        if n > N * c:
            dist.all_reduce(buf[N * c:], op=dist.ReduceOp.SUM, group=self.group)

    def _has_all_to_all(self, buf):
        """Does the process group's backend for this tensor's device have all_to_all_single / all_gather_into_tensor?  RCCL
        ('nccl') does; gloo does not (neither on device tensors — the 2-ranks-on-1-GPU pre-flight — nor, for all_to_all, on the
        CPU).  A pure function of (backend name, device type), so every rank decides alike; the choice is logged once."""
        kind = buf.device.type
        got = self._a2a.get(kind)
        if got is None:
            backend = str(dist.get_backend(self.group)).lower()
            config = str(dist.get_backend_config(self.group)).lower() if hasattr(dist, 'get_backend_config') else backend
            if ':' in config:                             # "cpu:gloo,cuda:nccl": the entry of this device type
                per = dict(item.split(':', 1) for item in config.split(',') if ':' in item)
                backend = per.get('cuda' if kind == 'cuda' else 'cpu', backend)
            got = self._a2a[kind] = backend in ('nccl', 'rccl')
            if not got and self.rank == 0 and os.environ.get('T2I_QUIET') != '1':
                print('[t2i dp] bf16 buckets on backend %r (%s tensors): no all-to-all, gathering whole buckets instead '
                      '(test transport; N x the bytes)' % (backend, kind), flush=True)
        return got

    def _launch_range(self, st, start, end):
        if not _roctx.available():                 # (no marker library / T2I_ROCTX=0: not even the range's name is formatted)
            return self._launch_range_impl(st, start, end)
        with _roctx.range('dp.exchange %s[%d:%d] %s' % (self._arena_tag(st), start, end, self.grad_dtype)):
            self._launch_range_impl(st, start, end)

    def _arena_tag(self, st):
        names = st['arena'].names
        return (names[0].split('/')[0] if names else 'arena')

    def _launch_range_impl(self, st, start, end):
        buf = st['arena'].grad[start:end]
        bf16 = self.grad_dtype == 'bf16'
        rs_ag = not bf16 and self.f32_exchange == 'rs_ag'
        if buf.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=buf.device)
            self._side.wait_stream(torch.cuda.current_stream(buf.device))   # gradients of this bucket are final
            from . import autograd as A
            if A.SIDE.stream is not None:                                   # ... including the filter-gradient stream's
                self._side.wait_stream(A.SIDE.stream)
            with torch.cuda.stream(self._side):
                stats = self._stats
                if stats is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                if rs_ag:            # everything ordered on the communication stream, like the bf16 form
                    self._exchange_f32_rs_ag(st, buf)
                elif not bf16:
                    w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    if stats is not None:
                        w.wait()         # the communication stream waits for the collective: the closing event sees its end
                    else:
                        st['works'].append(w)
                else:                # everything ordered on the communication stream
                    self._exchange_bf16(st, buf)
                if stats is not None:
                    e1.record()
                    stats['comm'].append((e0, e1))
                    stats['payload_bytes'] += buf.numel() * (2 if bf16 else 4)
                    stats['collectives'] += 2 if (bf16 or rs_ag) else 1
        elif rs_ag:
            self._exchange_f32_rs_ag(st, buf)
        elif not bf16:
            st['works'].append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._exchange_bf16(st, buf)

    def allreduce_arena(self, arena, extra=None):
        """Finish the exchange for `arena` (launch whatever the hooks did not, wait) and return the factor that turns the
        summed gradients into the mean.  `extra`: a small tensor (the kt gradient) summed in place alongside."""
        self.start_allreduce(arena, extra)
        return self.finish_allreduce(arena)

    def start_allreduce(self, arena, extra=None, ranges=None):
        """Issue every all-reduce of `arena` that is not in flight yet (communication stream; the calling stream is not
        blocked).  Work enqueued on the calling stream between this and finish_allreduce overlaps the exchange — it must not
        touch the arena's gradients.  ranges (graph-segment schedule only, i.e. not armed): [(start, end), ...] element ranges
        of the arena whose gradients are final NOW — the part of a backward that has been cut in two; the caller issues the
        remaining ranges with another call before finish_allreduce."""
        st = self.attach(arena)
        if hasattr(arena, 'finish_step'):
            arena.finish_step()                    # (filter slots no contribution was stored into this step: zeroed before anything reads them)
        if ranges is not None and st['armed']:
            raise RuntimeError('start_allreduce(ranges=...) belongs to the cut (un-armed) schedule: this arena is armed for the '
                               'bucket-overlap schedule, which would exchange every bucket now — including the ones the cut says '
                               'are not final yet')
        if st.get('snap'):                      # ... and compare with what the backward finally left there
            torch.cuda.synchronize()
            for bi, snap in st['snap'].items():
                b0, b1, names = st['buckets'][bi]
                cur = arena.grad[b0:b1]
                if not torch.equal(cur, snap):
                    late = [n for n in names if not torch.equal(arena.grad_of(n).reshape(-1), snap[arena.offsets[n][0] - b0:arena.offsets[n][0] - b0 + arena.offsets[n][1]])]
                    raise RuntimeError('bucket %d changed after its last announced contribution: %s' % (bi, late[:20]))
            st['snap'] = {}
        st['in_start'] = True
        if st['armed']:
            if st['seen'] and st['expect'] is None:
                st['expect'] = dict(st['seen'])    # contributions per sunk parameter, fixed by the model's structure
            elif st['expect'] is not None and st['seen'] and st['seen'] != st['expect']:      # (a step without sunk gradients: nothing to compare)
                names = {v.data_ptr(): n for n, v in arena.vars.items()}
                diff = {names.get(p, hex(p)): (st['seen'].get(p, 0), st['expect'].get(p, 0))
                        for p in set(st['seen']) | set(st['expect']) if st['seen'].get(p, 0) != st['expect'].get(p, 0)}
                raise RuntimeError('data-parallel overlap: sunk gradient contributions (seen, learned) differ from the first '
                                   'armed step for %s; a bucket may have been exchanged before it was complete' % diff)
            for bi in range(len(st['buckets'])):   # whatever hooks / notifications did not complete (unused parameters,
                self._launch(st, bi)               # the learning step): launched now; _launch skips the ones in flight
        else:                                      # not armed (graph segments, exchange-after-backward): the gradients are
            if ranges is None or not st.get('partial'):
                st['works'] = []                   # all final, so ONE collective over the whole arena (or one per cut) — ring
            st['partial'] = ranges is not None     # all-reduce is per-link bound, fewer and larger is better
            st['launched'] = set(range(len(st['buckets'])))
            if ranges is None:
                if id(arena) in self._stage:
                    self._stage[id(arena)][2][0] = 0
                self._launch_range(st, 0, arena.numel)
            else:
                for a, b in ranges:
                    if b > a:
                        self._launch_range(st, a, b)
        st['in_start'] = False
        if extra is not None:
            if extra.is_cuda and self._side is not None:
                self._side.wait_stream(torch.cuda.current_stream(extra.device))
                with torch.cuda.stream(self._side):
                    st['works'].append(dist.all_reduce(extra, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                st['works'].append(dist.all_reduce(extra, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish_allreduce(self, arena):
        """Make the calling stream wait for the exchange started by start_allreduce; returns 1/world."""
        roctx = _roctx
        st = self.attach(arena)
        roctx.push('dp.finish_allreduce %s (compute stream waits for the exchange)' % self._arena_tag(st))
        stats = self._stats if arena.grad.is_cuda else None
        if stats is not None:
            m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            m0.record()
        for w in st['works']:
            w.wait()
        if self._side is not None and arena.grad.is_cuda:
            torch.cuda.current_stream(arena.grad.device).wait_stream(self._side)
        if stats is not None:
            m1.record()
            stats['stall'].append((m0, m1))
            stats['exchanges'] += 1
        roctx.pop()
        st['armed'] = False
        st['works'] = []
        st['partial'] = False
        if id(arena) in self._stage:
            self._stage[id(arena)][2][0] = 0
        return 1.0 / self.world

    def broadcast_variables(self, store, src=0):
        """Make every rank start from rank `src`'s variables (weights, BN moving statistics)."""
        for v in store.vars.values():
            dist.broadcast(v.data, src=src, group=self.group)
        from . import kernels as K
        K.filter_cache_invalidate()


class LocalRounding(object):
    """A world-of-one stand-in with the DataParallel interface whose "exchange" only rounds the gradient arena to bf16 and back:
    the single-replica REFERENCE of a bf16-bucket run.  N ranks (N a power of two) that contribute identical gradients x get back
    N * bf16(x) from DataParallel(grad_dtype='bf16') — exactly, the sum being taken in fp32 — and Adam scales by 1/N: bit for bit
    what this class leaves in the arena.  bench.dp_preflight demands that equality."""
    world, rank, grad_dtype = 1, 0, 'bf16'

    def arm(self, arena):
        pass

    def start_allreduce(self, arena, extra=None, ranges=None):
        if hasattr(arena, 'finish_step'):
            arena.finish_step()
        for a, b in (ranges if ranges is not None else [(0, arena.numel)]):
            if b > a:
                g = arena.grad[a:b]
                g.copy_(g.bfloat16())

    def finish_allreduce(self, arena):
        return 1.0

    def allreduce_arena(self, arena, extra=None):
        self.start_allreduce(arena, extra)
        return 1.0

    def broadcast_variables(self, store, src=0):
        pass
