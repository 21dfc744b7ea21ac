#!/usr/bin/env python
"""Does a cross-stream wait issued from autograd's worker thread still order work after a graph capture?

Single process, no torch.distributed.  A custom Function's backward (run by autograd's device worker thread) enqueues a long
chain of kernels that finish a buffer on the current (null) stream, then makes a side stream wait for the current stream and
copies the buffer to pinned host memory on the side stream — the pattern dp._launch uses for an early bucket.  The copy
must see the final values.  Run before and after capturing an unrelated autograd backward into a hipGraph
(thread-local capture mode), as WGanCls.enable_graphs does under data parallelism.  usage: repro_worker_thread_wait.py [hop]   (hop: the copy runs on a THIRD stream that waits for an event recorded
on the empty side stream — the shape of a collective issued under `with torch.cuda.stream(side)`)"""
import threading

import torch

dev = torch.device('cuda', 0)
import sys
HOP = len(sys.argv) > 1 and sys.argv[1] == 'hop'
side = torch.cuda.Stream()
third = torch.cuda.Stream(priority=-1)
N = 1 << 24
buf = torch.zeros(N, device=dev)
host = torch.empty(N, pin_memory=True)
big = torch.randn(4096, 4096, device=dev)
result = {}


class Probe(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        buf.zero_()
        t = big
        for _ in range(30):                  # ~tens of ms of queued work in front of the final write
            t = t @ big * 1e-3
        buf.add_(t.sum() * 0 + 1.0)          # the "last gradient contribution": buf becomes all ones
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        if HOP:                              # what a collective does: record on the (otherwise empty) side stream, copy on a third
            ev2 = torch.cuda.Event(); ev2.record(side)
            third.wait_event(ev2)
            with torch.cuda.stream(third):
                host.copy_(buf, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(third)
        else:
            with torch.cuda.stream(side):
                host.copy_(buf, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(side)
        result['ev'] = ev
        result['thread'] = threading.current_thread().name
        result['stream'] = hex(cur.cuda_stream)
        return g


def probe(tag):
    x = torch.ones(4, device=dev, requires_grad=True)
    Probe.apply(x).sum().backward()
    result['ev'].synchronize()
    bad = int((host != 1.0).sum())
    torch.cuda.synchronize()
    print('%-28s thread %-10s stream %s  stale elements in the copy: %d of %d' % (tag, result['thread'], result['stream'], bad, N))
    return bad


probe('before any capture')
probe('before any capture (again)')
# an unrelated autograd step captured the way enable_graphs does it under data parallelism
w = torch.randn(512, 512, device=dev, requires_grad=True)
inp = torch.randn(64, 512, device=dev)
(inp @ w).sum().backward()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    w.grad = None
    (inp @ w).sum().backward()
torch.cuda.synchronize()
bad = probe('after a thread-local capture')
bad += probe('after a thread-local capture (again)')
g.replay(); torch.cuda.synchronize()
bad += probe('after a replay')
print('RESULT: %s' % ('cross-stream wait from the worker thread lost ordering after the capture' if bad else 'ordering held'))
