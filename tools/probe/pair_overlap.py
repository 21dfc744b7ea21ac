#!/usr/bin/env python
"""Do two independent bf16 GEMM launches (the input gradient and the filter gradient of one layer) overlap when they sit on two
streams of a captured graph?  Times: (a) both on one stream, (b) forked onto two streams and joined, per pair, graph-replayed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import t2i_amd  # noqa: E402,F401
from t2i_amd import kernels as K  # noqa: E402
from tools.bench_conv import LAYERS  # noqa: E402

K.set_math('bf16'); K.set_storage('bf16')
dev = torch.device('cuda')
K.workspace(dev, 1 << 30)
side = torch.cuda.Stream()
K.stream_lane(side, dev)
L = {l[0]: l for l in LAYERS}


def graph_time(body, reps=10):
    for _ in range(2):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * reps) * 1e3


for name, B in (('D2', 64), ('D3', 64), ('D4', 64), ('D10', 64), ('G5c', 64), ('G7c', 64), ('G8c', 64), ('D3', 192)):
    _, H, W, Ci, Co, k, s, pad = L[name]
    d, ws = K.conv_desc(B, H, W, Ci, Co, k, k, s, s, pad)
    x = torch.randn(B, H, W, Ci, device=dev).bfloat16()
    w = torch.randn(k, k, Ci, Co, device=dev) * 0.05
    dy = torch.randn(B, d.Ho, d.Wo, Co, device=dev).bfloat16()
    dw = torch.zeros(k, k, Ci, Co, device=dev)

    def one():
        K.conv_bwd_data(dy, w, None, d, 1 << 30, out_dtype=torch.bfloat16)
        K.conv_bwd_filter(x, dy, d, 1 << 30, out=dw)

    def two():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            K.conv_bwd_filter(x, dy, d, 1 << 30, out=dw)
        K.conv_bwd_data(dy, w, None, d, 1 << 30, out_dtype=torch.bfloat16)
        cur.wait_stream(side)
    td = graph_time(lambda: K.conv_bwd_data(dy, w, None, d, 1 << 30, out_dtype=torch.bfloat16))
    tf = graph_time(lambda: K.conv_bwd_filter(x, dy, d, 1 << 30, out=dw))
    t1, t2 = graph_time(one), graph_time(two)
    print('%-4s B=%-3d bwd_data %6.1f  bwd_filter %6.1f  pair on one stream %6.1f  pair on two streams %6.1f us' % (name, B, td, tf, t1, t2))
