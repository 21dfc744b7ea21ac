#!/usr/bin/env python
"""HBM-bound kernels of libt2i_hip.so against the 8 TB/s roofline: algorithmic bytes / measured time (HIP events on the
launch stream, 50 back-to-back launches after 5 warm-ups).  Sizes are the benchmark's (B=64 wgancls tensors, the Adam
arenas, a 256-image data batch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import t2i_amd  # noqa: E402,F401
from t2i_amd import kernels as K  # noqa: E402

PEAK = 8000.0  # GB/s, MI355X_MICROARCH.md


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


def row(name, nbytes, fn):
    t = timeit(fn)
    print('%-44s %9.1f MB %9.1f us %8.1f GB/s  %5.1f%% of 8 TB/s' % (name, nbytes / 1e6, t * 1e6, nbytes / t / 1e9, 100 * nbytes / t / 1e9 / PEAK))


def main():
    dev = 'cuda'
    B = 64
    x = torch.randn(B, 32, 32, 256, device=dev); y = torch.randn_like(x)          # 67 MB activation (G8 / D2 level)
    n = x.numel()
    C = 256
    sc, sh = torch.randn(C, device=dev), torch.randn(C, device=dev)
    row('act_fwd lrelu (r+w)', 8 * n, lambda: K.act_fwd(x, K.ACT_LRELU, 0.2))
    row('act_bwd lrelu (2r+w)', 12 * n, lambda: K.act_bwd(x, y, K.ACT_LRELU, 0.2))
    row('add_act relu (2r+w)', 12 * n, lambda: K.add_act(x, y, K.ACT_RELU, 0.0))
    row('act_bwd_colsum lrelu (2r+w, +bias grad)', 12 * n, lambda: K.act_bwd_colsum(x, y, K.ACT_LRELU, 0.2))
    row('col_reduce sum,sumsq (r)', 4 * n, lambda: K.col_reduce(x, None, True))
    row('bn_apply relu (r+w)', 8 * n, lambda: K.bn_apply(x, sc, sh, K.ACT_RELU, 0.0))
    mean, rstd = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
    s1, s2 = K.col_reduce(x, y, True)
    row('bn_bwd apply (2r+w)', 12 * n, lambda: K.bn_bwd(x, y, mean, rstd, sc, s1, s2))
    g = torch.randn(B, 64, 64, 3, device=dev)
    row('gp_slopes 64x64x3 (r)', 4 * g.numel(), lambda: K.gp_slopes(g))
    eps = torch.rand(B, 1, 1, 1, device=dev)
    row('interp x_hat (2r+w)', 12 * g.numel(), lambda: K.interp(eps, g, g))
    feat, emb = torch.randn(3 * B, 4, 4, 1024, device=dev), torch.randn(3 * B, 128, device=dev)
    row('concat_tile 4x4x(1024+128) (r+w)', 4 * (feat.numel() + 3 * B * 16 * 1152), lambda: K.concat_tile_fwd(feat, emb))
    N = 28995329                                                                  # critic arena
    w, gr, m, v = (torch.randn(N + 3, device=dev)[:N] for _ in range(4))
    w, gr, m, v = (torch.zeros((N + 3) // 4 * 4, device=dev) for _ in range(4))
    v.uniform_()
    row('adam_tf critic arena, beta1 = 0 (w, g, v read; w, m, v written)', 24 * w.numel(), lambda: K.adam_tf(w, gr, m, v, 1e-4, 0.0, 0.9, 1e-8, 1.0))
    row('adam_tf critic arena, beta1 = 0, m skipped (3r+2w): the product path', 20 * w.numel(), lambda: K.adam_tf(w, gr, None, v, 1e-4, 0.0, 0.9, 1e-8, 1.0))
    # PGGAN operators
    row('pool2 avg (r + w/4)', 5 * n, lambda: K.pool2_sum(x, 0.25))
    xs = torch.randn(B, 16, 16, 256, device=dev)
    row('upscale2 nearest (r/4 + w)', 5 * n, lambda: K.upscale2(xs, 1.0))
    row('row_moments (r)', 4 * n, lambda: K.row_moments(x))
    a, d = torch.rand(B, device=dev), torch.rand(B, device=dev)
    row('row_fma2 normalise (r+w)', 8 * n, lambda: K.row_fma2(x, a, delta=d))
    # data pipeline: a 256-image batch out of a 8192-image 76x76 store
    store = torch.randint(0, 256, (8192, 76, 76, 3), dtype=torch.uint8, device=dev)
    Bd = 256
    ids = torch.randint(0, 8192, (Bd,), dtype=torch.int32, device=dev)
    r0 = torch.randint(0, 12, (Bd,), dtype=torch.int32, device=dev); c0 = torch.randint(0, 12, (Bd,), dtype=torch.int32, device=dev)
    fl = torch.randint(0, 2, (Bd,), dtype=torch.int32, device=dev)
    row('crop_flip_normalize 76->64 (3 B r + 12 B w)', Bd * 64 * 64 * 15, lambda: K.crop_flip_normalize(store, ids, r0, c0, fl, 64))
    emb5 = torch.randn(8192, 5, 1024, device=dev)
    ch = torch.stack([torch.randperm(5, device=dev)[:4] for _ in range(Bd)]).to(torch.int32)
    row('gather_mean 4 of 5 captions (4r + w)', Bd * 1024 * 20, lambda: K.gather_mean(emb5, ids, ch))


if __name__ == '__main__':
    main()
