"""bf16 STORAGE (ABI v6, BASELINE config 3 end to end): activation tensors are bf16 in HBM between the kernels.

Kernel level — every dtype-aware entry point, called through the ctypes wrappers of t2i_amd.kernels, must give on bf16 tensors
EXACTLY what it gives on the same values held in fp32 tensors, rounded to bf16 (round to nearest even) where the output is an
activation and bit for bit where it is an fp32 reduction: the arithmetic is the fp32 path's, only loads widen and stores round.
That holds for the bf16-operand GEMMs too (a bf16 tensor is its own operand image; the products and the accumulation order are
those of the fp32-tensor call) and for the paths that run on fp32 staging copies (thin / head kernels).

Step level — the wgancls B = 64 iteration with bf16 tensors end to end against the float64 oracle, mask-pinned like
tests/test_step_b64_gpu.py::test_config3_bf16_steps_mask_pinned (and the all-bf16 side rows beside it), with their own stated tolerances; and hipGraph replay == eager."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def K():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import t2i_amd  # noqa: F401
    from t2i_amd import kernels as K
    K.set_math('bf16')
    yield K
    K.set_storage('f32')
    K.set_math('f32')


def bf(t):
    return t.bfloat16()


def same(got, want_f32):
    """got: bf16 tensor from the storage path; want_f32: the fp32-tensor path's result -> equal after one RNE rounding"""
    assert got.dtype == torch.bfloat16 and want_f32.dtype == torch.float32
    assert torch.equal(got, want_f32.bfloat16()), float((got.float() - want_f32).abs().max())


def test_elementwise_and_reductions_on_bf16_tensors(K):
    g = torch.Generator(device='cuda').manual_seed(0)
    shape = (8, 8, 8, 128)
    a = bf(torch.randn(shape, generator=g, device='cuda')); b = bf(torch.randn(shape, generator=g, device='cuda'))
    af, bfl = a.float(), b.float()
    sc = torch.randn(128, generator=g, device='cuda'); sh = torch.randn(128, generator=g, device='cuda')
    same(K.act_fwd(a, K.ACT_LRELU, 0.2), K.act_fwd(af, K.ACT_LRELU, 0.2))
    same(K.act_bwd(a, b, K.ACT_LRELU, 0.2), K.act_bwd(af, bfl, K.ACT_LRELU, 0.2))
    same(K.add_act(a, b, K.ACT_RELU), K.add_act(af, bfl, K.ACT_RELU))
    same(K.axpby(a, 0.25, b, -1.5), K.axpby(af, 0.25, bfl, -1.5))
    same(K.bn_apply(a, sc, sh, K.ACT_RELU), K.bn_apply(af, sc, sh, K.ACT_RELU))
    # fused activation backward + bias gradient: dx rounded, the column sums are those of the UNROUNDED dx (fp32 accumulators)
    dx, s = K.act_bwd_colsum(a, b, K.ACT_LRELU, 0.2)
    dxf, sf = K.act_bwd_colsum(af, bfl, K.ACT_LRELU, 0.2)
    same(dx, dxf); assert s.dtype == torch.float32 and torch.equal(s, sf)
    # column reductions and batch statistics: fp32 results, bit for bit
    s0, s1 = K.col_reduce(a, b, True)
    f0, f1 = K.col_reduce(af, bfl, True)
    assert torch.equal(s0, f0) and torch.equal(s1, f1)
    gamma = torch.rand(128, generator=g, device='cuda') + 0.5; beta = torch.randn(128, generator=g, device='cuda')
    st_h = K.bn_train_stats(a, gamma, beta, 1e-5, 0.9)
    st_f = K.bn_train_stats(af, gamma, beta, 1e-5, 0.9)
    assert all(torch.equal(x, y) for x, y in zip(st_h, st_f))
    mean, rstd = st_f[0], st_f[1]
    dxh, dg_h, db_h = K.bn_bwd_fused(a, b, a, mean, rstd, gamma, K.ACT_RELU)
    dxf, dg_f, db_f = K.bn_bwd_fused(af, bfl, af, mean, rstd, gamma, K.ACT_RELU)
    # the masked gradient travels as a bf16 buffer between the first and the last launch (one more rounding than the fp32-tensor
    # path): dgamma / dbeta come from the unrounded values and are exact, dx is within one bf16 ulp of the fp32-tensor result
    assert torch.equal(dg_h, dg_f) and torch.equal(db_h, db_f)
    assert dxh.dtype == torch.bfloat16 and float((dxh.float() - dxf).abs().max()) <= 2.0 ** -7 * float(dxf.abs().max())
    dxh2, _, _ = K.bn_bwd_fused(a, None, a, mean, rstd, gamma, K.ACT_NONE)
    dxf2, _, _ = K.bn_bwd_fused(af, None, af, mean, rstd, gamma, K.ACT_NONE)
    same(dxh2, dxf2)                                  # no activation: no intermediate buffer, exactly the rounded fp32 result
    # layout kernels: pure copies
    feat = bf(torch.randn(4, 4, 4, 1024, generator=g, device='cuda')); emb = bf(torch.randn(4, 128, generator=g, device='cuda'))
    cat = K.concat_tile_fwd(feat, emb)
    assert cat.dtype == torch.bfloat16 and torch.equal(cat.float(), K.concat_tile_fwd(feat.float(), emb.float()))
    dfeat, demb = K.concat_tile_bwd(cat, 1024, 128)
    dfeat_f, demb_f = K.concat_tile_bwd(cat.float(), 1024, 128)
    assert torch.equal(dfeat.float(), dfeat_f)
    same(demb, demb_f)
    x = bf(torch.randn(4, 64, 4, 4, generator=g, device='cuda'))
    assert torch.equal(K.nchw_to_nhwc(x).float(), K.nchw_to_nhwc(x.float()))
    assert torch.equal(K.nhwc_to_nchw(K.nchw_to_nhwc(x)), x)
    gg = bf(torch.randn(8, 1024, generator=g, device='cuda'))
    assert torch.equal(K.gp_slopes(gg), K.gp_slopes(gg.float()))
    coef = torch.randn(8, generator=g, device='cuda')
    same(K.row_scale(gg, coef), K.row_scale(gg.float(), coef))
    assert torch.equal(K.cast_f32(a), af)
    # a bf16 tensor on a scalar (unaligned) path is refused, not silently mis-read
    from t2i_amd._lib import T2IError
    with pytest.raises(T2IError):
        K.act_fwd(a.reshape(-1)[1:-3], K.ACT_LRELU, 0.2)


@pytest.mark.parametrize('splitk', [0, 4])
def test_convs_on_bf16_tensors_equal_the_rounded_fp32_tensor_path(K, splitk):
    g = torch.Generator(device='cuda').manual_seed(1)
    B, H, C, Co = 8, 8, 128, 256
    x = bf(torch.randn(B, H, H, C, generator=g, device='cuda'))
    w = torch.randn(3, 3, C, Co, generator=g, device='cuda') * 0.05
    bias = torch.randn(Co, generator=g, device='cuda')
    dy = bf(torch.randn(B, H, H, Co, generator=g, device='cuda'))
    K.tuning_set('force_splitk', splitk)
    try:
        d, ws = K.conv_desc(B, H, H, C, Co, 3, 3, 1, 1, 'SAME')
        ws = max(ws, 64 << 20)
        assert K.conv_algo(d, 'fwd') == 'implicit_gemm_bf16_operands'
        K.set_storage('f32')
        yf = K.conv_fwd(x.float(), w, bias, d, ws, K.ACT_LRELU, 0.2)
        dxf = K.conv_bwd_data(dy.float(), w, None, d, ws)
        dwf = K.conv_bwd_filter(x.float(), dy.float(), d, ws)
        K.set_storage('bf16')
        yh = K.conv_fwd(x, w, bias, d, ws, K.ACT_LRELU, 0.2)
        same(yh, yf)
        same(K.conv_bwd_data(dy, w, None, d, ws), dxf)
        assert torch.equal(K.conv_bwd_filter(x, dy, d, ws), dwf)                      # fp32 filter gradient, bit for bit
        # mixed: fp32 tensor in (a model input), bf16 out; bf16 in, fp32 out asked for by the caller (a gradient of an fp32 tensor)
        same(K.conv_fwd(x.float(), w, bias, d, ws, K.ACT_LRELU, 0.2), yf)
        assert torch.equal(K.conv_bwd_data(dy, w, None, d, ws, out_dtype=torch.float32), dxf)
    finally:
        K.tuning_set('force_splitk', 0)


@pytest.mark.parametrize('case', [
    (8, 16, 16, 256, 512, 4, 2, 0, 22, 'D3-like k4s2: input gradient in 4 stride phases + 128x128 filter-gradient tiles'),
    (16, 8, 8, 512, 512, 3, 1, 0, 22, '3x3 stride 1'),
    (16, 8, 8, 512, 512, 3, 1, 4, 22, '3x3, both GEMMs split-K by 4: two reductions behind the one launch'),
    (64, 4, 4, 1152, 1024, 3, 1, 0, 0, 'critic 4x4 map, K = 10368, the planner\'s own tiles (fused iff it takes 128x128 for the filter gradient)'),
    (64, 16, 16, 256, 256, 3, 1, 0, 0, 'generator 16x16 layer at the benchmark batch, the planner\'s own tiles'),
    (4, 8, 8, 192, 256, 3, 1, 0, 0, 'Cin = 192: the filter gradient is not the 128x128 DMA kernel -> the pair falls back to two launches'),
    (3, 8, 8, 128, 136, 3, 1, 0, 22, 'ragged N (136) and M')])
def test_bwd_pair_is_bit_identical_to_the_two_calls(K, case):
    """t2i_conv2d_bwd_pair (round 4): a layer's input gradient (PAIR_BWD_DATA; PAIR_FWD for the layer behind a transposed conv) and its
    sunk filter gradient in one launch.  Same tiles and arithmetic as the two entry points it replaces: every output bit for bit,
    including the accumulate-into-arena form; the fused launch is confirmed by the library's counter."""
    from t2i_amd._lib import lib
    B, H, W, Ci, Co, k, s, splitk, tile, _ = case
    g = torch.Generator(device='cuda').manual_seed(B * 7 + Ci)
    K.set_storage('bf16')
    K.tuning_set('force_splitk', splitk)
    K.tuning_set('force_tile', tile)
    # for the bit-for-bit comparison every layer may share a launch and both GEMMs keep the plan of their own entry point
    # (by default only small maps are fused, each GEMM planned for half the chip: another split-K grouping of the same sums)
    K.tuning_set('pair_max_px', 1 << 30)
    K.tuning_set('pair_cus', 256)
    try:
        d, ws = K.conv_desc(B, H, W, Ci, Co, k, k, s, s, 'SAME')
        ws = max(ws, 128 << 20)
        x = bf(torch.randn(B, H, W, Ci, generator=g, device='cuda'))
        dy = bf(torch.randn(B, d.Ho, d.Wo, Co, generator=g, device='cuda'))
        w = torch.randn(k, k, Ci, Co, generator=g, device='cuda') * 0.05
        arena0 = torch.randn(k * k * Ci * Co, generator=g, device='cuda')
        # conv backward: dx = conv^T(dy, w), dw += x (*) dy
        ref_dx = K.conv_bwd_data(dy, w, None, d, ws, out_dtype=torch.bfloat16)
        ref_dw = K.conv_bwd_filter(x, dy, d, ws, out=arena0.clone())
        n0 = lib.t2i_stat(b'pair_fused')
        got_dw = arena0.clone()
        got_dx = K.conv_bwd_pair(K.PAIR_BWD_DATA, dy, w, x, dy, d, ws, got_dw, out_dtype=torch.bfloat16)
        fused = lib.t2i_stat(b'pair_fused') - n0
        assert fused == 1 if (tile == 22 and Ci % 128 == 0 and Co % 64 == 0) else fused in ((0,) if Ci % 128 else (0, 1)), (fused, case)
        print('  %s: fused = %d' % (case[-1], fused))
        assert got_dx.dtype == torch.bfloat16 and torch.equal(got_dx, ref_dx)
        assert torch.equal(got_dw, ref_dw)
        # transposed-conv backward: g_dy = conv(gx, w), dw += gx (*) dy   (the incoming gradient has the conv's INPUT shape)
        if s == 1 or True:
            gx = bf(torch.randn(B, H, W, Ci, generator=g, device='cuda'))
            ref_y = K.conv_fwd(gx, w, None, d, ws, out_dtype=torch.bfloat16)
            ref_dw2 = K.conv_bwd_filter(gx, dy, d, ws, out=arena0.clone())
            got_dw2 = arena0.clone()
            got_y = K.conv_bwd_pair(K.PAIR_FWD, gx, w, gx, dy, d, ws, got_dw2, out_dtype=torch.bfloat16)
            assert torch.equal(got_y, ref_y) and torch.equal(got_dw2, ref_dw2)
        # the switch: two launches, same bits
        K.tuning_set('pair', 0)
        n1 = lib.t2i_stat(b'pair_fused')
        off_dw = arena0.clone()
        off_dx = K.conv_bwd_pair(K.PAIR_BWD_DATA, dy, w, x, dy, d, ws, off_dw, out_dtype=torch.bfloat16)
        assert lib.t2i_stat(b'pair_fused') == n1 and torch.equal(off_dx, ref_dx) and torch.equal(off_dw, ref_dw)
        # the default planning of a shared launch (half the chip each): same sums in another split-K grouping — fp32 rounding apart
        K.tuning_set('pair', 1); K.tuning_set('pair_cus', 128)
        if not splitk:
            K.tuning_set('force_tile', 0)
            d_dw = arena0.clone()
            d_dx = K.conv_bwd_pair(K.PAIR_BWD_DATA, dy, w, x, dy, d, ws, d_dw, out_dtype=torch.bfloat16)
            K.tuning_set('force_tile', 0)
            r_dx = K.conv_bwd_data(dy, w, None, d, ws, out_dtype=torch.bfloat16).float()
            r_dw = K.conv_bwd_filter(x, dy, d, ws, out=arena0.clone())
            assert float((d_dx.float() - r_dx).abs().max()) <= 2.0 ** -7 * float(r_dx.abs().max())          # one bf16 ulp of the largest element
            assert float((d_dw - r_dw).abs().max()) <= 1e-5 * float(r_dw.abs().max())
    finally:
        K.tuning_set('pair_cus', 128)
        K.tuning_set('pair_max_px', 49152)
        K.tuning_set('pair', 1)
        K.tuning_set('force_splitk', 0)
        K.tuning_set('force_tile', 0)
        K.set_storage('f32')


def test_conv_epilogue_batch_norm_statistics_on_bf16_tensors(K):
    """Round 4: the bf16-operand forward GEMM leaves the batch norm the per-tile column sums / centred second moments of the bf16
    tensor it writes (tile_stats_h) whenever the launch is unsplit; merged by t2i_bn_stats_tiles they equal the moments of the stored
    tensor (fp32 summation-order accuracy), for every tile shape incl. ragged M (bf16 tensors have multiples of 64 channels); a split launch declines."""
    g = torch.Generator(device='cuda').manual_seed(5)
    K.set_storage('bf16')
    K.tuning_set('h_stats', 1)          # opt-in since its A/B (profiles/r04_bf16_tile8.txt): the separate reduce is as fast
    try:
        for tile, (B, H, W, Ci, Co, k) in ((22, (4, 16, 16, 64, 192, 3)), (21, (3, 8, 8, 128, 64, 3)), (12, (2, 32, 32, 64, 128, 1)), (11, (5, 4, 4, 64, 192, 3)),
                                           (42, (5, 16, 16, 64, 128, 3)), (0, (64, 16, 16, 256, 256, 3))):
            K.tuning_set('force_tile', tile); K.tuning_set('force_splitk', 1 if tile else 0)
            d, ws = K.conv_desc(B, H, W, Ci, Co, k, k, 1, 1, 'SAME')
            x = bf(torch.randn(B, H, W, Ci, generator=g, device='cuda'))
            w = torch.randn(k, k, Ci, Co, generator=g, device='cuda') / (k * Ci ** 0.5)
            b = torch.randn(Co, generator=g, device='cuda')
            y = K.conv_fwd_stats(x, w, b, d, 256 << 20, K.ACT_NONE)
            assert y.dtype == torch.bfloat16
            got = K.take_stats(y)
            assert got is not None or tile == 0, (tile, 'a forced unsplit launch must leave statistics')
            assert torch.equal(y, K.conv_fwd(x, w, b, d, 256 << 20, K.ACT_NONE))
            if got is not None:
                ref = y.double().reshape(-1, Co)
                scale = float((ref * ref).sum(0).sqrt().max())
                assert float((got[0].double() - ref.sum(0)).abs().max()) <= 1e-5 * scale * ref.shape[0] ** 0.5, tile
                m2 = ((ref - ref.mean(0, keepdim=True)) ** 2).sum(0)
                assert float((got[1].double() - m2).abs().max()) <= 1e-5 * float(m2.max()), tile
        K.tuning_set('force_tile', 0); K.tuning_set('force_splitk', 3)
        d, ws = K.conv_desc(2, 4, 4, 512, 256, 3, 3, 1, 1, 'SAME')
        y = K.conv_fwd_stats(bf(torch.randn(2, 4, 4, 512, generator=g, device='cuda')), torch.randn(3, 3, 512, 256, generator=g, device='cuda') * 0.02, None, d, 256 << 20)
        assert K.take_stats(y) is None
    finally:
        K.tuning_set('force_tile', 0); K.tuning_set('force_splitk', 0); K.tuning_set('h_stats', 0)
        K.set_storage('f32')


def test_boundary_layers_with_bf16_tensors(K):
    """3 -> 128 stem (fp32 image in, bf16 out, native), its filter gradient and the 128 -> 3 transposed conv (bf16 in, staged), the
    logit head (bf16 in, fp32 out and back)."""
    g = torch.Generator(device='cuda').manual_seed(2)
    B = 4
    img = torch.rand(B, 64, 64, 3, generator=g, device='cuda') * 2 - 1
    w1 = torch.randn(4, 4, 3, 128, generator=g, device='cuda') / 7
    b1 = torch.randn(128, generator=g, device='cuda')
    d1, ws1 = K.conv_desc(B, 64, 64, 3, 128, 4, 4, 2, 2, 'SAME')
    K.set_storage('f32')
    yf = K.conv_fwd(img, w1, b1, d1, ws1, K.ACT_LRELU, 0.2)
    gy = bf(torch.randn(yf.shape, generator=g, device='cuda'))
    dwf = K.conv_bwd_filter(img, gy.float(), d1, ws1)
    dimg_f = K.conv_bwd_data(gy.float(), w1, None, d1, ws1)
    K.set_storage('bf16')
    yh = K.conv_fwd(img, w1, b1, d1, ws1, K.ACT_LRELU, 0.2)
    same(yh, yf)
    assert torch.equal(K.conv_bwd_filter(img, gy, d1, ws1), dwf)
    dimg = K.conv_bwd_data(gy, w1, None, d1, ws1)
    assert dimg.dtype == torch.float32 and torch.equal(dimg, dimg_f)            # 3 channels: an fp32 tensor in every mode
    feat = bf(torch.randn(B, 4, 4, 1024, generator=g, device='cuda'))
    wh = torch.randn(4, 4, 1024, 1, generator=g, device='cuda') * 0.01
    dh, wsh = K.conv_desc(B, 4, 4, 1024, 1, 4, 4, 4, 4, 'VALID')
    K.set_storage('f32')
    lf = K.conv_fwd(feat.float(), wh, None, dh, wsh)
    seed = torch.randn(B, 1, 1, 1, generator=g, device='cuda')
    dfeat_f = K.conv_bwd_data(seed, wh, None, dh, wsh)
    dwh_f = K.conv_bwd_filter(feat.float(), seed, dh, wsh)
    K.set_storage('bf16')
    lh = K.conv_fwd(feat, wh, None, dh, wsh)
    assert lh.dtype == torch.float32 and torch.equal(lh, lf)
    same(K.conv_bwd_data(seed, wh, None, dh, wsh), dfeat_f)
    assert torch.equal(K.conv_bwd_filter(feat, seed, dh, wsh), dwh_f)


def _tiny_cfg(B):
    from t2i_amd.utils.config import AttrDict
    return AttrDict({'MODEL': {'Z_DIM': 128, 'OUTPUT_SIZE': 64, 'EMBED_DIM': 1024, 'COMPRESSED_EMBED_DIM': 128, 'GF_DIM': 64, 'DF_DIM': 64,
                               'IMAGE_SHAPE': {'W': 64, 'H': 64, 'D': 3}},
                     'TRAIN': {'BATCH_SIZE': B, 'SAMPLE_NUM': 4, 'D_LR': 1e-4, 'G_LR': 1e-4, 'BETA1': 0.0, 'BETA2': 0.9, 'N_CRITIC': 1,
                               'SUMMARY_PERIOD': 10, 'MAX_STEPS': 10, 'COEFF': {'KL': 1.0, 'LAMBDA': 100.0}}})


def test_storage_iteration_graph_replay_matches_eager_and_uses_bf16_tensors(K):
    import bench
    from t2i_amd.models.wgancls.model import WGanCls
    from t2i_amd.models.wgancls.trainer import WGanClsTrainer
    K.set_storage('bf16')
    B = 8
    cfg = _tiny_cfg(B)
    dev = torch.device('cuda')
    feeds = [bench.synthetic_feed(cfg, dev, seed=10 + i) for i in range(4)]
    states = []
    for use_graphs in (False, True):
        m = WGanCls(cfg, device=dev, seed=0)
        tr = WGanClsTrainer(None, m, None, cfg)
        dtypes = []
        if not use_graphs:
            orig = K.conv_fwd

            def tapped(*a, **k):
                y = orig(*a, **k)
                dtypes.append((tuple(y.shape), y.dtype))
                return y
            K.conv_fwd = tapped
        try:
            tr.iteration(1, feeds[0])
        finally:
            if not use_graphs:
                K.conv_fwd = orig
                # every conv output with a multiple of 64 channels is a bf16 tensor, except the two [B, 128] conditioning heads
                assert all(dt == torch.bfloat16 for sh, dt in dtypes if sh[-1] % 64 == 0 and sh[-1] != 128)
                assert sum(dt == torch.bfloat16 for _, dt in dtypes) >= 20 and any(dt == torch.float32 for _, dt in dtypes)
        if use_graphs:
            m.enable_graphs(feeds[0])
        outs = [tr.iteration(2 + i, feeds[1 + i]) for i in range(3)]
        torch.cuda.synchronize()
        states.append((m.d_arena.flat.clone(), m.g_arena.flat.clone(), float(outs[-1]['d']['D_loss']), float(outs[-1]['g']['G_loss'])))
        m._graphs = None
    (d0, g0, ld0, lg0), (d1, g1, ld1, lg1) = states
    assert torch.equal(d0, d1) and torch.equal(g0, g1) and ld0 == ld1 and lg0 == lg1
    assert np.isfinite(ld0) and np.isfinite(lg0)
