"""CPU tests of the host side: the C-ABI library loads and exports every symbol the header declares, the reference's
operator surface / variable naming / trainer schedule are reproduced, and CPU tensors are refused (no fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def t2i():
    lib = os.path.join(ROOT, 'text-to-image_amd', 'lib', 'libt2i_hip.so')
    if not os.path.exists(lib):
        import __graft_entry__ as ge
        ge.build()
    import t2i_amd
    return t2i_amd


def test_abi_exports_every_declared_symbol(t2i):
    from t2i_amd import _lib
    header = open(os.path.join(ROOT, 'include', 't2i_hip.h')).read()
    declared = set(re.findall(r'\b(t2i_[a-z0-9_]+)\s*\(', header))
    declared -= {'t2i_stream_t'}
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(_lib.lib, name), name
    assert _lib.lib.t2i_version() == _lib.ABI_VERSION
    assert _lib.lib.t2i_last_error() is not None


def test_descriptor_validation_without_gpu(t2i):
    from t2i_amd import _lib, kernels as K
    import ctypes
    bad = _lib.ConvDesc(1, 4, 4, 8, 0, 2, 8, 4, 4, 2, 2, 1, 1)      # Ho = 0
    assert _lib.lib.t2i_conv2d_workspace_bytes(ctypes.byref(bad)) == 0
    d, ws = K.conv_desc(64, 4, 4, 1152, 1024, 3, 3, 1, 1, 'same')
    assert (d.Ho, d.Wo, d.pad_t, d.pad_l) == (4, 4, 1, 1) and ws >= 0
    d, _ = K.conv_desc(1, 6, 6, 4, 4, 4, 4, 1, 1, 'SAME')
    assert (d.Ho, d.pad_t) == (6, 1)                                # k4s1 -> pads (1,2)
    d, _ = K.conv_desc(1, 5, 5, 4, 4, 2, 2, 1, 1, 'SAME')
    assert (d.Ho, d.pad_t) == (5, 0)                                # k2s1 -> pads (0,1)
    d, _ = K.deconv_desc(2, 4, 4, 1024, 512, 4, 4, 2, 2, 'SAME')
    assert (d.H, d.W, d.Cin, d.Cout, d.Ho, d.Wo) == (8, 8, 512, 1024, 4, 4)
    with pytest.raises(ValueError):
        K.conv_desc(1, 4, 4, 4, 4, 3, 3, 1, 1, 'FULL')
    # hostile extents: four 31-bit factors overflow a 64-bit element count (UBSan finding of tools/sanitize_host.sh) — refused,
    # with the size message, by all three queries
    big = 2 ** 31 - 1
    for desc in (_lib.ConvDesc(big, big, big, 4, big, big, 4, 3, 3, 1, 1, 1, 1), _lib.ConvDesc(1, 4, 4, big, 4, 4, big, big, big, 1, 1, 1, 1),
                 _lib.ConvDesc(1, 46341, 46341, 1024, 46341, 46341, 1024, 3, 3, 1, 1, 1, 1)):
        assert _lib.lib.t2i_conv2d_workspace_bytes(ctypes.byref(desc)) == 0
        assert _lib.lib.t2i_conv2d_algo(ctypes.byref(desc), 0) == -1
        assert b'2^30' in _lib.lib.t2i_last_error() or b'inconsistent' in _lib.lib.t2i_last_error()


def test_grouped_batch_norm_refuses_bad_tile_partials_and_counts(t2i):
    """t2i_bn_train_fwd_grouped / t2i_bn_bwd_grouped argument checks: an over-long tile_chunks (its last tiles would start
    beyond the group: n <= 0 in the merge, inf / NaN into the moving averages), a moving_updates outside [1, 8] (a device loop count)
    and element counts beyond the kernels' 32-bit offsets are refused BEFORE any launch — so this runs without a GPU, on fake
    (non-null, 16-byte aligned) addresses that are never dereferenced."""
    import ctypes
    from t2i_amd import _lib
    L = _lib.lib
    P = lambda k: ctypes.c_void_p(0x10000 + 64 * k)
    ws_n = max(int(L.t2i_bn_grouped_workspace_bytes(256, 64, 2)), 1 << 16)

    def fwd(rows=256, C=64, groups=2, tile_chunks=4, tile_rows=64, moving_updates=1, tiles=True):
        return L.t2i_bn_train_fwd_grouped(P(1), rows, C, groups, P(2), P(3), 1e-5, 0.9, P(4), P(5), P(6), P(7), P(8), P(9), 0, 0.2, P(10), None,
                                          P(11) if tiles else None, P(12) if tiles else None, tile_chunks, tile_rows, moving_updates, 0, P(13), ws_n, 0, None)
    assert fwd(tile_chunks=5) == -1 and b'tile partials' in L.t2i_last_error()       # 5 tiles of 64 rows for 256 rows
    assert fwd(tile_chunks=3) == -1                                                  # too few
    assert fwd(moving_updates=0) == -1 and b'moving_updates' in L.t2i_last_error()
    assert fwd(moving_updates=9) == -1
    assert fwd(rows=1 << 28, C=64, groups=1, tiles=False) == -1 and b'2^30' in L.t2i_last_error()
    assert fwd(rows=1 << 20, C=64, groups=32, tiles=False) == -1 and b'2^30' in L.t2i_last_error()
    rc = L.t2i_bn_bwd_grouped(P(1), None, P(2), P(3), P(4), P(5), 1 << 28, 64, 1, 0, 0.2, None, P(6), None, P(7), P(8), 0, P(9), ws_n, 0, None)
    assert rc == -1 and b'2^30' in L.t2i_last_error()


@pytest.mark.skipif(os.environ.get('T2I_TEST_SANITIZE') != '1', reason='two minutes of hipcc: run with T2I_TEST_SANITIZE=1 (output of the last run: profiles/r05_sanitize_host.txt)')
def test_host_side_under_asan_ubsan():
    """tools/sanitize_host.sh: the C ABI's host code built with AddressSanitizer + UndefinedBehaviorSanitizer and swept by
    tests/workers/san_host.c (queries over every layer geometry and hostile descriptors; invalid compute calls).  No GPU."""
    import subprocess
    r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'sanitize_host.sh')], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert r.returncode == 0 and 'no sanitizer report' in r.stdout, r.stdout[-3000:]


def test_no_cpu_fallback(t2i):
    from t2i_amd import kernels as K
    with pytest.raises(RuntimeError, match='no CPU path'):
        K.act_fwd(torch.zeros(4), K.ACT_RELU)
    with K.dry_run():
        assert K.act_fwd(torch.zeros(4), K.ACT_RELU).shape == (4,)


def _cfg(B=4):
    from t2i_amd.utils.config import config_from_yaml
    cfg = config_from_yaml(os.path.join(ROOT, 'text-to-image_amd', 'models', 'wgancls', 'cfg', 'flowers.yml'))
    cfg.TRAIN.BATCH_SIZE = B
    return cfg


def test_variable_registry_matches_tf_names(t2i):
    """TF auto-names, shapes, totals; arenas alias the variables."""
    from oracle import torch_step as T
    from t2i_amd.models.wgancls.model import WGanCls
    m = WGanCls(_cfg(), device='cpu')
    mine = {n: tuple(v.shape) for n, v in m.store.vars.items()}
    ref = {n: tuple(s) for n, (s, k, f) in T.variable_shapes(T.Cfg()).items()}
    assert mine == ref
    for sc in ('g_net/', 'd_net/'):                                 # creation order inside each scope too
        assert [n for n in mine if n.startswith(sc)] == [n for n in ref if n.startswith(sc)]
    assert sum(v.numel() for v in m.g_vars.values()) == 22643287
    assert sum(v.numel() for v in m.d_vars.values()) == 28995329
    assert all(not n.endswith('moving_mean') for n in m.g_vars)
    w = m.d_vars['d_net/Conv_3/weights']
    assert w.data_ptr() == m.d_arena.flat.data_ptr() + 4 * m.d_arena.offsets['d_net/Conv_3/weights'][0]
    assert w.grad.data_ptr() == m.d_arena.grad.data_ptr() + 4 * m.d_arena.offsets['d_net/Conv_3/weights'][0]
    assert all(off % 4 == 0 for off, _ in m.d_arena.offsets.values())
    # He init: std = sqrt(2.6/fan_in), truncated at 2 std; biases zero; gamma one
    std = float(w.std()); want = (2.6 / (4 * 4 * 512)) ** 0.5
    assert abs(std / (want * 0.8796) - 1) < 0.02                    # std of a 2-sigma truncated normal = 0.8796 sigma
    assert float(w.abs().max()) <= 2 * want * 1.0001
    assert float(m.store.vars['d_net/Conv_3/biases'].abs().max()) == 0.0
    assert float(m.store.vars['g_net/BatchNorm_4/gamma'].min()) == 1.0
    # the deconv fan-in quirk: kh*kw*Cout
    wt = m.g_vars['g_net/Conv2d_transpose/weights']
    assert tuple(wt.shape) == (4, 4, 512, 1024)
    assert abs(float(wt.std()) / ((2.6 / (16 * 512)) ** 0.5 * 0.8796) - 1) < 0.02


def test_scope_reuse_semantics(t2i):
    from t2i_amd import kernels as K, scope as S
    from t2i_amd.utils import ops
    st = S.set_default_store(S.VariableStore(device='cpu'))
    x = torch.zeros(2, 8, 8, 4)
    with K.dry_run():
        with S.variable_scope('net'):
            y = ops.conv2d(x, 8); ops.conv2d(y, 8, ks=(3, 3), s=(1, 1)); ops.fc(torch.zeros(2, 5), 3); ops.linear(torch.zeros(2, 5), 3)
        assert list(st.vars) == ['net/Conv/weights', 'net/Conv/biases', 'net/Conv_1/weights', 'net/Conv_1/biases',
                                 'net/dense/kernel', 'net/dense/bias', 'net/dense_1/kernel', 'net/dense_1/bias']
        with S.variable_scope('net', reuse=True):
            ops.conv2d(x, 8)                                     # counters restart: resolves to net/Conv again
        assert len(st.vars) == 8
        with pytest.raises(ValueError):
            with S.variable_scope('net'):                       # exists and reuse=False -> TF raises too
                ops.conv2d(x, 8)
        with pytest.raises(ValueError):
            with S.variable_scope('other', reuse=True):         # reuse of a variable that was never created
                ops.conv2d(x, 8)
        with pytest.raises(ValueError):
            ops.conv2d(x, 8, df='NCWH')
        # logical NCHW in == logical NCHW out, storage NHWC
        with S.variable_scope('fmt'):
            out = ops.conv2d(ops.to_nchw(x), 16, df=ops.NCHW)
        assert tuple(out.shape) == (2, 16, 4, 4) and ops.to_nhwc(out).is_contiguous()
    assert ops.deconv2d is ops.conv2d_transpose and ops.bn is ops.batch_norm


def test_trainer_schedule(t2i):
    from t2i_amd.models.wgancls.trainer import WGanClsTrainer

    class M(object):
        device = torch.device('cpu'); batch_size = 2; z_dim = 3
    cfg = _cfg()
    tr = WGanClsTrainer(None, M(), None, cfg)
    assert tr.lr_scale(1) == 1.0 and tr.lr_scale(9999) == 1.0
    assert abs(tr.lr_scale(10000) - 0.95) < 1e-12 and abs(tr.lr_scale(25000) - 0.95 ** 2) < 1e-12
    cfg.TRAIN.N_CRITIC = 5
    assert abs(tr.lr_scale(50000) - 0.95) < 1e-12                  # (idx // n_critic) // 10000


def test_config_keys_match_reference_yaml(t2i):
    cfg = _cfg()
    assert (cfg.MODEL.Z_DIM, cfg.MODEL.EMBED_DIM, cfg.MODEL.COMPRESSED_EMBED_DIM, cfg.MODEL.GF_DIM, cfg.MODEL.DF_DIM) == \
        (128, 1024, 128, 128, 128)
    assert (cfg.TRAIN.D_LR, cfg.TRAIN.BETA1, cfg.TRAIN.BETA2, cfg.TRAIN.N_CRITIC, cfg.TRAIN.COEFF.KL) == (1e-4, 0.0, 0.9, 1, 1.0)


def test_committed_bench_line_has_the_contract_keys():
    """The bench line kept with the profiles (profiles/r01_bench_line.json, written by bench.py on the GPU box) carries every
    key of the driver's contract, the roofline block and the CPU baseline block."""
    import json
    line = json.load(open(os.path.join(ROOT, 'profiles', 'r01_bench_line.json')))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line, k
    assert line['unit'] == 'images/sec' and line['dtype'] == 'f32' and line['higher_is_better'] is True and line['scaling'] == 'weak'
    assert line['vs_baseline'] is None and line['data'] == 'synthetic' and 'workload' in line['config'] and 'model' not in line['config']
    assert abs(line['value'] - 64 * line['n_gpus'] * 1e3 / line['ms_per_step']) <= 1e-6 * line['value']
    r = line['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['frac'] < 1
    c = line['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0


def test_conv_algorithm_choice_on_the_benchmark_layers():
    """t2i_conv2d_algo (host logic only): which algorithm the three conv entry points take at the benchmark's layer shapes,
    B = 64.  Pins the dispatch the roofline accounting and the Winograd parity tests rely on."""
    import t2i_amd  # noqa: F401
    from t2i_amd import kernels as K
    G, W3, W2, S = 'implicit_gemm', 'winograd_f2x2_3x3', 'winograd_f2x2_2x2', 'direct_small'
    want = {   # (H, W, Cin, Cout, k, stride, pad): (fwd, bwd_data, bwd_filter)
        (64, 64, 3, 128, 4, 2, 'SAME'): (S, S, S),            # critic layer 1 / generator out_deconv (as its adjoint conv): stem kernels (fwd, thin deconv, filter gradient)
        (32, 32, 128, 256, 4, 2, 'SAME'): (W2, W2, W2),       # since the persistent batched GEMM the per-phase transforms of dy pay at 128 channels too
        (16, 16, 256, 512, 4, 2, 'SAME'): (W2, W2, W2),
        (8, 8, 512, 1024, 4, 2, 'SAME'): (W2, W2, W2),
        (4, 4, 512, 1024, 3, 1, 'SAME'): (W3, W3, W3),
        (4, 4, 1152, 1024, 3, 1, 'SAME'): (W3, W3, W3),
        (8, 8, 512, 512, 3, 1, 'SAME'): (W3, W3, W3),
        (16, 16, 256, 256, 3, 1, 'SAME'): (W3, W3, W3),
        (32, 32, 128, 128, 3, 1, 'SAME'): (W3, W3, G),        # 32x32 maps pay since the persistent batched GEMM; 128 x 128 filters: direct filter gradient
        (8, 8, 128, 512, 3, 1, 'SAME'): (W3, W3, W3),
        (8, 8, 128, 128, 3, 1, 'SAME'): (G, G, G),            # too little work for 16 GEMMs at B = 64
        (4, 4, 1024, 256, 1, 1, 'SAME'): (G, G, G),
        (64, 64, 3, 3, 3, 1, 'SAME'): (S, S, S),
        (4, 4, 1024, 1, 4, 4, 'VALID'): (S, S, S),
    }
    for (H, W, Ci, Co, k, s, pad), algos in want.items():
        d, _ = K.conv_desc(64, H, W, Ci, Co, k, k, s, s, pad)
        assert tuple(K.conv_algo(d, m) for m in ('fwd', 'bwd_data', 'bwd_filter')) == algos, (H, W, Ci, Co, k, s)
    # round 5: the stride-2 F(2x2,2x2) form only where its position GEMMs fill the chip (T * 4 Cin * Cout >= 1.6e8, >= 400 work items):
    # at the yml's batch of 8 (and for the 8x8 layer up to B = 24) one direct split-K GEMM is faster (profiles/r05_b8_winograd_threshold.txt)
    for B, H, Ci, Co, algo in ((8, 32, 128, 256, G), (8, 16, 256, 512, G), (8, 8, 512, 1024, G), (16, 16, 256, 512, G), (24, 32, 128, 256, W2),
                               (24, 16, 256, 512, W2), (24, 8, 512, 1024, G), (192, 8, 512, 1024, W2)):
        d, _ = K.conv_desc(B, H, H, Ci, Co, 4, 4, 2, 2, 'SAME')
        assert tuple(K.conv_algo(d, m) for m in ('fwd', 'bwd_data', 'bwd_filter')) == (algo,) * 3, (B, H, Ci, Co)
    d, _ = K.conv_desc(64, 8, 8, 512, 512, 3, 3, 1, 1, 'SAME', math=K.MATH_BF16)
    # bf16 math mode: no Winograd; all three primitives stage bf16 operand copies where the channel counts allow
    assert all(K.conv_algo(d, m) == 'implicit_gemm_bf16_operands' for m in ('fwd', 'bwd_data', 'bwd_filter'))


def test_bf16_image_cache_is_keyed_on_base_version_and_capture(monkeypatch):
    """kernels.bf16_image (host logic, no GPU): one image per tensor — found again through a permuted-and-permuted-back view
    (the layout helpers hand the convs such views), made anew after an in-place write, and never shared between launch
    contexts (an image made eagerly or in another capture is not part of the graph being captured)."""
    import torch
    from t2i_amd import kernels as K
    made = []

    class Shim(object):
        cap = 0

        def t2i_capture_id(self, stream):
            return self.cap
    shim = Shim()
    monkeypatch.setattr(K, 'lib', shim)
    monkeypatch.setattr(K, '_stream', lambda: None)
    monkeypatch.setattr(K, 'cast_bf16', lambda t: made.append(t) or torch.zeros(t.shape, dtype=torch.bfloat16))
    y = torch.randn(2, 4, 4, 8)
    img = K.bf16_image(y)
    assert len(made) == 1 and K.bf16_image(y) is img and len(made) == 1
    view = y.permute(0, 3, 1, 2).permute(0, 2, 3, 1)                 # NHWC -> "NCHW" -> NHWC: another object, same memory, same order
    assert view is not y and K.bf16_image(view) is img and len(made) == 1
    part = K.bf16_image(y[1:])                                       # round 6: a contiguous run of rows of a tensor that has an image reads
    assert part is not img and len(made) == 1 and part.data_ptr() == img[1:].data_ptr() and part.shape == y[1:].shape   # the same rows of that image
    z = torch.randn(2, 4, 4, 8)
    assert K.bf16_image(z[1:]).shape == z[1:].shape and len(made) == 2     # (no image on the base: the slice gets its own)
    y.add_(1.0)                                                       # version bump: the image is stale
    assert len(made) == 2
    img2 = K.bf16_image(view)
    assert img2 is not img and len(made) == 3 and K.bf16_image(y) is img2
    assert K.bf16_image(y[:1]).data_ptr() == img2.data_ptr() and len(made) == 3
    shim.cap = 7                                                      # inside a capture: the eager image is not reused ...
    img3 = K.bf16_image(y)
    assert img3 is not img2 and len(made) == 4 and K.bf16_image(view) is img3
    shim.cap = 8                                                      # ... nor is one made in an earlier capture
    assert K.bf16_image(y) is not img3 and len(made) == 5
    assert K.bf16_image(torch.randn(3, 5)) is None                    # 15 elements: no 8-element groups


def test_capture_mode_follows_the_process_group(t2i, tmp_path):
    """graphs.capture_mode: 'global' (torch's default) in a process without a process group; 'thread_local' as soon as one exists —
    its watchdog thread may touch the runtime while a capture is open (seen once as an abort of the full GPU suite).  An
    explicit 'thread_local' is never downgraded.  (One-rank gloo group over a file store: no network needed.)"""
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, %r)
import torch.distributed as dist
import t2i_amd
from t2i_amd.graphs import capture_mode
assert capture_mode() == 'global' and capture_mode('thread_local') == 'thread_local'
dist.init_process_group('gloo', init_method='file://%s', rank=0, world_size=1)
assert capture_mode() == 'thread_local' and capture_mode('global') == 'thread_local'
dist.destroy_process_group()
assert capture_mode() == 'global'
print('ok')
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / 'store'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-2000:]


def test_config3_arithmetic_is_per_network_and_per_direction(t2i):
    """kernels.math_scope / bwd_geom / CONFIG3_NET_MATH: inside the generator's scope descriptors are fp32-math and the
    tensors fp32 while their BACKWARD descriptors are bf16-math with the same geometry; outside the scope the process-wide setting
    (bf16 math, bf16 storage: the critic) holds again; a dry (launch-free) critic + generator step of the model routes every generator
    forward conv to fp32 math, every generator backward GEMM and every critic conv to bf16 math."""
    from t2i_amd import kernels as K
    from t2i_amd.models.wgancls.model import WGanCls
    K.set_math('bf16')
    K.set_storage('bf16')
    try:
        assert K._act_dtype((4, 4, 4, 64)) is torch.bfloat16
        with K.math_scope(*K.CONFIG3_NET_MATH['g_net']):
            g = K.conv_desc(4, 8, 8, 64, 128, 3, 3, 1, 1, 'SAME')
            gb = K.bwd_geom(g)
            assert g[0].math == K.MATH_F32 and gb[0].math == K.MATH_BF16 and gb is not g
            assert [getattr(gb[0], f) for f, _ in g[0]._fields_[:-1]] == [getattr(g[0], f) for f, _ in g[0]._fields_[:-1]]
            assert K.bwd_geom(g)[0] is gb[0]                                 # cached
            assert K._act_dtype((4, 4, 4, 64)) is torch.float32
        assert K.bwd_geom(g) is g and K._act_dtype((4, 4, 4, 64)) is torch.bfloat16      # outside: the layer's own arithmetic again
        d = K.conv_desc(4, 8, 8, 64, 128, 3, 3, 1, 1, 'SAME')
        assert d[0].math == K.MATH_BF16
        # the model: record (entry point, descriptor math) of every conv call of a dry critic step + generator step
        calls = []
        saved = {n: getattr(K, n) for n in ('conv_fwd', 'conv_fwd_stats', 'conv_bwd_data', 'conv_bwd_filter', 'conv_bwd_pair')}

        def tap(name, dpos):
            fn = saved[name]

            def f(*a, **kw):
                calls.append((name, a[dpos].math, a[0].dtype))
                return fn(*a, **kw)
            setattr(K, name, f)
        for n, pos in (('conv_fwd', 3), ('conv_fwd_stats', 3), ('conv_bwd_data', 3), ('conv_bwd_filter', 2), ('conv_bwd_pair', 5)):
            tap(n, pos)
        try:
            m = WGanCls(_cfg(4), device='cpu')
            m.net_math = dict(K.CONFIG3_NET_MATH)
            B = 4
            feed = {'x': torch.zeros(B, 64, 64, 3), 'x_mismatch': torch.zeros(B, 64, 64, 3), 'cond': torch.zeros(B, 1024), 'z': torch.zeros(B, 128),
                    'epsilon': torch.zeros(B, 1, 1, 1), 'ca_noise_d': torch.zeros(B, 128), 'ca_noise_g': torch.zeros(B, 128)}
            with K.dry_run():
                calls.clear()
                with torch.no_grad():
                    m.generator(feed['z'], feed['cond'], reuse=True)
                gen_fwd = list(calls)
                calls.clear()
                m.g_losses(feed)
        finally:
            for n, fn in saved.items():
                setattr(K, n, fn)
        assert gen_fwd and all(mth == K.MATH_F32 and dt == torch.float32 for _, mth, dt in gen_fwd), gen_fwd
        n_fwd = len(gen_fwd)
        step = calls
        assert [c[1] for c in step[:n_fwd]] == [K.MATH_F32] * n_fwd                    # the generator step's own forward
        rest = step[n_fwd:]                                                            # critic on G, then the whole backward
        assert rest and all(mth == K.MATH_BF16 for _, mth, _ in rest), [c for c in rest if c[1] != K.MATH_BF16][:5]
        assert any(n == 'conv_bwd_filter' and dt == torch.float32 for n, _, dt in rest)   # generator backward: bf16 math on fp32 tensors
    finally:
        K.set_storage('f32')
        K.set_math('f32')
