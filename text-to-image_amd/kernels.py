"""Tensor-level wrappers over the C ABI (include/t2i_hip.h).  Each function allocates its output with torch (device
memory + caching allocator are PyTorch's job), passes raw device pointers and the current HIP stream to libt2i_hip.so
and returns.  No arithmetic happens in Python or in torch here.

CPU tensors are refused: the product path has no CPU implementation.  The only exception is ``dry_run()``, used by
the host-logic tests and by variable creation, in which launches are skipped and outputs are uninitialised
``torch.empty`` of the right shape (shape inference only — values are garbage by construction).
"""
import contextlib
import ctypes
import os

import torch

from ._lib import DT_BF16, DT_F32, XFORM_HAVE, XFORM_KEEP, ConvDesc, ConvOpts, check, lib

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3

_DRY = [False]


@contextlib.contextmanager
def dry_run():
    """Shape-inference mode: no kernel is launched, outputs are uninitialised."""
    prev = _DRY[0]
    _DRY[0] = True
    try:
        yield
    finally:
        _DRY[0] = prev


def is_dry():
    return _DRY[0]


def _live(t):
    """True -> launch on the GPU.  False -> dry run.  CPU tensor outside dry_run -> error (no fallback)."""
    if _DRY[0]:
        return False
    if t.is_cuda:
        return True
    raise RuntimeError('text-to-image_amd kernels need a ROCm device tensor (got %s); there is no CPU path' % t.device)


def _chk(t, name='tensor', f32=False):
    """Activation tensors are float32, or bfloat16 under bf16 storage (set_storage); parameters and reduction results (f32=True)
    are always float32."""
    if t.dtype != torch.float32 and (f32 or t.dtype != torch.bfloat16):
        raise TypeError('%s must be float32%s, got %s' % (name, '' if f32 else ' or bfloat16', t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous (shape %s, strides %s)' % (name, tuple(t.shape), t.stride()))
    return t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- workspace: one grow-only buffer per (device, lane); all users of a buffer are ordered on one stream.  Lane 0 is the
# caller's current stream; autograd's filter-gradient side stream launches under lane 1 (WS_LANE) -----------------------
WS_LANE = [0]
_STREAM_LANE = {}        # hipStream_t -> lane, for streams that run convolutions of their own (stream_lane)
_WS = {}
_WS_MIN = 64 << 20
_WS_CAPTURED = set()     # lanes whose CURRENT buffer has been handed to a kernel inside a graph capture
_WS_RETIRED = []         # outgrown buffers that captured graphs still point at: kept alive for the life of the process


PAIR_LANE = 1000          # lane + PAIR_LANE: the second workspace of a fused backward pair (kernels.conv_bwd_pair)


def workspace(device, nbytes):
    """The lane's scratch buffer, grown on demand.  A hipGraph captured while a buffer was current has that buffer's
    address baked into its kernel arguments, so once a capture has used a buffer it is never freed: a later eager call
    that needs more (the sampler at SAMPLE_NUM > BATCH_SIZE, a bigger batch) gets a NEW buffer and the old one is retired
    but kept alive — replays keep writing into memory they still own."""
    lane = WS_LANE[0]
    if lane == 0 and _STREAM_LANE and device.type == 'cuda':
        lane = _STREAM_LANE.get(torch.cuda.current_stream(device).cuda_stream, 0)
    key = (device.type, device.index, lane)
    buf = _WS.get(key)
    capturing = torch.cuda.is_available() and device.type == 'cuda' and torch.cuda.is_current_stream_capturing()
    if buf is None or buf.numel() < nbytes:
        if capturing:
            raise RuntimeError('workspace would grow during graph capture; run warm-up iterations first')
        if buf is not None and key in _WS_CAPTURED:
            _WS_RETIRED.append(buf)
            _WS_CAPTURED.discard(key)
        buf = torch.empty(max(int(nbytes), _WS_MIN), dtype=torch.uint8, device=device)
        _WS[key] = buf
    if capturing:
        _WS_CAPTURED.add(key)
    return buf


def stream_lane(stream, device=None):
    """Give the HIP stream behind `stream` its own workspace lane, sized like lane 0: launches on it never share scratch with the
    main stream's.  The lane belongs to the underlying hipStream_t (PyTorch hands out 32 pooled streams per device round-robin):
    two models whose second streams are different pool streams get different lanes; if the pool gives two models the SAME stream
    their work is ordered on it anyway, and sharing the lane is safe.  At most 32 lanes ever exist."""
    lane = _STREAM_LANE.setdefault(stream.cuda_stream, 2 + len(_STREAM_LANE))
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    main = _WS.get((dev.type, dev.index, 0))
    key = (dev.type, dev.index, lane)
    for key in (key, (dev.type, dev.index, PAIR_LANE + lane)):       # ... and the lane of conv_bwd_pair's second workspace
        if main is not None and (_WS.get(key) is None or _WS[key].numel() < main.numel()):
            if _WS.get(key) is not None and key in _WS_CAPTURED:
                _WS_RETIRED.append(_WS[key]); _WS_CAPTURED.discard(key)
            _WS[key] = torch.empty(main.numel(), dtype=torch.uint8, device=dev)


# ---- convolution geometry (TF padding rules; reference utils/ops.py:58-71 passes the string through to TF) -------------
def same_pad(in_size, k, s):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2


_DESC_CACHE = {}

# Arithmetic of the conv family (t2i_conv_desc.math): 'f32' = exact fp32 matrix pipe (default, BASELINE config 2);
# 'bf16' = operands rounded to bf16 inside the kernel, bf16 MFMA with fp32 accumulation, fp32 tensors (config 3).
MATH_F32, MATH_BF16 = 0, 1
_MATH = [MATH_F32]


_MATH_CODES = {'f32': MATH_F32, 'fp32': MATH_F32, 'bf16': MATH_BF16}


def set_math(mode):
    """Select the arithmetic of every conv/deconv/dense descriptor created from now on: 'f32' or 'bf16'.  (Also forgets that a
    math_scope has been used: a new configuration starts from the process-wide rules for bf16 twins.)"""
    _MATH[0] = _MATH_CODES[str(mode).lower()]
    _MIXED[0] = False


def get_math():
    return 'bf16' if _MATH[0] == MATH_BF16 else 'f32'


# Storage of the activation tensors the wrappers allocate (ABI v6, BASELINE config 3 end to end): 'f32' (default), or 'bf16' —
# every activation / activation-gradient tensor with a multiple of 64 channels is then a torch.bfloat16 tensor, written by the
# producing kernel and read by the consuming kernel as bf16; the 3-channel image side, logits, losses, parameters, gradients of
# parameters, optimizer state and batch-norm statistics stay float32.  Needs set_math('bf16').
_STORE = [torch.float32]
_FORCE_F32 = [0]


def set_storage(mode):
    mode = {'f32': torch.float32, 'fp32': torch.float32, 'bf16': torch.bfloat16}[str(mode).lower()]
    if mode is torch.bfloat16 and _MATH[0] != MATH_BF16:
        raise ValueError("set_storage('bf16') needs set_math('bf16') first (bf16 tensors feed the bf16 matrix pipe)")
    _STORE[0] = mode
    _MIXED[0] = False          # a new configuration: the twin policy of an earlier math_scope does not carry over (see _MIXED)


def get_storage():
    return 'bf16' if _STORE[0] is torch.bfloat16 else 'f32'


@contextlib.contextmanager
def f32_outputs():
    """Inside, the wrappers allocate float32 outputs whatever the storage mode (small tensors that feed fp32-only kernels: the two
    conditioning-augmentation heads in front of t2i_ca_kl_fwd)."""
    _FORCE_F32[0] += 1
    try:
        yield
    finally:
        _FORCE_F32[0] -= 1


# Per-network arithmetic (config 3's compliant mode): a scope in which descriptors are created with another math mode and
# activations are allocated with another storage dtype than the process-wide setting.  The backward of a layer follows its forward
# (descriptors travel in the autograd context, gradients take the dtype of the tensor they are the gradient of), so entering the
# scope around a network's FORWARD is enough.  _MIXED: a scope has been used — with bf16 storage outside the scope every tensor
# that wants a bf16 image is one already, so float32 outputs (the scoped network's) get no bf16 twin.
# _MIXED is process-wide and sticky BY DESIGN within one configuration (the scoped network's backward runs outside the scope, long
# after it exited, and must still see the policy); it is cleared by set_math(), set_storage() and forget_scopes() — the models call
# the latter when they are constructed outside any scope, so a later, unscoped model in the same process starts from the plain rules.
_MIXED = [False]
_BWD_MATH = [None]
_SCOPE_TWINS = [os.environ.get('T2I_SCOPE_TWINS', '1') != '0']      # 0: the rule of rounds 4-5 (A/B; it never made a twin, see _twin_for)
_SCOPE_GRAD = [True]     # torch.is_grad_enabled() when the innermost math_scope was entered — by model code, outside any autograd Function (inside a
                         # Function.forward grad mode is always off): "will this forward be differentiated?", for _twin_for


def forget_scopes():
    """Clear the sticky twin policy a finished math_scope left behind (no-op while a scope is open)."""
    if _BWD_MATH[0] is None:
        _MIXED[0] = False
          # arithmetic of the BACKWARD GEMMs of layers created in the current scope (None: the forward's)


@contextlib.contextmanager
def math_scope(math=None, storage=None, bwd_math=None):
    """with math_scope('f32', 'f32'): ...  — conv descriptors and activation tensors created inside use this arithmetic / storage.
    bwd_math: the input- and filter-gradient GEMMs of those layers run in this arithmetic instead (the backward is linear in the
    upstream gradient, so its rounding errors add up once; the forward's are amplified by every nonlinear term behind them)."""
    if math is None and storage is None and bwd_math is None:
        yield
        return
    pm, ps, pb, pg = _MATH[0], _STORE[0], _BWD_MATH[0], _SCOPE_GRAD[0]
    _SCOPE_GRAD[0] = torch.is_grad_enabled()
    _MIXED[0] = 'bwd_bf16' if (bwd_math is not None and _MATH_CODES[str(bwd_math).lower()] == MATH_BF16) or _MIXED[0] == 'bwd_bf16' else 'fwd'
    try:
        if math is not None:
            _MATH[0] = _MATH_CODES[str(math).lower()]
        if storage is not None:
            _STORE[0] = {'f32': torch.float32, 'fp32': torch.float32, 'bf16': torch.bfloat16}[str(storage).lower()]
        _BWD_MATH[0] = _MATH_CODES[str(bwd_math).lower()] if bwd_math is not None else None
        yield
    finally:
        _MATH[0], _STORE[0], _BWD_MATH[0], _SCOPE_GRAD[0] = pm, ps, pb, pg


# BASELINE configs[2] ("bf16 MFMA") as benchmarked and as its parity test asserts (<= 2e-2 on every tensor, tests/test_step_b64_gpu.py):
# per-network arithmetic (math, storage, backward math) handed to math_scope by the models.  Networks not named run in the
# process-wide setting (set_math('bf16') + set_storage('bf16')): the critic.  DESIGN.md 4.16.
CONFIG3_NET_MATH = {'g_net': ('f32', 'f32', 'bf16')}
# Every forward GEMM in fp32 math on fp32 tensors, every input- and filter-gradient GEMM in bf16 math (bf16 operand images, fp32 accumulate): the
# arithmetic under which the batch-normalised StackGAN Stage-II step stays inside 2e-2 (tests/test_fullsize_gpu.py)
FWD_F32_BWD_BF16 = ('f32', 'f32', 'bf16')


def bwd_geom(geom):
    """(descriptor, workspace bytes) for the backward GEMMs of a layer whose forward uses `geom`: the same geometry in the scope's
    backward arithmetic (math_scope(bwd_math=...)), or `geom` itself."""
    bm = _BWD_MATH[0]
    d = geom[0]
    if bm is None or bm == d.math:
        return geom
    key = ('bwd', id(d), bm)
    hit = _DESC_CACHE.get(key)
    if hit is None:
        d2 = ConvDesc(d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.SH, d.SW, d.pad_t, d.pad_l, bm)
        hit = _DESC_CACHE[key] = (d2, int(lib.t2i_conv2d_workspace_bytes(ctypes.byref(d2))), d)      # (keeps d alive: id(d) is the key)
    return hit[0], hit[1]


def _act_dtype(shape, want=None):
    """dtype of a freshly allocated activation tensor of `shape`; want: the caller's explicit choice (a gradient takes the dtype of
    the tensor it is the gradient of)."""
    if want is not None:
        return want
    if _STORE[0] is torch.bfloat16 and not _FORCE_F32[0] and _MATH[0] == MATH_BF16 and shape[-1] % 64 == 0:
        n = 1
        for k in shape:
            n *= int(k)
        if n % 8 == 0:
            return torch.bfloat16
    return torch.float32


def _dt(t):
    return DT_BF16 if t.dtype == torch.bfloat16 else DT_F32


def _same_dt(*ts):
    ts = [t for t in ts if t is not None]
    if any(t.dtype != ts[0].dtype for t in ts):
        raise TypeError('activation tensors of one call must share a dtype, got %s' % [str(t.dtype) for t in ts])
    return _dt(ts[0])


def cast_f32(t):
    """bf16 activation -> float32 tensor (exact), through t2i_cast_f32."""
    if t.dtype == torch.float32:
        return t
    _chk(t, 't')
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    if _live(t):
        check(lib.t2i_cast_f32(_ptr(t), t.numel(), _ptr(out), _stream()), 't2i_cast_f32')
    return out


def conv_desc(B, H, W, Cin, Cout, KH, KW, SH, SW, padding, math=None):
    """-> (ConvDesc, workspace_bytes) for y = conv(x[B,H,W,Cin], w[KH,KW,Cin,Cout]) with a TF padding string."""
    math = _MATH[0] if math is None else math
    key = (B, H, W, Cin, Cout, KH, KW, SH, SW, padding.upper(), math)
    hit = _DESC_CACHE.get(key)
    if hit is not None:
        return hit
    p = padding.upper()
    if p == 'SAME':
        Ho, pt = same_pad(H, KH, SH)
        Wo, pl = same_pad(W, KW, SW)
    elif p == 'VALID':
        Ho, Wo, pt, pl = (H - KH) // SH + 1, (W - KW) // SW + 1, 0, 0
    else:
        raise ValueError('Invalid padding %s' % padding)
    if Ho <= 0 or Wo <= 0:
        raise ValueError('convolution output is empty for input %dx%d kernel %dx%d' % (H, W, KH, KW))
    d = ConvDesc(B, H, W, Cin, Ho, Wo, Cout, KH, KW, SH, SW, pt, pl, math)
    ws = int(lib.t2i_conv2d_workspace_bytes(ctypes.byref(d)))
    _DESC_CACHE[key] = (d, ws)
    return d, ws


def rebatch(geom, B):
    """(descriptor, workspace bytes) of the same convolution on another batch size: a slice of a stacked pass (stacked.py) is an
    ordinary pass of its own.  Geometry, padding and arithmetic are copied from `geom`'s descriptor."""
    d = geom[0]
    if d.B == B:
        return geom
    key = ('rebatch', B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.SH, d.SW, d.pad_t, d.pad_l, d.math)
    hit = _DESC_CACHE.get(key)
    if hit is None:
        d2 = ConvDesc(B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.SH, d.SW, d.pad_t, d.pad_l, d.math)
        hit = _DESC_CACHE[key] = (d2, int(lib.t2i_conv2d_workspace_bytes(ctypes.byref(d2))))
    return hit


def deconv_desc(B, H, W, Cin, Cout, KH, KW, SH, SW, padding, math=None):
    """Descriptor of the ADJOINT conv of a TF conv2d_transpose x[B,H,W,Cin] -> [B,Hout,Wout,Cout]: that conv maps
    [B,Hout,Wout,Cout] -> [B,H,W,Cin] with HWIO filter [KH,KW,Cout,Cin] (the TF deconv layout)."""
    p = padding.upper()
    if p == 'SAME':
        Hout, Wout = H * SH, W * SW
    elif p == 'VALID':
        Hout, Wout = (H - 1) * SH + KH, (W - 1) * SW + KW
    else:
        raise ValueError('Invalid padding %s' % padding)
    d, ws = conv_desc(B, Hout, Wout, Cout, Cin, KH, KW, SH, SW, padding, math)
    if (d.Ho, d.Wo) != (H, W):
        raise ValueError('conv2d_transpose geometry mismatch: adjoint conv gives %dx%d, input is %dx%d' % (d.Ho, d.Wo, H, W))
    return d, ws


# Optional per-launch timing hook (bench.py): an object with begin(flops) -> end_event; the wrappers record the end
# event right after the launch, on the same (current) stream.
_TIMER = [None]


def set_conv_timer(timer):
    _TIMER[0] = timer


def conv_flops(d):
    """Algorithmic FLOPs of any of the three conv primitives for descriptor d (2 x MACs incl. padded taps)."""
    return 2 * d.B * d.Ho * d.Wo * d.Cout * d.KH * d.KW * d.Cin


def _ws_args(t, nbytes):
    if nbytes == 0:
        return None, 0
    buf = workspace(t.device, nbytes)
    return _ptr(buf), buf.numel()


# bf16 math: the GEMMs read bf16 images of their activation operands.  An activation is used by up to three convs (forward,
# filter gradient, the second-order pieces of the gradient penalty), a gradient by two (input and filter gradient): the image
# is made once per tensor (and version) here, kept on the tensor object and handed to every conv that reads the tensor
# (t2i_conv_opts.a_image / b_image), instead of each entry point staging its own copy into the workspace.
_BF16_IMAGES = [os.environ.get('T2I_BF16_IMAGES', '1') != '0']
_H_ALGO = {}


def bf16_images(on):
    prev, _BF16_IMAGES[0] = _BF16_IMAGES[0], bool(on)
    return prev


def _h_path(d, mode):
    """does this descriptor's `mode` run the GEMM with bf16 operands in memory?"""
    if d.math != MATH_BF16 or not _BF16_IMAGES[0]:
        return False
    key = (id(d), mode)
    hit = _H_ALGO.get(key)
    if hit is None:
        hit = _H_ALGO[key] = (conv_algo(d, mode) == 'implicit_gemm_bf16_operands')
    return hit


def cast_bf16(t):
    _chk(t, 't')
    img = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    check(lib.t2i_cast_bf16(_ptr(t), t.numel(), _ptr(img), _stream()), 't2i_cast_bf16')
    return img


def _image_holder(t):
    """The tensor object the image is kept on: the base of `t` when t is a full view of it (the layout helpers of utils/ops.py hand
    the convs permuted-and-permuted-back views of the producing kernel's output: same memory, same order, another object)."""
    b = t._base
    return b if (b is not None and b.data_ptr() == t.data_ptr() and b.numel() == t.numel()) else t


def bf16_image(t):
    """The bf16 image of fp32 tensor `t` (None if it cannot have one): cached on the tensor object with the version it was made of.
    INVARIANT: the kernels of this package write through raw pointers and do not bump tensor._version, so every wrapper that
    overwrites an EXISTING tensor (the `out=` arguments of axpby, col_reduce, conv_bwd_filter) drops the cached image of that
    tensor itself (_drop_image); all other wrappers write freshly allocated outputs."""
    if t.dtype == torch.bfloat16:
        return t                       # bf16 storage: the tensor is its own image
    if t.numel() % 8 != 0 or t.data_ptr() % 16 != 0:
        return None
    h = _image_holder(t)
    c = getattr(h, '_t2i_h', None)
    cap = int(lib.t2i_capture_id(_stream()))      # an image made outside the current capture (eager, or an earlier capture) is
    if c is not None and c[0] == t._version and c[2] == t.data_ptr() and c[3] == cap:      # not part of this graph: never reused
        return c[1]
    if c is None and h is t:
        # a contiguous run of rows of a tensor that HAS an image (a part of a stacked pass, stacked.py: the producer wrote the twin of the
        # whole stacked tensor): the same rows of that image
        b = t._base
        cb = getattr(b, '_t2i_h', None) if b is not None else None
        if (cb is not None and cb[0] == t._version and cb[3] == cap and cb[2] == b.data_ptr() and t.is_contiguous() and b.is_contiguous() and
                t.dim() == b.dim() and t.shape[1:] == b.shape[1:] and t.dtype == b.dtype):
            row = t[0].numel() * t.element_size() if t.shape[0] else 0
            off = t.data_ptr() - b.data_ptr()
            if row and off % row == 0 and 0 <= off // row and off // row + t.shape[0] <= b.shape[0]:
                r0 = off // row
                part = cb[1][r0:r0 + t.shape[0]]
                if part.data_ptr() % 16 == 0:
                    return part
    img = cast_bf16(t)
    h._t2i_h = (t._version, img, t.data_ptr(), cap)
    return img


def _operand_images(opts, a, b=None):
    """Put the bf16 images of the call's activation operands into its t2i_conv_opts; the returned tensors must stay alive until
    the conv call has been issued."""
    ia = bf16_image(a) if (a is not None and a.dtype != torch.bfloat16) else None      # a bf16 tensor is passed as the operand itself
    ib = bf16_image(b) if (b is not None and b.dtype != torch.bfloat16) else None
    opts.a_image = ia.data_ptr() if ia is not None else None
    opts.b_image = ib.data_ptr() if ib is not None else None
    return ia, ib


# ... and where the tensor a conv will read comes out of one of our own kernels (activation / batch-norm apply / residual join /
# a conv epilogue with its activation fused / activation backward), that kernel writes the bf16 image as a TWIN of its fp32
# output in the same pass (the y_h arguments / t2i_conv_opts.out_image): no cast launch at all for it.
_TWINS = [os.environ.get('T2I_BF16_TWINS', '1') != '0']


def bf16_twins(on):
    prev, _TWINS[0] = _TWINS[0], bool(on)
    return prev


def _twin_for(out, *inputs):
    """A buffer for the bf16 twin of `out` (bf16 math, a multiple of 64 channels: a conv reads it next), or None.  The producers
    write it on their vectorised path only, so every tensor of the call must be 16-byte aligned (the entry points refuse otherwise)."""
    want = _MATH[0] == MATH_BF16
    if _MIXED[0]:
        if _BWD_MATH[0] == MATH_BF16:            # inside a scope whose backward GEMMs read bf16 images: the saved activations get theirs here
            # (round 6: the grad mode of the scope's entry — the producers run inside autograd Functions, where torch.is_grad_enabled() is
            # always False, so this rule never made a twin and every saved activation of the scoped network was cast in a launch of its own)
            want = _SCOPE_GRAD[0] if _SCOPE_TWINS[0] else torch.is_grad_enabled()
        elif _STORE[0] is torch.bfloat16:        # outside the scope, bf16 storage: a float32 tensor here belongs to the scoped network's backward
            want = _MIXED[0] == 'bwd_bf16'
    if out.dtype != torch.float32 or not want or not (_TWINS[0] and _BF16_IMAGES[0]) or out.shape[-1] % 64 or out.numel() % 8:
        return None
    if any(t is not None and t.data_ptr() % 16 for t in (out,) + inputs):
        return None
    return torch.empty(out.shape, dtype=torch.bfloat16, device=out.device)


def _twin_keep(out, img, written=True):
    """Remember `img` as the bf16 image of `out` (the call that was handed it has been issued and wrote it)."""
    if img is not None and written:
        out._t2i_h = (out._version, img, out.data_ptr(), int(lib.t2i_capture_id(_stream())))


# fp32 Winograd: the forward conv of a layer and its filter gradient transform the same x.  conv_fwd(..., keep_xform=True) leaves
# the transform in a tensor it returns through `LAST_XFORM` (None if the call took another path); conv_bwd_filter(..., xform=V)
# reads it instead of transforming x again (t2i_conv2d_input_transform).
_XFORM_BYTES = {}
LAST_XFORM = [None]
_SHARE_XFORM = [os.environ.get('T2I_SHARE_XFORM', '1') != '0']


def share_xform(on):
    prev, _SHARE_XFORM[0] = _SHARE_XFORM[0], bool(on)
    return prev


def conv_xform_bytes(d):
    if d.math != MATH_F32 or not _SHARE_XFORM[0]:
        return 0
    n = _XFORM_BYTES.get(id(d))
    if n is None:
        n = _XFORM_BYTES[id(d)] = int(lib.t2i_conv2d_input_transform_bytes(ctypes.byref(d)))
    return n


def _xform_offer(opts, x, d, keep):
    """Offer the forward conv a buffer to leave its Winograd input transform in (t2i_conv_opts.xform, T2I_XFORM_KEEP)."""
    LAST_XFORM[0] = None
    nb = conv_xform_bytes(d) if keep else 0
    if not nb:
        return None
    V = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
    opts.xform, opts.xform_bytes, opts.xform_mode = V.data_ptr(), nb, XFORM_KEEP
    return V


def _xform_taken(opts, V):
    if V is not None and opts.xform_kept:
        LAST_XFORM[0] = V


def _storage_flags(opts, a, b, out):
    opts.in_dtype = (1 if a.dtype == torch.bfloat16 else 0) | (2 if (b is not None and b.dtype == torch.bfloat16) else 0)
    opts.out_dtype = DT_BF16 if (out is not None and out.dtype == torch.bfloat16) else DT_F32


# Where the NEXT conv_fwd whose output has exactly this shape and dtype writes (one shot): the critic step lets the generator's last
# kernel put G straight into its slot of the stacked critic input [G | x | x_mismatch | x_hat] instead of concatenating afterwards.
_OUT_INTO = [None]


class output_into(object):
    def __init__(self, t):
        self.t = t

    def __enter__(self):
        self.prev, _OUT_INTO[0] = _OUT_INTO[0], self.t

    def __exit__(self, *a):
        self.taken = _OUT_INTO[0] is None
        _OUT_INTO[0] = self.prev


def _take_out(shape, dtype, device):
    t = _OUT_INTO[0]
    if t is not None and tuple(t.shape) == tuple(shape) and t.dtype == dtype and t.is_contiguous() and t.device == device:
        _OUT_INTO[0] = None
        _drop_image(t)
        return t
    return None


def conv_fwd(x, w, bias, d, ws_bytes, act=ACT_NONE, alpha=0.2, keep_xform=False, out_dtype=None):
    _chk(x, 'x'); _chk(w, 'w', f32=True)
    oshape, odt = (d.B, d.Ho, d.Wo, d.Cout), _act_dtype((d.B, d.Ho, d.Wo, d.Cout), out_dtype)
    y = _take_out(oshape, odt, x.device) if _OUT_INTO[0] is not None else None
    if y is None:
        y = torch.empty(oshape, dtype=odt, device=x.device)
    if _live(x):
        wsp, wsn = _ws_args(x, ws_bytes)
        ev = _TIMER[0].begin(conv_flops(d), conv_algo(d, 'fwd')) if _TIMER[0] is not None else None
        opts = ConvOpts()          # every allocation happens BEFORE the call; nothing is armed on the library side
        _storage_flags(opts, x, None, y)
        keep = _operand_images(opts, x) if _h_path(d, 'fwd') else None
        twin = _twin_for(y) if (act != ACT_NONE and d.math == MATH_BF16) else None     # conv + bias + lrelu feeds the next conv directly
        opts.out_image = twin.data_ptr() if twin is not None else None
        V = _xform_offer(opts, x, d, keep_xform)
        check(lib.t2i_conv2d_fwd(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(_chk(bias, 'bias') if bias is not None else None),
                                 _ptr(y), act, alpha, ctypes.byref(opts), wsp, wsn, _stream()), 't2i_conv2d_fwd')
        _twin_keep(y, twin, opts.out_image_written)
        _xform_taken(opts, V)
        if ev is not None:
            ev.record()
    return y


# Batch-norm statistics produced by a conv epilogue, waiting for the batch norm that consumes that conv's output:
# output address -> (partials [2, chunks, C], chunks).  Filled by conv_fwd(..., stats=True), emptied by take_stats().
_STATS = {}


def take_stats(x):
    """(column sums, centred second moments sum (x - mean)^2) of `x` if the conv that produced it left its epilogue
    partials (per-tile sums and second moments about each tile's mean, merged here with Chan's update), else None."""
    hit = _STATS.pop(x.data_ptr(), None)
    if hit is None:
        return None
    part, chunks, tile_rows, shape = hit
    if tuple(x.shape) != shape:
        return None
    C = x.shape[-1]
    s0 = torch.empty(C, dtype=torch.float32, device=x.device); s1 = torch.empty_like(s0)
    check(lib.t2i_bn_stats_tiles(_ptr(part), ctypes.c_void_p(part.data_ptr() + chunks * C * 4), chunks, tile_rows, x.numel() // C, C,
                                 _ptr(s0), _ptr(s1), _stream()), 't2i_bn_stats_tiles')
    return s0, s1


def bn_train_stats(x, gamma, beta, eps, decay, moving_mean=None, moving_var=None):
    """Batch statistics of x (view [rows, C]) and the batch norm's finalize step in one chain -> (mean, rstd, scale, shift);
    moving averages updated in place when given.  Uses the producing conv's epilogue partials when it left any (take_stats)."""
    _chk(x, 'x')
    C = x.shape[-1]
    rows = x.numel() // C
    mean, rstd, scale, shift = (torch.empty(C, dtype=torch.float32, device=x.device) for _ in range(4))
    if _live(x):
        hit = _STATS.pop(x.data_ptr(), None)
        if hit is not None and tuple(x.shape) != hit[3]:
            hit = None
        if hit is not None:
            part, chunks, tile_rows, _ = hit
            check(lib.t2i_bn_train_fwd_stats(None, _ptr(part), ctypes.c_void_p(part.data_ptr() + chunks * C * 4), chunks, tile_rows, rows, C,
                                             _ptr(_chk(gamma)), _ptr(_chk(beta)), eps, decay, _ptr(mean), _ptr(rstd), _ptr(scale), _ptr(shift),
                                             _ptr(moving_mean), _ptr(moving_var), None, 0, DT_F32, _stream()), 't2i_bn_train_fwd_stats')
        else:
            wsp, wsn = _ws_args(x, int(lib.t2i_col_reduce_workspace_bytes(rows, C)))
            check(lib.t2i_bn_train_fwd_stats(_ptr(x), None, None, 0, 0, rows, C, _ptr(_chk(gamma)), _ptr(_chk(beta)), eps, decay, _ptr(mean),
                                             _ptr(rstd), _ptr(scale), _ptr(shift), _ptr(moving_mean), _ptr(moving_var), wsp, wsn, _dt(x),
                                             _stream()), 't2i_bn_train_fwd_stats')
    return mean, rstd, scale, shift


def bn_bwd_fused(dy, y, x, mean, rstd, gamma, act, alpha=0.2, dgamma_out=None, dbeta_out=None, out=None):
    """Training-mode batch-norm backward in three launches (C % 4 == 0).  y: the activation output behind the batch norm or
    None.  -> (dx, dgamma, dbeta); dgamma_out / dbeta_out: gradient-arena slots to ACCUMULATE into; out: where dx goes (a group's
    slice of a batched pass; no bf16 twin then)."""
    _chk(dy, 'dy'); _chk(x, 'x')
    C = x.shape[-1]
    rows = x.numel() // C
    if out is not None:
        assert out.shape == x.shape and out.dtype == x.dtype and out.is_contiguous()
        _drop_image(out)
    dx = out if out is not None else torch.empty_like(x)
    gmask = torch.empty_like(x) if y is not None else None
    acc = dgamma_out is not None
    assert acc == (dbeta_out is not None)
    dgamma = dgamma_out if acc else torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = dbeta_out if acc else torch.empty(C, dtype=torch.float32, device=x.device)
    if _live(x):
        wsp, wsn = _ws_args(x, int(lib.t2i_bn_bwd_fused_workspace_bytes(rows, C)))
        twin = _twin_for(dx) if out is None else None     # dx is the gradient of the conv in front of the batch norm: its bwd_data / bwd_filter operand
        check(lib.t2i_bn_bwd_fused(_ptr(dy), _ptr(_chk(y, 'y') if y is not None else None), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(_chk(gamma)),
                                   rows, C, act, alpha, _ptr(gmask), _ptr(dx), _ptr(twin), _ptr(dgamma), _ptr(dbeta), 1 if acc else 0, wsp, wsn,
                                   _same_dt(dy, y, x), _stream()), 't2i_bn_bwd_fused')
        _twin_keep(dx, twin)
    return dx, dgamma, dbeta


def bn_stats(x):
    """x viewed as [rows, C] -> (sum over rows, sum over rows of (x - mean)^2), numerically stable (t2i_bn_stats)."""
    _chk(x, 'x')
    C = x.shape[-1]
    rows = x.numel() // C
    s0 = torch.empty(C, dtype=torch.float32, device=x.device); s1 = torch.empty_like(s0)
    if _live(x):
        need = int(lib.t2i_col_reduce_workspace_bytes(rows, C))
        wsp, wsn = _ws_args(x, need)
        check(lib.t2i_bn_stats(_ptr(x), rows, C, _ptr(s0), _ptr(s1), wsp, wsn, _stream()), 't2i_bn_stats')
    return s0, s1


def conv_fwd_stats(x, w, bias, d, ws_bytes, act=ACT_NONE, alpha=0.2, keep_xform=False, out_dtype=None):
    """conv_fwd whose epilogue also leaves the per-tile column sums of y, y*y for the batch norm behind it (take_stats)."""
    _chk(x, 'x'); _chk(w, 'w', f32=True)
    y = torch.empty((d.B, d.Ho, d.Wo, d.Cout), dtype=_act_dtype((d.B, d.Ho, d.Wo, d.Cout), out_dtype), device=x.device)
    if _live(x):
        wsp, wsn = _ws_args(x, ws_bytes)
        nbytes = int(lib.t2i_conv2d_stats_bytes(ctypes.byref(d)))
        part = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        chunks, tile_rows = ctypes.c_int32(0), ctypes.c_int32(0)
        ev = _TIMER[0].begin(conv_flops(d), conv_algo(d, 'fwd')) if _TIMER[0] is not None else None
        opts = ConvOpts()
        _storage_flags(opts, x, None, y)
        keep = _operand_images(opts, x) if _h_path(d, 'fwd') else None
        V = _xform_offer(opts, x, d, keep_xform)
        check(lib.t2i_conv2d_fwd_stats(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(_chk(bias, 'bias') if bias is not None else None),
                                       _ptr(y), act, alpha, _ptr(part), nbytes, ctypes.byref(chunks), ctypes.byref(tile_rows),
                                       ctypes.byref(opts), wsp, wsn, _stream()), 't2i_conv2d_fwd_stats')
        _xform_taken(opts, V)
        if ev is not None:
            ev.record()
        if chunks.value > 0:
            if len(_STATS) > 64:
                _STATS.clear()
            _STATS[y.data_ptr()] = (part, int(chunks.value), int(tile_rows.value), tuple(y.shape))
    return y


def conv_bwd_data(dy, w, bias, d, ws_bytes, act=ACT_NONE, alpha=0.2, out_dtype=None):
    _chk(dy, 'dy'); _chk(w, 'w', f32=True)
    dx = torch.empty((d.B, d.H, d.W, d.Cin), dtype=_act_dtype((d.B, d.H, d.W, d.Cin), out_dtype), device=dy.device)
    if _live(dy):
        wsp, wsn = _ws_args(dy, ws_bytes)
        ev = _TIMER[0].begin(conv_flops(d), conv_algo(d, 'bwd_data')) if _TIMER[0] is not None else None
        opts = ConvOpts()
        _storage_flags(opts, dy, None, dx)
        keep = _operand_images(opts, dy) if _h_path(d, 'bwd_data') else None
        twin = _twin_for(dx) if (act != ACT_NONE and d.math == MATH_BF16) else None
        opts.out_image = twin.data_ptr() if twin is not None else None
        check(lib.t2i_conv2d_bwd_data(ctypes.byref(d), _ptr(dy), _ptr(w),
                                      _ptr(_chk(bias, 'bias') if bias is not None else None), _ptr(dx), act, alpha, ctypes.byref(opts),
                                      wsp, wsn, _stream()), 't2i_conv2d_bwd_data')
        _twin_keep(dx, twin, opts.out_image_written)
        if ev is not None:
            ev.record()
    return dx


def conv_bwd_filter(x, dy, d, ws_bytes, out=None, xform=None, xform_valid_rows=0, xform_plane_rows=0, accumulate=True):
    """dw = x (*) dy.  out: an existing [KH,KW,Cin,Cout]-sized buffer to ACCUMULATE into (dw += ...), e.g. the
    optimizer's gradient arena; returns it.  xform_valid_rows (with xform): the kept transform is current for that many leading images
    only — the library regenerates the rest from x, in place in `xform` (t2i_conv_opts.xform_valid_rows)."""
    _chk(x, 'x'); _chk(dy, 'dy')
    if out is not None:
        _chk(out, 'out', f32=True)
        assert out.numel() == d.KH * d.KW * d.Cin * d.Cout
        _drop_image(out)
    dw = out if out is not None else torch.empty((d.KH, d.KW, d.Cin, d.Cout), dtype=torch.float32, device=x.device)
    if _live(x):
        wsp, wsn = _ws_args(x, ws_bytes)
        ev = _TIMER[0].begin(conv_flops(d), conv_algo(d, 'bwd_filter')) if _TIMER[0] is not None else None
        opts = ConvOpts()
        _storage_flags(opts, x, dy, None)
        keep = _operand_images(opts, x, dy) if _h_path(d, 'bwd_filter') else None
        if xform is not None and conv_xform_bytes(d):
            opts.xform, opts.xform_bytes, opts.xform_mode = xform.data_ptr(), xform.numel() * 4, XFORM_HAVE
            opts.xform_valid_rows = int(xform_valid_rows)
            opts.xform_plane_rows = int(xform_plane_rows)       # the transform of a larger, stacked batch whose leading images are x
        # accumulate=False with out=: the slot's FIRST contribution of a step is a plain store (optim.Arena: no zero-fill, no read)
        check(lib.t2i_conv2d_bwd_filter(ctypes.byref(d), _ptr(x), _ptr(dy), _ptr(dw), 1 if (out is not None and accumulate) else 0, ctypes.byref(opts),
                                        wsp, wsn, _stream()), 't2i_conv2d_bwd_filter')
        if ev is not None:
            ev.record()
    return dw


PAIR_FWD, PAIR_BWD_DATA = 0, 1
_PAIR = [os.environ.get('T2I_PAIR_CALLS', '1') != '0']


def pair_calls(on=None):
    """Should a layer's backward hand its two GEMMs to conv_bwd_pair (one launch where the library can fuse them)?"""
    if on is not None:
        _PAIR[0] = bool(on)
    return _PAIR[0]


def conv_bwd_pair(first, g, w, fx, fdy, d, ws_bytes, dw_out, out_dtype=None, accumulate=True):
    """One layer's backward pair: out1 = conv^T(g, w) (first = PAIR_BWD_DATA) or conv(g, w) (PAIR_FWD), and dw_out += fx (*) fdy
    (the gradient sink).  Same results as conv_bwd_data / conv_fwd followed by conv_bwd_filter(out=dw_out); on bf16 tensors the
    two GEMMs share one launch (t2i_conv2d_bwd_pair).  Returns out1."""
    _chk(g, 'g'); _chk(w, 'w', f32=True); _chk(fx, 'fx'); _chk(fdy, 'fdy'); _chk(dw_out, 'dw_out', f32=True)
    assert dw_out.numel() == d.KH * d.KW * d.Cin * d.Cout
    _drop_image(dw_out)
    shape = (d.B, d.H, d.W, d.Cin) if first == PAIR_BWD_DATA else (d.B, d.Ho, d.Wo, d.Cout)
    out1 = torch.empty(shape, dtype=_act_dtype(shape, out_dtype), device=g.device)
    if _live(g):
        which = 'bwd_data' if first == PAIR_BWD_DATA else 'fwd'
        ws1p, ws1n = _ws_args(g, ws_bytes)
        lane0 = WS_LANE[0]
        WS_LANE[0] = PAIR_LANE + (lane0 if lane0 else (_STREAM_LANE.get(torch.cuda.current_stream(g.device).cuda_stream, 0) if _STREAM_LANE else 0))
        try:
            ws2p, ws2n = _ws_args(g, ws_bytes)            # the pair's second workspace: both GEMMs are in flight together
        finally:
            WS_LANE[0] = lane0
        ev = _TIMER[0].begin(2 * conv_flops(d), conv_algo(d, which)) if _TIMER[0] is not None else None
        o1, o2 = ConvOpts(), ConvOpts()
        _storage_flags(o1, g, None, out1)
        keep1 = _operand_images(o1, g) if _h_path(d, which) else None
        _storage_flags(o2, fx, fdy, None)
        keep2 = _operand_images(o2, fx, fdy) if _h_path(d, 'bwd_filter') else None
        check(lib.t2i_conv2d_bwd_pair(ctypes.byref(d), first, _ptr(g), _ptr(w), _ptr(out1), ctypes.byref(o1), _ptr(fx), _ptr(fdy), _ptr(dw_out), 1 if accumulate else 0,
                                      ctypes.byref(o2), ws1p, ws1n, ws2p, ws2n, _stream()), 't2i_conv2d_bwd_pair')
        del keep1, keep2
        if ev is not None:
            ev.record()
    return out1


def col_reduce(a, b=None, want_second=False, out=None, center=None):
    """a viewed as [rows, C] (C = last dim).  -> (colsum(a), colsum(a*(b - center) or a*a) or None).
    out: an existing [C] buffer to ACCUMULATE colsum(a) into (a bias slot of the gradient arena)."""
    _chk(a, 'a')
    C = a.shape[-1]
    rows = a.numel() // C
    if out is not None:
        assert out.numel() == C and not want_second
        _drop_image(out)
    out0 = out if out is not None else torch.empty(C, dtype=torch.float32, device=a.device)
    out1 = torch.empty(C, dtype=torch.float32, device=a.device) if want_second else None
    if b is not None:
        _chk(b, 'b')
        assert b.shape == a.shape
    if _live(a):
        need = int(lib.t2i_col_reduce_workspace_bytes(rows, C))
        wsp, wsn = _ws_args(a, need)
        check(lib.t2i_col_reduce(_ptr(a), _ptr(b), _ptr(_chk(center, 'center') if center is not None else None), rows, C, _ptr(out0),
                                 _ptr(out1), 1 if out is not None else 0, wsp, wsn, _same_dt(a, b), _stream()), 't2i_col_reduce')
    return out0, out1


def bn_finalize(s, ss, n, gamma, beta, eps, decay, moving_mean=None, moving_var=None):
    """s = column sums, ss = CENTRED second moments sum (x - mean)^2 (bn_stats / take_stats)."""
    C = s.numel()
    mean, rstd, scale, shift = (torch.empty(C, dtype=torch.float32, device=s.device) for _ in range(4))
    if _live(s):
        check(lib.t2i_bn_finalize(_ptr(s), _ptr(ss), n, C, _ptr(_chk(gamma)), _ptr(_chk(beta)), eps, decay, _ptr(mean),
                                  _ptr(rstd), _ptr(scale), _ptr(shift), _ptr(moving_mean), _ptr(moving_var), _stream()),
              't2i_bn_finalize')
    return mean, rstd, scale, shift


def bn_apply(x, scale, shift, act=ACT_NONE, alpha=0.2, out=None):
    """out: a contiguous tensor (view) of x's shape and dtype to write into — one group's slice of a batched pass (no bf16 twin then)."""
    _chk(x, 'x')
    C = x.shape[-1]
    if out is not None:
        assert out.shape == x.shape and out.dtype == x.dtype and out.is_contiguous()
        _drop_image(out)
    y = out if out is not None else torch.empty_like(x)
    if _live(x):
        twin = _twin_for(y, x, scale, shift) if out is None else None
        check(lib.t2i_bn_apply(_ptr(x), _ptr(scale), _ptr(shift), x.numel() // C, C, act, alpha, _ptr(y), _ptr(twin), _dt(x), _stream()),
              't2i_bn_apply')
        _twin_keep(y, twin)
    return y


def bn_apply_groups(x, scales, shifts, act=ACT_NONE, alpha=0.2):
    """bn_apply on a batched pass: slice g of x along the batch axis is normalised with (scales[g], shifts[g]); one output tensor.
    (A wrapper of its own so that instrumentation sees one call per batch-norm layer, like every other layer of a batched pass.)"""
    groups = len(scales)
    b = x.shape[0] // groups
    y = torch.empty_like(x)
    for g in range(groups):
        _bn_apply_one(x[g * b:(g + 1) * b], scales[g], shifts[g], act, alpha, out=y[g * b:(g + 1) * b])
    return y


def bn_train_fwd_grouped(x, gamma, beta, eps, decay, groups, act=ACT_NONE, alpha=0.2, moving_mean=None, moving_var=None, moving_updates=1,
                         moving_groups=0):
    """Training-mode batch norm in at most three launches (t2i_bn_train_fwd_grouped): x [groups * b, ..., C] with per-group statistics
    (groups = 1: the ordinary batch norm).  Uses the producing conv's epilogue partials when it left any (conv_fwd_stats).
    -> (y, mean [groups, C], rstd [groups, C]); moving averages updated in place once per group, in group order."""
    _chk(x, 'x')
    C = x.shape[-1]
    rows_g = x.numel() // C // groups
    stat = torch.empty((4, groups, C), dtype=torch.float32, device=x.device)       # mean, rstd, scale, shift
    y = torch.empty_like(x)
    if _live(x):
        hit = _STATS.pop(x.data_ptr(), None)
        if hit is not None and (tuple(x.shape) != hit[3] or (groups > 1 and rows_g % hit[2] != 0) or hit[0].data_ptr() % 16 or (hit[1] * C * 4) % 16):
            hit = None
        tsum = tm2 = None
        tchunks = trows = 0
        if hit is not None:
            part, chunks, trows, _ = hit
            tsum, tm2, tchunks = ctypes.c_void_p(part.data_ptr()), ctypes.c_void_p(part.data_ptr() + chunks * C * 4), chunks // groups
        wsp, wsn = _ws_args(x, int(lib.t2i_bn_grouped_workspace_bytes(rows_g, C, groups)))
        twin = _twin_for(y, x)
        check(lib.t2i_bn_train_fwd_grouped(_ptr(x), rows_g, C, groups, _ptr(_chk(gamma)), _ptr(_chk(beta)), eps, decay, _ptr(stat[0]), _ptr(stat[1]),
                                           _ptr(stat[2]), _ptr(stat[3]), _ptr(moving_mean), _ptr(moving_var), act, alpha, _ptr(y), _ptr(twin),
                                           tsum, tm2, tchunks, trows, int(moving_updates), int(moving_groups), wsp, wsn, _dt(x), _stream()), 't2i_bn_train_fwd_grouped')
        _twin_keep(y, twin)
    return y, stat[0], stat[1]


def bn_bwd_grouped(dy, y, x, mean, rstd, gamma, groups, act, alpha=0.2, dgamma_out=None, dbeta_out=None):
    """Backward of bn_train_fwd_grouped in three launches.  mean / rstd: [groups, C].  -> (dx, dgamma, dbeta) with dgamma / dbeta summed
    over the groups; dgamma_out / dbeta_out: gradient-arena slots to ACCUMULATE into."""
    _chk(dy, 'dy'); _chk(x, 'x')
    C = x.shape[-1]
    rows_g = x.numel() // C // groups
    dx = torch.empty_like(x)
    gmask = torch.empty_like(x) if y is not None else None
    acc = dgamma_out is not None
    assert acc == (dbeta_out is not None)
    dgamma = dgamma_out if acc else torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = dbeta_out if acc else torch.empty(C, dtype=torch.float32, device=x.device)
    if _live(x):
        wsp, wsn = _ws_args(x, int(lib.t2i_bn_grouped_workspace_bytes(rows_g, C, groups)))
        twin = _twin_for(dx)
        check(lib.t2i_bn_bwd_grouped(_ptr(dy), _ptr(_chk(y, 'y') if y is not None else None), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(_chk(gamma)),
                                     rows_g, C, groups, act, alpha, _ptr(gmask), _ptr(dx), _ptr(twin), _ptr(dgamma), _ptr(dbeta), 1 if acc else 0,
                                     wsp, wsn, _same_dt(dy, y, x), _stream()), 't2i_bn_bwd_grouped')
        _twin_keep(dx, twin)
    return dx, dgamma, dbeta


_bn_apply_one = bn_apply          # (instrumentation replaces the public name; the per-slice calls above are not layers of their own)


def bn_bwd(dy, x, mean, rstd, gamma, sum_dy, sum_dy_x, dgamma_out=None, dbeta_out=None):
    """sum_dy_x = colsum(dy * (x - mean)) (col_reduce / act_bwd_colsum with center=mean).
    dgamma_out / dbeta_out: gradient-arena slots to ACCUMULATE into instead of fresh tensors."""
    _chk(dy, 'dy'); _chk(x, 'x')
    C = x.shape[-1]
    dx = torch.empty_like(x)
    acc = dgamma_out is not None
    assert acc == (dbeta_out is not None)
    dgamma = dgamma_out if acc else torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = dbeta_out if acc else torch.empty(C, dtype=torch.float32, device=x.device)
    if _live(x):
        wsp, wsn = _ws_args(x, 3 * C * 4)
        check(lib.t2i_bn_bwd(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(_chk(gamma)), _ptr(sum_dy), _ptr(sum_dy_x),
                             x.numel() // C, C, _ptr(dx), _ptr(dgamma), _ptr(dbeta), 1 if acc else 0, wsp, wsn, _stream()),
              't2i_bn_bwd')
    return dx, dgamma, dbeta


def act_fwd(x, act, alpha=0.2):
    _chk(x, 'x')
    y = torch.empty_like(x)
    if _live(x):
        twin = _twin_for(y, x)
        check(lib.t2i_act_fwd(_ptr(x), x.numel(), act, alpha, _ptr(y), _ptr(twin), _dt(x), _stream()), 't2i_act_fwd')
        _twin_keep(y, twin)
    return y


def act_bwd(dy, y, act, alpha=0.2, out=None):
    """out: an existing contiguous tensor (a slice of a stacked buffer) to write into; no bf16 twin then."""
    _chk(dy, 'dy'); _chk(y, 'y')
    if out is not None:
        assert out.shape == dy.shape and out.dtype == dy.dtype and out.is_contiguous()
        _drop_image(out)
    dx = out if out is not None else torch.empty_like(dy)
    if _live(dy):
        twin = _twin_for(dx, dy, y) if out is None else None
        check(lib.t2i_act_bwd(_ptr(dy), _ptr(y), dy.numel(), act, alpha, _ptr(dx), _ptr(twin), _same_dt(dy, y), _stream()), 't2i_act_bwd')
        _twin_keep(dx, twin)
    return dx


def act_bwd_colsum(dy, y, act, alpha=0.2, x2=None, out=None, center=None, dx_out=None):
    """-> (dx = dy*act'(y), colsum(dx)[, colsum(dx*(x2 - center))]) in one pass: a conv layer's activation backward + bias
    gradient, or (with x2 = layer input, center = its batch mean) the masked gradient and both reductions of the batch-norm backward.
    out: gradient-arena slot to ACCUMULATE colsum(dx) into (then the second return value is `out`)."""
    _chk(dy, 'dy'); _chk(y, 'y')
    C = dy.shape[-1]
    rows = dy.numel() // C
    if dx_out is not None:              # a slice of a stacked buffer to write dx into (no bf16 twin then)
        assert dx_out.shape == dy.shape and dx_out.dtype == dy.dtype and dx_out.is_contiguous()
        _drop_image(dx_out)
    dx = dx_out if dx_out is not None else torch.empty_like(dy)
    s = out if out is not None else torch.empty(C, dtype=torch.float32, device=dy.device)
    s2 = torch.empty(C, dtype=torch.float32, device=dy.device) if x2 is not None else None
    if x2 is not None:
        _chk(x2, 'x2')
        assert out is None
    if _live(dy):
        need = int(lib.t2i_col_reduce_workspace_bytes(rows, C))
        wsp, wsn = _ws_args(dy, need)
        twin = _twin_for(dx) if (x2 is None and dx_out is None) else None        # conv bias path: dx is the next input / filter gradient's operand
        check(lib.t2i_act_bwd_colsum(_ptr(dy), _ptr(y), _ptr(x2), _ptr(_chk(center, 'center') if center is not None else None), rows, C,
                                     act, alpha, _ptr(dx), _ptr(twin), _ptr(s), _ptr(s2),
                                     1 if out is not None else 0, wsp, wsn, _same_dt(dy, y, x2), _stream()), 't2i_act_bwd_colsum')
        _twin_keep(dx, twin)
    return (dx, s) if x2 is None else (dx, s, s2)


def add_act(a, b, act=ACT_NONE, alpha=0.2):
    _chk(a, 'a'); _chk(b, 'b')
    assert a.shape == b.shape
    y = torch.empty_like(a)
    if _live(a):
        twin = _twin_for(y, a, b)
        check(lib.t2i_add_act(_ptr(a), _ptr(b), a.numel(), act, alpha, _ptr(y), _ptr(twin), _same_dt(a, b), _stream()), 't2i_add_act')
        _twin_keep(y, twin)
    return y


def _drop_image(t):
    """A wrapper wrote `t` through its raw pointer: tensor._version does not move, so a bf16 image cached on it would be stale."""
    if t is not None:
        h = _image_holder(t)
        if getattr(h, '_t2i_h', None) is not None:
            h._t2i_h = None


def axpby(a, alpha, b=None, beta=0.0, out=None):
    """out: an existing tensor to overwrite (its cached bf16 image, if any, is dropped — see bf16_image)."""
    _chk(a, 'a')
    y = torch.empty_like(a) if out is None else out
    _drop_image(out)
    if _live(a):
        check(lib.t2i_axpby(_ptr(a), alpha, _ptr(_chk(b, 'b') if b is not None else None), beta, a.numel(), _ptr(y),
                            _same_dt(a, b, y), _stream()), 't2i_axpby')
    return y


def interp(eps, g, x, out=None):
    """out: an existing contiguous tensor of g's shape to write into (the x_hat slot of the stacked critic input)."""
    _chk(g, 'g', f32=True); _chk(x, 'x', f32=True)           # the 3-channel image side is float32 in every storage mode
    eps = _chk(eps.reshape(-1), 'eps', f32=True)
    B = g.shape[0]
    assert eps.numel() == B and g.shape == x.shape
    if out is not None:
        _chk(out, 'out', f32=True)
        assert out.shape == g.shape
        _drop_image(out)
    else:
        out = torch.empty_like(g)
    if _live(g):
        check(lib.t2i_interp(_ptr(eps), _ptr(g), _ptr(x), B, g.numel() // B, _ptr(out), _stream()), 't2i_interp')
    return out


def concat_tile_fwd(feat, emb):
    """feat [B,H,W,Cf], emb [B,Ce] -> [B,H,W,Cf+Ce]"""
    _chk(feat, 'feat'); _chk(emb, 'emb')
    B, H, W, Cf = feat.shape
    Ce = emb.shape[1]
    out = torch.empty((B, H, W, Cf + Ce), dtype=feat.dtype, device=feat.device)
    if _live(feat):
        check(lib.t2i_concat_tile_fwd(_ptr(feat), _ptr(emb), B, H * W, Cf, Ce, _ptr(out), _same_dt(feat, emb), _stream()), 't2i_concat_tile_fwd')
    return out


def concat_tile_bwd(dout, Cf, Ce):
    _chk(dout, 'dout')
    B, H, W, Ct = dout.shape
    assert Ct == Cf + Ce
    dfeat = torch.empty((B, H, W, Cf), dtype=dout.dtype, device=dout.device)
    demb = torch.empty((B, Ce), dtype=dout.dtype, device=dout.device)
    if _live(dout):
        check(lib.t2i_concat_tile_bwd(_ptr(dout), B, H * W, Cf, Ce, _ptr(dfeat), _ptr(demb), _dt(dout), _stream()), 't2i_concat_tile_bwd')
    return dfeat, demb


def nchw_to_nhwc(x):
    """x physically [B,C,H,W] contiguous -> physically [B,H,W,C] contiguous"""
    _chk(x, 'x')
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    if _live(x):
        check(lib.t2i_nchw_to_nhwc(_ptr(x), B, C, H * W, _ptr(y), _dt(x), _stream()), 't2i_nchw_to_nhwc')
    return y


def nhwc_to_nchw(x):
    _chk(x, 'x')
    B, H, W, C = x.shape
    y = torch.empty((B, C, H, W), dtype=x.dtype, device=x.device)
    if _live(x):
        check(lib.t2i_nhwc_to_nchw(_ptr(x), B, C, H * W, _ptr(y), _dt(x), _stream()), 't2i_nhwc_to_nchw')
    return y


def gp_slopes(g):
    _chk(g, 'g')
    B = g.shape[0]
    s = torch.empty(B, dtype=torch.float32, device=g.device)
    if _live(g):
        check(lib.t2i_gp_slopes(_ptr(g), B, g.numel() // B, _ptr(s), _dt(g), _stream()), 't2i_gp_slopes')
    return s


def row_scale(g, coef):
    _chk(g, 'g'); _chk(coef, 'coef', f32=True)
    B = g.shape[0]
    assert coef.numel() == B
    out = torch.empty_like(g)
    if _live(g):
        check(lib.t2i_row_scale(_ptr(g), _ptr(coef), B, g.numel() // B, _ptr(out), _dt(g), _stream()), 't2i_row_scale')
    return out


def row_scale_div(g, num, den):
    """out[b] = (den[b] > 0 ? num[b] / max(den[b], 1e-30) : 0) * g[b]: the slope norm's backward, coefficient formed in the kernel."""
    _chk(g, 'g'); _chk(num, 'num', f32=True); _chk(den, 'den', f32=True)
    B = g.shape[0]
    assert num.numel() == B and den.numel() == B
    out = torch.empty_like(g)
    if _live(g):
        check(lib.t2i_row_scale_div(_ptr(g), _ptr(num), _ptr(den), B, g.numel() // B, _ptr(out), _dt(g), _stream()), 't2i_row_scale_div')
    return out


def adam_tf(w, g, m, v, lr_t, beta1, beta2, eps=1e-8, grad_scale=1.0, lr_t_dev=None):
    """In place on flat arenas.  lr_t_dev: optional device scalar that overrides lr_t (graph replay)."""
    for t in (w, g, v) + ((m,) if m is not None else ()):
        _chk(t)
    assert w.numel() == g.numel() == v.numel() and (m is None or m.numel() == w.numel())
    assert m is not None or beta1 == 0.0, 'the first moment can be skipped only with beta1 == 0'
    if _live(w):
        check(lib.t2i_adam_tf(_ptr(w), _ptr(g), _ptr(m), _ptr(v), w.numel(), lr_t, _ptr(lr_t_dev), beta1, beta2, eps, grad_scale,
                              _stream()),
              't2i_adam_tf')


def device_info(device=0):
    cu, clk = ctypes.c_int32(0), ctypes.c_int32(0)
    arch = ctypes.create_string_buffer(64)
    check(lib.t2i_device_info(device, ctypes.byref(cu), ctypes.byref(clk), arch, 64), 't2i_device_info')
    return dict(cu_count=cu.value, clock_khz=clk.value, arch=arch.value.decode())


# ---- data pipeline (reference preprocess/dataset.py) --------------------------------------------------------------------
def crop_flip_normalize(src_u8, ids, row0, col0, flip, out_size):
    """src_u8 [N,S,S,3] uint8 (device); ids/row0/col0/flip int32 [B] (device) -> float32 [B,out_size,out_size,3]."""
    if src_u8.dtype != torch.uint8 or src_u8.dim() != 4 or src_u8.shape[3] != 3 or src_u8.shape[1] != src_u8.shape[2]:
        raise ValueError('crop_flip_normalize expects a uint8 [N,S,S,3] store, got %s %s' % (src_u8.dtype, tuple(src_u8.shape)))
    B = ids.numel()
    out = torch.empty((B, out_size, out_size, 3), dtype=torch.float32, device=src_u8.device)
    if _live(src_u8):
        a, b, c, d = ids.to(torch.int32).contiguous(), row0.to(torch.int32).contiguous(), col0.to(torch.int32).contiguous(), flip.to(torch.int32).contiguous()
        check(lib.t2i_crop_flip_normalize(_ptr(src_u8.contiguous()), src_u8.shape[0], src_u8.shape[1], _ptr(a), _ptr(b), _ptr(c),
                                          _ptr(d), B, out_size, _ptr(out), _stream()), 't2i_crop_flip_normalize')
    return out


def gather_mean(emb, ids, choice):
    """emb [N,En,D] float32; ids int32 [B]; choice int32 [B,k] -> [B,D] mean of the chosen rows, in choice order."""
    _chk(emb, 'emb')
    B, k = choice.shape
    out = torch.empty((B, emb.shape[2]), dtype=torch.float32, device=emb.device)
    if _live(emb):
        a, c = ids.to(torch.int32).contiguous(), choice.to(torch.int32).contiguous()
        check(lib.t2i_gather_mean(_ptr(emb), emb.shape[0], emb.shape[1], emb.shape[2], _ptr(a), _ptr(c), B, k, _ptr(out), _stream()),
              't2i_gather_mean')
    return out


# ---- PGGAN operators (reference utils/ops.py:74-81,100-101,109-111) ------------------------------------------------------
def pool2_sum(x, scale):
    """x [B,H,W,C] (H, W even) -> scale * 2x2 window sums [B,H/2,W/2,C]."""
    _chk(x, 'x')
    B, H, W, C = x.shape
    if H % 2 or W % 2:
        raise ValueError('pool(x, 2): odd extents %dx%d are not supported (the reference only pools powers of two)' % (H, W))
    y = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    if _live(x):
        check(lib.t2i_pool2_sum(_ptr(x), B, H, W, C, scale, _ptr(y), _stream()), 't2i_pool2_sum')
    return y


def upscale2(x, scale=1.0):
    """x [B,H,W,C] -> scale * nearest-neighbour x2 [B,2H,2W,C]."""
    _chk(x, 'x')
    B, H, W, C = x.shape
    y = torch.empty((B, 2 * H, 2 * W, C), dtype=torch.float32, device=x.device)
    if _live(x):
        check(lib.t2i_upscale2(_ptr(x), B, H, W, C, scale, _ptr(y), _stream()), 't2i_upscale2')
    return y


def row_moments(a, b=None):
    """a (and b) [B, ...] -> (sum over everything but axis 0 of a, of a*b or a*a)."""
    _chk(a, 'a')
    B = a.shape[0]
    s1 = torch.empty(B, dtype=torch.float32, device=a.device); s2 = torch.empty_like(s1)
    if _live(a):
        wsp, wsn = _ws_args(a, int(lib.t2i_row_moments_workspace_bytes(B)))
        check(lib.t2i_row_moments(_ptr(a), _ptr(_chk(b, 'b') if b is not None else None), B, a.numel() // B, _ptr(s1), _ptr(s2),
                                  wsp, wsn, _stream()), 't2i_row_moments')
    return s1, s2


def row_fma2(a, alpha, b=None, gamma=None, delta=None):
    """out[r,...] = a[r,...]*alpha[r] + b[r,...]*gamma[r] + delta[r]   (per-sample scalars; b/gamma and delta optional)."""
    _chk(a, 'a')
    B = a.shape[0]
    out = torch.empty_like(a)
    if _live(a):
        al = _chk(alpha.reshape(-1), 'alpha')
        ga = _chk(gamma.reshape(-1), 'gamma') if gamma is not None else None
        de = _chk(delta.reshape(-1), 'delta') if delta is not None else None
        assert al.numel() == B and (ga is None or ga.numel() == B) and (de is None or de.numel() == B)
        check(lib.t2i_row_fma2(_ptr(a), _ptr(_chk(b, 'b') if b is not None else None), _ptr(al), _ptr(ga), _ptr(de), B,
                               a.numel() // B, _ptr(out), _stream()), 't2i_row_fma2')
    return out


# ---- loss heads (reference models/wgancls/model.py:72-92,117-127) -------------------------------------------------------
D_HEAD_KEYS = ('D_loss', 'D_loss_real', 'D_loss_fake', 'D_loss_mismatch', 'wdist', 'wdist2', 'real_gp', 'real_gp2', 'reg_loss',
               'balance_loss', 'kt_grad', 'kt')


def wgan_d_head(logits, slopes1, slopes2, kt, gp_coeff, seed_l_into=None):
    """logits [3B] (fake | real | mismatch), slopes [B], kt: device scalar tensor or None (= 1).
    -> (scalars [12] in D_HEAD_KEYS order, dD/dlogits [3B], dD/dslopes1 [B], dD/dslopes2 [B]).  dD/dlogits depends on kt and B only
    (1/B | -(1+kt)/B | kt/B): the stacked critic step calls the head once BEFORE the slopes exist, for these seeds alone
    (seed_l_into: a contiguous float32 [3B] buffer that receives them), and once after, for the scalars and the slope seeds."""
    _chk(logits, 'logits'); _chk(slopes1, 'slopes1'); _chk(slopes2, 'slopes2')
    B = slopes1.numel()
    assert logits.numel() == 3 * B and slopes2.numel() == B
    scal = torch.empty(12, dtype=torch.float32, device=logits.device)
    if seed_l_into is not None:
        _chk(seed_l_into, 'seed_l_into', f32=True)
        assert seed_l_into.numel() == 3 * B
    sl = seed_l_into if seed_l_into is not None else torch.empty_like(logits)
    s1, s2 = torch.empty_like(slopes1), torch.empty_like(slopes2)
    if _live(logits):
        check(lib.t2i_wgan_d_head(_ptr(logits), _ptr(slopes1), _ptr(slopes2), _ptr(kt), B, gp_coeff, _ptr(sl), _ptr(s1), _ptr(s2),
                                  _ptr(scal), _stream()), 't2i_wgan_d_head')
    return scal, sl, s1, s2


def sigmoid_ce_head(logits, labels, weights, want_prob=True, seeds_into=None):
    """logits: 1-3 tensors of B logits each; labels / weights: one float per head.  -> (losses [4]: total, per head; seeds: list of
    d total / d logits_k; probs: list of sigmoid(logits_k) or None)  — t2i_sigmoid_ce_head, one launch.
    seeds_into: a flat float32 tensor of n*B elements whose slices receive the seeds (the heads of ONE batched pass)."""
    n = len(logits)
    assert 1 <= n <= 3 and len(labels) == n and len(weights) == n
    for t in logits:
        _chk(t, 'logits', f32=True)
    B = logits[0].numel()
    assert all(t.numel() == B and t.is_contiguous() for t in logits)
    losses = torch.empty(4, dtype=torch.float32, device=logits[0].device)
    if seeds_into is not None:
        _chk(seeds_into, 'seeds_into', f32=True)
        assert seeds_into.numel() == n * B
        flat = seeds_into.view(-1)
        seeds = [flat[k * B:(k + 1) * B] for k in range(n)]
    else:
        seeds = [torch.empty_like(t) for t in logits]
    probs = [torch.empty_like(t) for t in logits] if want_prob else None
    if _live(logits[0]):
        pad = lambda xs, fill: list(xs) + [fill] * (3 - n)
        lp = pad([_ptr(t) for t in logits], None)
        sp = pad([_ptr(t) for t in seeds], None)
        pp = pad([_ptr(t) for t in probs], None) if want_prob else [None] * 3
        y, w = pad([float(v) for v in labels], 0.0), pad([float(v) for v in weights], 0.0)
        check(lib.t2i_sigmoid_ce_head(lp[0], lp[1], lp[2], y[0], y[1], y[2], w[0], w[1], w[2], B, sp[0], sp[1], sp[2], pp[0], pp[1], pp[2],
                                      _ptr(losses), _stream()), 't2i_sigmoid_ce_head')
    return losses, seeds, probs


def ca_kl_fwd(mean, log_sigma, eps):
    _chk(mean, 'mean', f32=True); _chk(log_sigma, 'log_sigma', f32=True); _chk(eps, 'eps', f32=True)      # [B, 128]: kernels.f32_outputs()
    code = torch.empty_like(mean)
    kl = torch.empty(1, dtype=torch.float32, device=mean.device)
    if _live(mean):
        check(lib.t2i_ca_kl_fwd(_ptr(mean), _ptr(log_sigma), _ptr(eps), mean.numel(), _ptr(code), _ptr(kl), _stream()), 't2i_ca_kl_fwd')
    return code, kl


def ca_kl_bwd(mean, log_sigma, eps, dcode, dkl):
    dmean, dls = torch.empty_like(mean), torch.empty_like(mean)
    if _live(mean):
        check(lib.t2i_ca_kl_bwd(_ptr(mean), _ptr(log_sigma), _ptr(eps), _ptr(_chk(dcode, 'dcode') if dcode is not None else None),
                                _ptr(_chk(dkl, 'dkl') if dkl is not None else None), mean.numel(), _ptr(dmean), _ptr(dls),
                                _stream()), 't2i_ca_kl_bwd')
    return dmean, dls


ALGO_NAMES = ('implicit_gemm', 'winograd_f2x2_3x3', 'winograd_f2x2_2x2', 'direct_small', 'implicit_gemm_bf16_operands')
ALGO_MAC_RATIO = (1.0, 1.0 / 2.25, 9.0 / 16.0, 1.0, 1.0)       # executed / direct-convolution multiply-adds


def conv_algo(d, which):
    """Algorithm the library picks for descriptor `d`; which: 'fwd' | 'bwd_data' | 'bwd_filter'."""
    a = int(lib.t2i_conv2d_algo(ctypes.byref(d), ('fwd', 'bwd_data', 'bwd_filter').index(which)))
    if a < 0:
        raise ValueError('invalid conv descriptor')
    return ALGO_NAMES[a]


_FC_ARENA = [None]
_FC_ON = [False]


def filter_cache(on, device=None):
    """Opt into the transformed-filter cache of the Winograd conv paths (include/t2i_hip.h: contract).  The library owns no
    device memory: the first call allocates the arena the transforms live in (T2I_FILTER_CACHE_MB, default 1024; the
    wgancls step needs 650 MB) and attaches it.  Everything in this package that writes filter memory outside t2i_adam_tf
    calls filter_cache_invalidate(): ParamStore.load, Saver.restore, optim.Arena creation, dp.broadcast_variables, hipGraph
    replays.  Returns the previous state."""
    if on and _FC_ARENA[0] is None and torch.cuda.is_available():
        import os
        dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        buf = torch.empty(int(os.environ.get('T2I_FILTER_CACHE_MB', '1024')) << 20, dtype=torch.uint8, device=dev)
        check(lib.t2i_filter_cache_attach(_ptr(buf), buf.numel()), 't2i_filter_cache_attach')
        _FC_ARENA[0] = buf           # kept for the life of the process: captured graphs point into it
    _FC_ON[0] = bool(on) and _FC_ARENA[0] is not None
    return bool(lib.t2i_filter_cache_enable(1 if on else 0))


def filter_cache_enabled():
    """True while the transformed-filter cache is on through filter_cache() (and has its arena)."""
    return _FC_ON[0]


def filter_cache_reset():
    """Drop every cached filter image and hand the library the same arena again, empty.  Legal only when no captured graph that
    used the cache is alive any more (bench.py calls it between two configurations, after the first model and its graphs are gone):
    slots are never moved or reused while attached, so a process that builds model after model would otherwise fill the arena with
    images of filters that no longer exist."""
    if _FC_ARENA[0] is not None:
        buf = _FC_ARENA[0]
        check(lib.t2i_filter_cache_attach(None, 0), 't2i_filter_cache_attach')
        check(lib.t2i_filter_cache_attach(_ptr(buf), buf.numel()), 't2i_filter_cache_attach')


def tuning_set(key, value):
    """Planner / diagnostic switch (include/t2i_hip.h t2i_tuning_set); drops cached descriptors, whose workspace sizes
    were computed under the old setting."""
    check(lib.t2i_tuning_set(key.encode(), float(value)), 't2i_tuning_set')
    _DESC_CACHE.clear()
    _H_ALGO.clear()
    _XFORM_BYTES.clear()


def zero_ranges(base, table):
    """base[start:start + length] = 0 for every (start, length) row of the device int64 table [n, 2], one launch (t2i_zero_ranges)."""
    _chk(base, 'base', f32=True)
    assert table.dtype == torch.int64 and table.dim() == 2 and table.shape[1] == 2 and table.is_contiguous() and table.device == base.device
    if _live(base) and table.shape[0]:
        check(lib.t2i_zero_ranges(_ptr(base), _ptr(table), int(table.shape[0]), _stream()), 't2i_zero_ranges')


def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
    """In place: t = mean + std * n with n ~ N(0,1) truncated to [a, b] (a, b in units of the standard normal: the reference's
    tf.truncated_normal cuts at +-2 standard deviations; with mean 0, std 1 these are torch.nn.init.trunc_normal_'s arguments).
    One launch (t2i_trunc_normal) instead of the tensor library's eight.  The Philox (seed, offset) come from — and advance — the
    device's torch generator, so torch.manual_seed / torch.cuda.manual_seed make the draws reproducible like any other."""
    _chk(t, 't', f32=True)
    if _live(t):
        gen = torch.cuda.default_generators[t.device.index if t.device.index is not None else torch.cuda.current_device()]
        seed, off = int(gen.initial_seed()) & 0xFFFFFFFFFFFFFFFF, int(gen.get_offset())
        quads = (t.numel() + 3) // 4
        gen.set_offset(off + 4 * ((quads + 3) // 4))              # (torch wants multiples of 4)
        check(lib.t2i_trunc_normal(_ptr(t), t.numel(), seed, off, float(mean), float(std), float(a), float(b), _stream()), 't2i_trunc_normal')
    return t


def kt_sgd(kt, wdist_sums, scale, lr):
    """kt -= lr * d balance_loss / d kt from the (rank-summed) batch means wdist, wdist2; in place on the device scalar."""
    _chk(wdist_sums, 'wdist_sums')
    assert wdist_sums.numel() == 2 and kt.numel() == 1
    if _live(wdist_sums):
        check(lib.t2i_kt_sgd(_ptr(kt), _ptr(wdist_sums), scale, lr, _stream()), 't2i_kt_sgd')


_FILTER_EPOCH = [0]


def filter_epoch():
    """Counts the filter writes announced from OUTSIDE a training step (checkpoint loads, broadcasts, new arenas ...): a captured
    iteration that trusts the images its previous replay left (filter_cache_assume) regenerates them eagerly when this moved."""
    return _FILTER_EPOCH[0]


def filter_cache_invalidate(t=None, external=True):
    """Drop cached transforms of the filters inside tensor `t` (None: all).  external=False: the caller is a training step's own
    replay, which left the images of the arenas it regenerates behind its updates valid in memory (only the host's bookkeeping is
    stale); every other writer of filter memory leaves the default, which also moves filter_epoch()."""
    if external:
        _FILTER_EPOCH[0] += 1
    if t is None:
        lib.t2i_filter_cache_invalidate(None, 0)
    elif t.is_cuda:
        lib.t2i_filter_cache_invalidate(_ptr(t), t.numel() * t.element_size())


def filter_cache_bytes():
    return int(lib.t2i_filter_cache_bytes())


def filter_cache_refresh(t=None):
    """Regenerate, in one launch, every cached filter image (of the filters inside tensor `t`; None = all, legal only while every
    filter the cache has seen is still allocated) that is stale in the current launch context.  t2i_adam_tf does this for its arena; graphs.StepGraphs.capture does it at the head of every graph,
    so that a graph holds ONE batched refresh instead of one small fill per filter at its first use."""
    if t is None:
        check(lib.t2i_filter_cache_refresh(None, 0, _stream()), 't2i_filter_cache_refresh')
    else:
        check(lib.t2i_filter_cache_refresh(_ptr(t), t.numel() * t.element_size(), _stream()), 't2i_filter_cache_refresh')


def filter_cache_assume(t):
    """Mark the cached images of the filters inside tensor `t` as filled for the current launch context without regenerating them
    (include/t2i_hip.h t2i_filter_cache_assume: the caller vouches that they will be current whenever the following work runs)."""
    check(lib.t2i_filter_cache_assume(_ptr(t), t.numel() * t.element_size(), _stream()), 't2i_filter_cache_assume')


def lerp_dev(a, b, t_dev, mode=0):
    """mode 0: (1-t)*a + t*b;  1: t*a;  2: (1-t)*a   with t a 1-element device tensor."""
    _chk(a, 'a')
    out = torch.empty_like(a)
    if _live(a):
        check(lib.t2i_lerp_dev(_ptr(a), _ptr(_chk(b, 'b') if b is not None else None), _ptr(_chk(t_dev, 't')), mode, a.numel(), _ptr(out),
                               _stream()), 't2i_lerp_dev')
    return out
