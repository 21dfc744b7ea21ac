"""ctypes binding of libt2i_hip.so (C ABI: include/t2i_hip.h).  There is NO fallback: if the shared library is
missing or a symbol is absent, importing this module raises — the product path never silently computes elsewhere."""
import ctypes
import os

# The HIP runtime must come into the process through PyTorch first: libt2i_hip.so links against libamdhip64.so.7 by SONAME, and
# PyTorch ships its own copy.  Loaded in the other order (`from t2i_amd import _lib` before anything imported torch) the loader binds
# torch to /opt/rocm's runtime as well, and torch's device enumeration and this library's launches then disagree about the device
# ("no ROCm-capable device is detected" at the first launch: __graft_entry__.py run as a script did exactly that).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('T2I_HIP_LIB', os.path.join(_HERE, 'lib', 'libt2i_hip.so'))


class ConvDesc(ctypes.Structure):
    """t2i_conv_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in
                ('B', 'H', 'W', 'Cin', 'Ho', 'Wo', 'Cout', 'KH', 'KW', 'SH', 'SW', 'pad_t', 'pad_l', 'math')]


class ConvOpts(ctypes.Structure):
    """t2i_conv_opts: optional side inputs / outputs of one conv call (include/t2i_hip.h)"""
    _fields_ = [('a_image', ctypes.c_void_p), ('b_image', ctypes.c_void_p), ('out_image', ctypes.c_void_p), ('xform', ctypes.c_void_p),
                ('xform_bytes', ctypes.c_size_t), ('xform_mode', ctypes.c_int32), ('out_image_written', ctypes.c_int32),
                ('xform_kept', ctypes.c_int32), ('in_dtype', ctypes.c_int32), ('out_dtype', ctypes.c_int32), ('xform_valid_rows', ctypes.c_int32), ('xform_plane_rows', ctypes.c_int32), ('reserved', ctypes.c_int32)]


XFORM_NONE, XFORM_KEEP, XFORM_HAVE = 0, 1, 2
DT_F32, DT_BF16 = 0, 1           # t2i_dtype: element type of the activation tensors of a call

_p = ctypes.c_void_p
_i32, _i64, _f, _sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
_dp = ctypes.POINTER(ConvDesc)
_op = ctypes.POINTER(ConvOpts)

# name -> (restype, argtypes); must list every symbol include/t2i_hip.h declares (tests/test_abi.py checks that)
SIGNATURES = {
    't2i_version': (ctypes.c_int, []),
    't2i_last_error': (ctypes.c_char_p, []),
    't2i_device_info': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_i32), ctypes.POINTER(_i32), ctypes.c_char_p, _sz]),
    't2i_conv2d_workspace_bytes': (_sz, [_dp]),
    't2i_conv2d_fwd': (ctypes.c_int, [_dp, _p, _p, _p, _p, ctypes.c_int, _f, _op, _p, _sz, _p]),
    't2i_conv2d_bwd_data': (ctypes.c_int, [_dp, _p, _p, _p, _p, ctypes.c_int, _f, _op, _p, _sz, _p]),
    't2i_conv2d_bwd_filter': (ctypes.c_int, [_dp, _p, _p, _p, ctypes.c_int, _op, _p, _sz, _p]),
    't2i_col_reduce_workspace_bytes': (_sz, [_i64, _i32]),
    't2i_col_reduce': (ctypes.c_int, [_p, _p, _p, _i64, _i32, _p, _p, ctypes.c_int, _p, _sz, _i32, _p]),
    't2i_bn_stats': (ctypes.c_int, [_p, _i64, _i32, _p, _p, _p, _sz, _p]),
    't2i_bn_stats_tiles': (ctypes.c_int, [_p, _p, _i32, _i32, _i64, _i32, _p, _p, _p]),
    't2i_bn_train_fwd_stats': (ctypes.c_int, [_p, _p, _p, _i32, _i32, _i64, _i32, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _sz, _i32, _p]),
    't2i_bn_bwd_fused_workspace_bytes': (_sz, [_i64, _i32]),
    't2i_bn_bwd_fused': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, ctypes.c_int, _f, _p, _p, _p, _p, _p, ctypes.c_int, _p, _sz, _i32, _p]),
    't2i_bn_finalize': (ctypes.c_int, [_p, _p, _i64, _i32, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p]),
    't2i_bn_apply': (ctypes.c_int, [_p, _p, _p, _i64, _i32, ctypes.c_int, _f, _p, _p, _i32, _p]),
    't2i_bn_bwd': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, ctypes.c_int, _p, _sz, _p]),
    't2i_act_fwd': (ctypes.c_int, [_p, _i64, ctypes.c_int, _f, _p, _p, _i32, _p]),
    't2i_act_bwd': (ctypes.c_int, [_p, _p, _i64, ctypes.c_int, _f, _p, _p, _i32, _p]),
    't2i_act_bwd_colsum': (ctypes.c_int, [_p, _p, _p, _p, _i64, _i32, ctypes.c_int, _f, _p, _p, _p, _p, ctypes.c_int, _p, _sz, _i32, _p]),
    't2i_add_act': (ctypes.c_int, [_p, _p, _i64, ctypes.c_int, _f, _p, _p, _i32, _p]),
    't2i_axpby': (ctypes.c_int, [_p, _f, _p, _f, _i64, _p, _i32, _p]),
    't2i_interp': (ctypes.c_int, [_p, _p, _p, _i32, _i64, _p, _p]),
    't2i_concat_tile_fwd': (ctypes.c_int, [_p, _p, _i32, _i32, _i32, _i32, _p, _i32, _p]),
    't2i_concat_tile_bwd': (ctypes.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p, _i32, _p]),
    't2i_nchw_to_nhwc': (ctypes.c_int, [_p, _i32, _i32, _i32, _p, _i32, _p]),
    't2i_nhwc_to_nchw': (ctypes.c_int, [_p, _i32, _i32, _i32, _p, _i32, _p]),
    't2i_gp_slopes': (ctypes.c_int, [_p, _i32, _i64, _p, _i32, _p]),
    't2i_row_scale': (ctypes.c_int, [_p, _p, _i32, _i64, _p, _i32, _p]),
    't2i_row_scale_div': (ctypes.c_int, [_p, _p, _p, _i32, _i64, _p, _i32, _p]),
    't2i_adam_tf': (ctypes.c_int, [_p, _p, _p, _p, _i64, _f, _p, _f, _f, _f, _f, _p]),
    't2i_wgan_d_head': (ctypes.c_int, [_p, _p, _p, _p, _i32, _f, _p, _p, _p, _p, _p]),
    't2i_bn_grouped_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    't2i_bn_train_fwd_grouped': (ctypes.c_int, [_p, _i64, _i32, _i32, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, ctypes.c_int, _f, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _sz, _i32, _p]),
    't2i_bn_bwd_grouped': (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, ctypes.c_int, _f, _p, _p, _p, _p, _p, ctypes.c_int, _p, _sz, _i32, _p]),
    't2i_sigmoid_ce_head': (ctypes.c_int, [_p, _p, _p, _f, _f, _f, _f, _f, _f, _i32, _p, _p, _p, _p, _p, _p, _p, _p]),
    't2i_ca_kl_fwd': (ctypes.c_int, [_p, _p, _p, _i64, _p, _p, _p]),
    't2i_ca_kl_bwd': (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _p, _p, _p]),
    't2i_lerp_dev': (ctypes.c_int, [_p, _p, _p, _i32, _i64, _p, _p]),
    't2i_conv2d_algo': (ctypes.c_int, [_p, _i32]),
    't2i_stat': (ctypes.c_longlong, [ctypes.c_char_p]),
    't2i_conv2d_bwd_pair': (ctypes.c_int, [_dp, ctypes.c_int, _p, _p, _p, _op, _p, _p, _p, ctypes.c_int, _op, _p, _sz, _p, _sz, _p]),
    't2i_filter_cache_attach': (ctypes.c_int, [_p, _sz]),
    't2i_filter_cache_enable': (ctypes.c_int, [ctypes.c_int]),
    't2i_tuning_set': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_double]),
    't2i_kt_sgd': (ctypes.c_int, [_p, _p, _f, _f, _p]),
    't2i_zero_ranges': (ctypes.c_int, [_p, _p, _i32, _p]),
    't2i_trunc_normal': (ctypes.c_int, [_p, _i64, ctypes.c_uint64, ctypes.c_uint64, _f, _f, _f, _f, _p]),
    't2i_filter_cache_invalidate': (None, [_p, ctypes.c_size_t]),
    't2i_filter_cache_bytes': (ctypes.c_size_t, []),
    't2i_filter_cache_refresh': (ctypes.c_int, [_p, _sz, _p]),
    't2i_filter_cache_assume': (ctypes.c_int, [_p, _sz, _p]),
    't2i_cast_bf16': (ctypes.c_int, [_p, _i64, _p, _p]),
    't2i_cast_f32': (ctypes.c_int, [_p, _i64, _p, _p]),
    't2i_conv2d_input_transform_bytes': (ctypes.c_size_t, [_dp]),
    't2i_capture_id': (ctypes.c_uint64, [_p]),
    't2i_conv2d_stats_bytes': (ctypes.c_size_t, [_dp]),
    't2i_conv2d_fwd_stats': (ctypes.c_int, [_dp, _p, _p, _p, _p, ctypes.c_int, _f, _p, _sz, ctypes.POINTER(ctypes.c_int32),
                                            ctypes.POINTER(ctypes.c_int32), _op, _p, _sz, _p]),
    't2i_col_reduce_partials': (ctypes.c_int, [_p, _p, _i32, _i32, _p, _p, ctypes.c_int, _p]),
    't2i_pool2_sum': (ctypes.c_int, [_p, _i32, _i32, _i32, _i32, _f, _p, _p]),
    't2i_upscale2': (ctypes.c_int, [_p, _i32, _i32, _i32, _i32, _f, _p, _p]),
    't2i_row_moments_workspace_bytes': (ctypes.c_size_t, [_i32]),
    't2i_row_moments': (ctypes.c_int, [_p, _p, _i32, _i64, _p, _p, _p, _sz, _p]),
    't2i_row_fma2': (ctypes.c_int, [_p, _p, _p, _p, _p, _i32, _i64, _p, _p]),
    't2i_crop_flip_normalize': (ctypes.c_int, [_p, _i64, _i32, _p, _p, _p, _p, _i32, _i32, _p, _p]),
    't2i_gather_mean': (ctypes.c_int, [_p, _i64, _i32, _i32, _p, _p, _i32, _i32, _p, _p]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError('libt2i_hip.so not found at %s — build it with text-to-image_amd/csrc/build.sh '
                      '(or `python -c "import __graft_entry__ as g; g.build()"`); there is no CPU fallback' % LIB_PATH)

ABI_VERSION = 9          # include/t2i_hip.h T2I_ABI_VERSION: argument lists changed in v5, v6 and v7 — symbols alone do not tell

lib = ctypes.CDLL(LIB_PATH)
try:
    lib.t2i_version.restype = ctypes.c_int
    _got = lib.t2i_version()
except AttributeError:
    _got = None
if _got != ABI_VERSION:
    raise ImportError('%s reports ABI version %r, this package binds version %d: a stale build would be called with shifted '
                      'arguments — rebuild with text-to-image_amd/csrc/build.sh' % (LIB_PATH, _got, ABI_VERSION))
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here == ABI mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


class T2IError(RuntimeError):
    pass


def check(rc, what=''):
    if rc != 0:
        raise T2IError('%s failed (%d): %s' % (what, rc, lib.t2i_last_error().decode('utf-8', 'replace')))
