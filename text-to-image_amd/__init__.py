"""text-to-image_amd: the wgancls / gancls training hot path of crisbodnar/text-to-image, MI355X-native.

Layout (mirrors the reference files it replaces):
  csrc/ + lib/libt2i_hip.so   hand-written gfx950 HIP kernels behind the C ABI of include/t2i_hip.h
  _lib.py, kernels.py         ctypes loader (fails loudly without the .so) and thin tensor-level wrappers
  autograd.py                 double-differentiable torch.autograd.Functions built only from those kernels
  scope.py                    TF-1.x style variable scopes / auto-naming / initializers
  utils/ops.py                the reference's operator surface (utils/ops.py) — conv2d, conv2d_transpose, fc, batch_norm ...
  models/wgancls/             WGanCls model, WGanClsTrainer, run.py, cfg/flowers.yml (reference models/wgancls/*)
  optim.py, dp.py             TF-flavoured Adam over flat arenas; data-parallel gradient all-reduce (RCCL)
"""
__version__ = '0.1.0'
