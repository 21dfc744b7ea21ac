#!/usr/bin/env python
"""The bf16 LDS-DMA GEMM (igemm_hd_kernel) as a plain GEMM (1x1 conv on bf16 tensors, bf16 output) against torch.mm in bf16
(hipBLASLt / rocBLAS behind torch) on the same shapes, same graph-replay timer.  M = pixels, N = output channels, K = input channels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import t2i_amd  # noqa
from t2i_amd import kernels as K
from tools.bench_conv import timeit
K.set_math('bf16'); K.set_storage('bf16')
K.workspace(torch.device('cuda', 0), 1 << 30)
print('%-30s %10s %8s %12s %8s' % ('M x N x K', 't2i us', 'TF/s', 'torch.mm us', 'TF/s'))
for M, N, Kd in ((16384, 256, 512), (16384, 256, 2304), (16384, 256, 4096), (4096, 512, 4608), (1024, 1024, 8192), (65536, 128, 1152), (65536, 256, 2048),
                 (131072, 256, 2304), (32768, 512, 4608), (8192, 1024, 4608), (8192, 8192, 8192)):
    B = M // 256
    d, ws = K.conv_desc(B, 16, 16, Kd, N, 1, 1, 1, 1, 'VALID')
    x = torch.randn(B, 16, 16, Kd, device='cuda').bfloat16()
    w = torch.randn(1, 1, Kd, N, device='cuda') * 0.05
    t1 = timeit(lambda: K.conv_fwd(x, w, None, d, 1 << 30), 10) * 1e6
    a = x.reshape(M, Kd); b = w.reshape(Kd, N).bfloat16().contiguous(); out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    t2 = timeit(lambda: torch.mm(a, b, out=out), 10) * 1e6
    fl = 2.0 * M * N * Kd
    print('%-30s %10.1f %8.1f %12.1f %8.1f' % ('%d x %d x %d' % (M, N, Kd), t1, fl / t1 / 1e6, t2, fl / t2 / 1e6))
