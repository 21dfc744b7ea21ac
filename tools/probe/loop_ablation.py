#!/usr/bin/env python
"""us per 128x128x64 K-tile of igemm_hd_kernel (4-wave, two-phase loop: run with T2I_BF16_WAVES=4) for timing-only builds:
    for v in 8 16 24 64; do HIPCC="/opt/rocm/bin/hipcc -DT2I_HEXP=$v" bash text-to-image_amd/csrc/build.sh $PWD/tools/probe/libs/x$v; done
T2I_HEXP bits: 8 fragments read once, 16 no DMA in the loop, 64 no first barrier (1 / 2: no epilogue stores / no K loop).  The "cheap
addressing" build of profiles/r04_bf16_loop_ablation.txt (bit 4) went away with the loader rewrite it motivated.  The libraries
compute wrong results and are never shipped (tools/probe/libs/ is git-ignored).  One subprocess per library; the environment
(T2I_BF16_WAVES, T2I_BF16_DMA, T2I_BF16_PAIR_TILES) is passed through."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch, t2i_amd
from t2i_amd import kernels as K
from t2i_amd._lib import lib
from tools.bench_conv import timeit
K.set_math('bf16'); K.set_storage('bf16')
K.workspace(torch.device('cuda', 0), 1 << 30)
def run(M, N, Kd, k=1):
    B = M // 256
    d, ws = K.conv_desc(B, 16, 16, Kd, N, k, k, 1, 1, 'SAME')
    x = torch.randn(B, 16, 16, Kd, device='cuda').bfloat16()
    w = torch.randn(k, k, Kd, N, device='cuda') * 0.05
    lib.t2i_tuning_set(b'force_tile', 22.0); lib.t2i_tuning_set(b'force_splitk', 1.0)
    t = timeit(lambda: K.conv_fwd(x, w, None, d, 1 << 30), 20)
    return t * 1e6
out = []
for M in (16384, 2048):
    a, b = run(M, 256, 512), run(M, 256, 4096)
    out.append('M=%%5d: K512 %%6.1f K4096 %%6.1f -> %%.3f us/K-tile' %% (M, a, b, (b - a) / 56))
a, b = run(16384, 256, 128, 3), run(16384, 256, 512, 3)
out.append('3x3 M=16384: Cin128 %%6.1f Cin512 %%6.1f -> %%.3f us/K-tile' %% (a, b, (b - a) / 54))
print(' | '.join(out))
''' % ROOT
libs = sys.argv[1:] or ['default', 'x8', 'x16', 'x24', 'x64']
for name in libs:
    env = dict(os.environ)
    if name != 'default':
        env['T2I_HIP_LIB'] = os.path.join(ROOT, 'tools', 'probe', 'libs', name, 'libt2i_hip.so')
        if not os.path.exists(env['T2I_HIP_LIB']):
            print('%-8s (not built)' % name)
            continue
    r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=300)
    print('%-8s %s' % (name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'FAILED ' + r.stderr[-300:]))
    sys.stdout.flush()
