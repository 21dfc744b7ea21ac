import os, sys
sys.path.insert(0, '/root/repo')
import torch, t2i_amd
from t2i_amd import kernels as K
from tools.bench_conv import LAYERS, timeit
L = {l[0]: l for l in LAYERS}
for name, B, mode, tile, sk in [('D4', 64, 'bwdF', 22, 1), ('G8c', 64, 'fwd', 22, 1), ('D2', 192, 'bwdD', 22, 1), ('D4', 192, 'fwd', 21, 2)]:
    _, H, W, Ci, Co, k, s, pad = L[name]
    d, ws = K.conv_desc(B, H, W, Ci, Co, k, k, s, s, pad)
    x = torch.randn(B, H, W, Ci, device='cuda'); w = torch.randn(k, k, Ci, Co, device='cuda') * 0.05
    dy = torch.randn(B, d.Ho, d.Wo, Co, device='cuda')
    big = 1 << 30; K.workspace(torch.device('cuda', 0), big)
    fn = {'fwd': lambda: K.conv_fwd(x, w, None, d, big), 'bwdD': lambda: K.conv_bwd_data(dy, w, None, d, big), 'bwdF': lambda: K.conv_bwd_filter(x, dy, d, big)}[mode]
    os.environ['T2I_FORCE_TILE'] = str(tile); os.environ['T2I_FORCE_SPLITK'] = str(sk)
    fl = K.conv_flops(d)
    out = []
    for ab, label in [(0, 'full'), (1, 'no-gload'), (3, 'no-gload/store'), (7, 'no-gload/store/barrier'), (15, 'mfma-only'), (4, 'no-barrier'), (8, 'no-fragreads'), (2, 'no-store')]:
        os.environ['T2I_ABLATE'] = str(ab)
        t = timeit(fn, 10)
        out.append('%s %.1fus %.0fTF' % (label, t * 1e6, fl / t / 1e12))
    print(name, B, mode, 't%d/s%d' % (tile, sk), ' | '.join(out))
