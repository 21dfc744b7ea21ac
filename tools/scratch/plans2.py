import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['T2I_DEBUG_PLAN'] = '1'
import torch, t2i_amd
from t2i_amd import kernels as K
from tools.bench_conv import LAYERS
L = {l[0]: l for l in LAYERS}
for name, B in [('D2',192),('D2',64),('D3',192),('D4',192),('G8c',64),('G8dc',64),('D10',64),('D10',192),('D4',64),('G5dc',64),('D7',192)]:
    _, H, W, Ci, Co, k, s, pad = L[name]
    d, ws = K.conv_desc(B, H, W, Ci, Co, k, k, s, s, pad)
    x = torch.randn(B, H, W, Ci, device='cuda'); w = torch.randn(k, k, Ci, Co, device='cuda') * 0.05
    dy = torch.randn(B, d.Ho, d.Wo, Co, device='cuda')
    for mode, fn in (('fwd', lambda: K.conv_fwd(x, w, None, d, ws)), ('bwdD', lambda: K.conv_bwd_data(dy, w, None, d, ws)), ('bwdF', lambda: K.conv_bwd_filter(x, dy, d, ws))):
        sys.stderr.write('## %s B=%d %s\n' % (name, B, mode)); sys.stderr.flush()
        fn(); torch.cuda.synchronize()
