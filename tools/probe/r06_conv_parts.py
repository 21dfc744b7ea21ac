"""each conv primitive on a 4B batch against the same primitive on its [3B | B] parts (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import t2i_amd  # noqa
from t2i_amd import kernels as K
nf = int(os.environ.get('WIDTH', '32'))
b = int(os.environ.get('B', '8'))
L = [('Conv', 64, 3, nf, 4, 2, 'SAME'), ('Conv_1', 32, nf, 2 * nf, 4, 2, 'SAME'), ('Conv_2', 16, 2 * nf, 4 * nf, 4, 2, 'SAME'), ('Conv_3', 8, 4 * nf, 8 * nf, 4, 2, 'SAME'),
     ('Conv_4', 4, 8 * nf, 2 * nf, 1, 1, 'VALID'), ('Conv_5', 4, 2 * nf, 4 * nf, 3, 1, 'SAME'), ('Conv_6', 4, 4 * nf, 8 * nf, 3, 1, 'SAME'),
     ('Conv_7', 4, 8 * nf + 128, 8 * nf, 3, 1, 'SAME'), ('Conv_8', 4, 8 * nf, 8 * nf, 1, 1, 'VALID'), ('Conv_9', 4, 8 * nf, 1, 4, 4, 'VALID')]
g = torch.Generator(device='cuda').manual_seed(0)
def rel(a, r):
    return float((a.double() - r.double()).abs().max() / r.double().abs().max().clamp_min(1e-30))
for name, H, Ci, Co, k, s, pad in L:
    B4, R = 4 * b, 3 * b
    g4 = K.conv_desc(B4, H, H, Ci, Co, k, k, s, s, pad); gm = K.rebatch(g4, R); gh = K.rebatch(g4, B4 - R)
    d = g4[0]
    x = torch.randn(B4, H, H, Ci, generator=g, device='cuda'); w = torch.randn(k, k, Ci, Co, generator=g, device='cuda') * 0.05
    bias = torch.randn(Co, generator=g, device='cuda')
    dy = torch.randn(B4, d.Ho, d.Wo, Co, generator=g, device='cuda')
    y4 = K.conv_fwd(x, w, bias, g4[0], g4[1], K.ACT_LRELU, 0.2)
    yp = torch.cat([K.conv_fwd(x[:R].contiguous(), w, bias, gm[0], gm[1], K.ACT_LRELU, 0.2), K.conv_fwd(x[R:].contiguous(), w, bias, gh[0], gh[1], K.ACT_LRELU, 0.2)])
    d4 = K.conv_bwd_data(dy, w, None, g4[0], g4[1])
    dp = torch.cat([K.conv_bwd_data(dy[:R].contiguous(), w, None, gm[0], gm[1]), K.conv_bwd_data(dy[R:].contiguous(), w, None, gh[0], gh[1])])
    f4 = K.conv_bwd_filter(x, dy, g4[0], g4[1])
    fp = K.conv_bwd_filter(x[:R].contiguous(), dy[:R].contiguous(), gm[0], gm[1]) + K.conv_bwd_filter(x[R:].contiguous(), dy[R:].contiguous(), gh[0], gh[1])
    # views instead of copies (what the stacked step hands the kernels)
    yv = K.conv_fwd(x[R:], w, bias, gh[0], gh[1], K.ACT_LRELU, 0.2)
    dv = K.conv_bwd_data(dy[R:], w, None, gh[0], gh[1])
    print('%-7s fwd %.1e  bwdD %.1e  bwdF %.1e | hat views: fwd %.1e bwdD %.1e | algo %s/%s/%s  parts %s/%s' % (
        name, rel(y4, yp), rel(d4, dp), rel(f4, fp), rel(yv, yp[R:]), rel(dv, dp[R:]), K.conv_algo(g4[0], 'fwd'), K.conv_algo(g4[0], 'bwd_data'), K.conv_algo(g4[0], 'bwd_filter'),
        K.conv_algo(gm[0], 'fwd'), K.conv_algo(gh[0], 'fwd')))
