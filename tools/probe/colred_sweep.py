"""act_bwd_colsum / col_reduce / bn statistics on the shapes of the iteration, against the chunk planner's two tunables."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, t2i_amd
from t2i_amd import kernels as K
from tools.bench_conv import timeit
K.workspace(torch.device('cuda', 0), 1 << 28)
shapes = [(192, 32, 32, 128), (192, 16, 16, 256), (192, 8, 8, 512), (192, 4, 4, 1024), (64, 32, 32, 128), (64, 16, 16, 256), (64, 8, 8, 512)]
for dt in (torch.bfloat16, torch.float32):
    K.set_storage('f32'); K.set_math('bf16' if dt == torch.bfloat16 else 'f32'); K.set_storage('bf16' if dt == torch.bfloat16 else 'f32')
    for cap, wgs in ((192, 768), (384, 1536), (768, 3072), (1536, 6144), (3072, 12288)):
        K.tuning_set('colred_cap', cap); K.tuning_set('colred_wgs', wgs)
        row = []
        for shp in shapes:
            dy = torch.randn(shp, device='cuda').to(dt); y = torch.randn(shp, device='cuda').to(dt)
            t1 = timeit(lambda: K.act_bwd_colsum(dy, y, K.ACT_LRELU, 0.2), 10) * 1e6
            t2 = timeit(lambda: K.col_reduce(dy, dy, want_second=True), 10) * 1e6
            row.append('%5.1f/%5.1f' % (t1, t2))
        print('%-8s cap %4d wgs %5d | act_bwd_colsum / col_reduce(sum, sumsq) us: %s' % (str(dt).split('.')[-1], cap, wgs, '  '.join(row)))
K.tuning_set('colred_cap', 192); K.tuning_set('colred_wgs', 768); K.set_storage('f32'); K.set_math('f32')
print('shapes:', shapes)
