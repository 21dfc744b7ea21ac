"""Which conv calls of one bf16-storage wgancls iteration stage a cast (T2I_DEBUG_PLAN=1 prints them)?"""
import os, sys
os.environ['T2I_DEBUG_PLAN'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench, t2i_amd
from t2i_amd import kernels as K
from t2i_amd.models.wgancls.model import WGanCls
from t2i_amd.models.wgancls.trainer import WGanClsTrainer
K.filter_cache(True); K.set_math('bf16'); K.set_storage('bf16')
dev = torch.device('cuda'); cfg = bench.make_cfg(64)
m = WGanCls(cfg, device=dev, seed=0); tr = WGanClsTrainer(None, m, None, cfg)
feed = bench.synthetic_feed(cfg, dev, seed=1, with_noise=False)
tr.iteration(1, feed); torch.cuda.synchronize()
sys.stderr.write('==== iteration 2\n'); sys.stderr.flush()
tr.iteration(2, feed); torch.cuda.synchronize()
