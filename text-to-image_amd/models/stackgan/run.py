"""Entry point for the two StackGAN stages (reference models/stackgan/stageI/run.py:25-79 and stageII/run.py:26-88):
`--stage 1 --cfg <stageI yaml>` or `--stage 2 --cfg_stage_I <yaml> --cfg <stageII yaml>`, then train on the synthetic
stand-in dataset.  Evaluation / visualisation modes are outside the hot path."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))

import torch  # noqa: E402
import t2i_amd  # noqa: E402,F401
from t2i_amd.data import SyntheticTextDataset  # noqa: E402
from t2i_amd.models.stackgan.stageI.model import ConditionalGan as StageI  # noqa: E402
from t2i_amd.models.stackgan.stageI.trainer import ConditionalGanTrainer as StageITrainer  # noqa: E402
from t2i_amd.models.stackgan.stageII.model import ConditionalGan as StageII  # noqa: E402
from t2i_amd.models.stackgan.stageII.trainer import ConditionalGanTrainer as StageIITrainer  # noqa: E402
from t2i_amd.utils.config import config_from_yaml  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def build(stage, cfg, cfg_stage_i=None, device=None):
    if stage == 1:
        model = StageI(cfg, device=device)
        return model, StageITrainer(None, model, SyntheticTextDataset(cfg, model.device), cfg)
    model = StageII(StageI(cfg_stage_i, build_model=False, device=device), cfg)
    return model, StageIITrainer(None, model, SyntheticTextDataset(cfg, model.device), cfg, cfg_stage_i)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--stage', type=int, choices=[1, 2], default=1)
    ap.add_argument('--cfg', default=None, help='config of the stage being trained')
    ap.add_argument('--cfg_stage_I', default=os.path.join(HERE, 'stageI', 'cfg', 'flowers.yml'))
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=None, help='override TRAIN.BATCH_SIZE')
    ap.add_argument('--math', choices=['f32', 'bf16'], default='f32')
    ap.add_argument('--graphs', type=int, default=1, help='1: replay the two halves of the iteration as hipGraphs (default)')
    args = ap.parse_args(argv)
    from t2i_amd import kernels as K
    K.set_math(args.math)
    cfg1 = config_from_yaml(args.cfg_stage_I)
    cfg = config_from_yaml(args.cfg or os.path.join(HERE, 'stageI' if args.stage == 1 else 'stageII', 'cfg', 'flowers.yml'))
    if args.batch:
        cfg.TRAIN.BATCH_SIZE = cfg1.TRAIN.BATCH_SIZE = args.batch
    model, trainer = build(args.stage, cfg, cfg1)
    feed = trainer.make_feed()
    for i in range(3):
        if args.graphs and i == 2:
            trainer.enable_graphs(feed)
        trainer.iteration(feed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = trainer.iteration(feed)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print('stackgan stage %d  batch %d  %s %s  %.2f ms/iteration  %.1f images/s  d_loss %.4f g_loss %.4f' % (
        args.stage, model.batch_size, args.math, 'graphs' if args.graphs else 'eager', dt * 1e3, model.batch_size / dt, float(out['d']['D_loss']), float(out['g']['G_loss'])))


if __name__ == '__main__':
    main()
