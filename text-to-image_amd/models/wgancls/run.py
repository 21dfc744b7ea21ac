"""Entry point mirroring reference models/wgancls/run.py:13-74: `--cfg <yaml>` then train when cfg.TRAIN.FLAG.
Evaluation / visualisation modes (Inception score, caption grids) are outside the hot path."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))

import t2i_amd  # noqa: E402,F401
from t2i_amd.data import SyntheticTextDataset  # noqa: E402
from t2i_amd.models.wgancls.model import WGanCls  # noqa: E402
from t2i_amd.models.wgancls.trainer import WGanClsTrainer  # noqa: E402
from t2i_amd.utils.config import config_from_yaml  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cfg', 'flowers.yml'),
                    help='Relative path to the config of the model')
    ap.add_argument('--steps', type=int, default=None, help='override TRAIN.MAX_STEPS')
    ap.add_argument('--batch', type=int, default=None, help='override TRAIN.BATCH_SIZE')
    ap.add_argument('--graphs', type=int, default=1, help='1: replay the iteration from hipGraphs once it has run eagerly (default)')
    args = ap.parse_args(argv)
    print(args.cfg)
    cfg = config_from_yaml(args.cfg)
    if args.batch:
        cfg.TRAIN.BATCH_SIZE = args.batch
    for d in (cfg.CHECKPOINT_DIR, cfg.SAMPLE_DIR, cfg.LOGS_DIR):
        os.makedirs(d, exist_ok=True)
    if cfg.EVAL.FLAG:
        raise NotImplementedError('EVAL mode (Inception score / FID) is outside the hot path; see DESIGN.md')
    from t2i_amd import kernels as K
    K.filter_cache(os.environ.get('T2I_FILTER_CACHE', '1') != '0')     # weights change only through Adam / Saver here
    wgan = WGanCls(cfg)
    dataset = SyntheticTextDataset(cfg, wgan.device)
    trainer = WGanClsTrainer(sess=None, model=wgan, dataset=dataset, cfg=cfg)
    trainer.train(max_steps=args.steps, graphs=bool(args.graphs))


if __name__ == '__main__':
    main()
