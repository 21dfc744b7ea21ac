"""Entry point of the wgancls model — reference models/wgancls/run.py:13-74.

    python -m t2i_amd.models.wgancls.run --cfg <yaml> [--train] [--synthetic] [--steps N] [--batch B] [--graphs 0|1]

Behaviour of the reference's main(): read the config, create CHECKPOINT_DIR / SAMPLE_DIR / LOGS_DIR, load the pickled
dataset from cfg.DATASET_DIR (`TextDataset(datadir, 64)`, splits `<datadir>/test` and `<datadir>/train`), then switch on
the mode flags: EVAL.FLAG -> Inception-score evaluation, TRAIN.FLAG -> `WGanClsTrainer(...).train()` with its periodic
side effects (captions, sample grids, checkpoints, resume), neither -> the caption visualiser.  Evaluation and
visualisation are outside this build's scope and say so instead of silently doing something else.
Additions that the reference does not have, all explicit: `--train` forces the training mode whatever the yml says (the
shipped yml has TRAIN.FLAG: False); `--synthetic` replaces the pickled dataset by the on-device synthetic one
(t2i_amd.data) for machines without the data; `--steps` / `--batch` override TRAIN.MAX_STEPS / TRAIN.BATCH_SIZE."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))

import t2i_amd  # noqa: E402,F401
from t2i_amd.models.wgancls.model import WGanCls  # noqa: E402
from t2i_amd.models.wgancls.trainer import WGanClsTrainer  # noqa: E402
from t2i_amd.utils.config import config_from_yaml  # noqa: E402


def load_dataset(cfg, device, synthetic=False):
    """reference run.py:33-40.  The pickles are read once; images and caption embeddings then live in HBM."""
    if synthetic:
        from t2i_amd.data import SyntheticTextDataset
        return SyntheticTextDataset(cfg, device)
    from t2i_amd.preprocess.dataset import TextDataset
    datadir = cfg.DATASET_DIR
    if not os.path.isdir(datadir):
        raise FileNotFoundError('DATASET_DIR %r does not exist (expected <dir>/train and <dir>/test with the pickles of '
                                'preprocess/dataset.py); pass --synthetic to train on synthetic inputs instead' % datadir)
    dataset = TextDataset(datadir, cfg.MODEL.OUTPUT_SIZE, device=device)
    dataset.test = dataset.get_data('%s/test' % datadir)
    dataset.train = dataset.get_data('%s/train' % datadir)
    return dataset


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cfg', 'flowers.yml'),
                    help='Relative path to the config of the model')
    ap.add_argument('--train', action='store_true', help='train even if the yml says TRAIN.FLAG: False')
    ap.add_argument('--synthetic', action='store_true', help='synthetic on-device dataset instead of cfg.DATASET_DIR')
    ap.add_argument('--steps', type=int, default=None, help='override TRAIN.MAX_STEPS')
    ap.add_argument('--batch', type=int, default=None, help='override TRAIN.BATCH_SIZE')
    ap.add_argument('--graphs', type=int, default=1, help='1: replay the iteration from hipGraphs once it has run eagerly (default)')
    args = ap.parse_args(argv)
    print(args.cfg)
    cfg = config_from_yaml(args.cfg)
    if args.batch:
        cfg.TRAIN.BATCH_SIZE = args.batch
    if args.steps:
        cfg.TRAIN.MAX_STEPS = args.steps
    for d in (cfg.CHECKPOINT_DIR, cfg.SAMPLE_DIR, cfg.LOGS_DIR):
        if not os.path.exists(d):
            os.makedirs(d)

    if cfg.EVAL.FLAG:
        raise NotImplementedError('EVAL.FLAG: Inception-score / FID evaluation (reference models/wgancls/eval_wgan.py) is '
                                  'outside the hot path this build covers; see DESIGN.md §7')
    if not (cfg.TRAIN.FLAG or args.train):
        raise NotImplementedError('TRAIN.FLAG is False: the reference would start its caption visualiser (visualize_wgan.py), '
                                  'which is outside the hot path this build covers; pass --train or set TRAIN.FLAG: True')

    from t2i_amd import kernels as K
    K.filter_cache(os.environ.get('T2I_FILTER_CACHE', '1') != '0')     # weights change only through Adam / Saver here
    wgan = WGanCls(cfg)
    dataset = load_dataset(cfg, wgan.device, synthetic=args.synthetic)
    trainer = WGanClsTrainer(sess=None, model=wgan, dataset=dataset, cfg=cfg)
    return trainer.train(side_effects=True, graphs=bool(args.graphs))


if __name__ == '__main__':
    main()
