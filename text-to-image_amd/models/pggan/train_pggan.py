"""Stage driver of the progressive-growing schedule — reference models/pggan/train_pggan.py:17-69: stages
1, 2t, 2, 3t, 3, ... (t = fade-in transition), batch 16 (8 from stage 6 on), 600000 images per stage, checkpoints
written under `<CHECKPOINT_DIR>/stage<k>/` and read from the previous stage's directory.  Synthetic data stands in for
the pickled datasets; `--iters` bounds the iterations per stage (the full 600000 // batch when omitted)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))

import torch  # noqa: E402
import t2i_amd  # noqa: E402,F401
from t2i_amd import kernels as K  # noqa: E402
from t2i_amd.data import SyntheticTextDataset  # noqa: E402
from t2i_amd.models.pggan.pggan import PGGAN  # noqa: E402
from t2i_amd.utils.config import AttrDict  # noqa: E402

STAGE = [1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8]            # train_pggan.py:19-20
PREV_STAGE = [1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8]


def dataset_for(size, device):
    cfg = AttrDict({'MODEL': {'IMAGE_SHAPE': {'H': size, 'W': size, 'D': 3}, 'EMBED_DIM': 1024}})
    return SyntheticTextDataset(cfg, device)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='./pggan_run/')
    ap.add_argument('--iters', type=int, default=None, help='iterations per stage (default: 600000 // batch as the reference)')
    ap.add_argument('--first', type=int, default=0, help='index into the 15-entry schedule to start from')
    ap.add_argument('--last', type=int, default=len(STAGE) - 1)
    ap.add_argument('--math', choices=['f32', 'bf16'], default='f32')
    ap.add_argument('--eager', action='store_true', help='--bench: keep eager launches instead of hipGraph replay')
    ap.add_argument('--bench', action='store_true', help='time `--iters` iterations of each entry instead of training with side effects')
    args = ap.parse_args(argv)
    K.set_math(args.math)
    dev = torch.device('cuda')
    for i in range(args.first, args.last + 1):
        t = (i % 2 == 1)
        batch_size = 8 if STAGE[i] >= 6 else 16
        max_iters = 600000 // batch_size
        wdir = os.path.join(args.out, 'checkpoints', 'stage%d/' % STAGE[i])
        rdir = os.path.join(args.out, 'checkpoints', 'stage%d/' % PREV_STAGE[i])
        sample_path = os.path.join(args.out, 'samples', ('stage_t%d/' if t else 'stage%d/') % STAGE[i])
        for d in (wdir, rdir, sample_path):
            os.makedirs(d, exist_ok=True)
        size = 4 * 2 ** (STAGE[i] - 1)
        pggan = PGGAN(batch_size=batch_size, steps=max_iters, check_dir_write=wdir, check_dir_read=rdir,
                      dataset=dataset_for(size, dev), sample_path=sample_path, log_dir=None, stage=STAGE[i], trans=t, device=dev)
        if args.bench:
            gen = torch.Generator(device=dev).manual_seed(0)
            feed = pggan.make_feed(gen)
            for k in range(3):
                if k == 2 and not args.eager:
                    pggan.enable_graphs(feed)
                pggan.iteration(1 + k, feed)
            torch.cuda.synchronize()
            n = args.iters or 10
            t0 = time.perf_counter()
            for k in range(n):
                pggan.iteration(4 + k, feed)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print('pggan stage %d%s  %3dx%-3d batch %2d  %s  %.2f ms/iteration  %.1f images/s' % (
                STAGE[i], 't' if t else ' ', size, size, batch_size, args.math + (' eager' if args.eager else ' graphs'), dt * 1e3, batch_size / dt))
        else:
            pggan.train(max_steps=args.iters, side_effects=True)
        del pggan
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
