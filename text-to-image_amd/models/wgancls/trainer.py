"""WGanClsTrainer — the reference's training harness (reference models/wgancls/trainer.py:11-126).

Same constructor arity (`sess` is accepted and ignored: there is no session in an eager runtime) and the same
iteration order: batch -> z ~ N(0,1) -> eps ~ U(0,1) -> lr decay 0.95^((idx//n_critic)//10000) -> critic step (+kt) ->
generator step when idx % n_critic == 0 -> periodic side effects.  All per-iteration inputs are generated on the device
(the reference builds them in NumPy and pays a host->device feed every sess.run, trainer.py:77-95)."""
import sys
import time

import torch


class WGanClsTrainer(object):
    def __init__(self, sess, model, dataset, cfg):
        self.sess = sess            # unused; kept for call-site compatibility (reference run.py:55-60)
        self.model = model
        self.dataset = dataset
        self.cfg = cfg
        self.lr_d = self.cfg.TRAIN.D_LR
        self.lr_g = self.cfg.TRAIN.G_LR
        self.gen = torch.Generator(device=model.device).manual_seed(1234)
        self.last = {}

    def lr_scale(self, idx):
        """reference trainer.py:82-86"""
        n_critic = self.cfg.TRAIN.N_CRITIC
        return 0.95 ** ((idx // n_critic) // 10000)

    def make_feed(self, idx):
        """The feed_dict of reference trainer.py:84-95 (keys = placeholder names of model.py:36-46)."""
        m = self.model
        images, wrong_images, embed, _, _ = self.dataset.train.next_batch(m.batch_size, 4, embeddings=True, wrong_img=True)
        dev = m.device
        scale = self.lr_scale(idx)
        return {
            'learning_rate_d': self.lr_d * scale,
            'learning_rate_g': self.lr_g * scale,
            'x': images,
            'x_mismatch': wrong_images,
            'cond': embed,
            'z': torch.randn((m.batch_size, m.z_dim), generator=self.gen, device=dev),
            'epsilon': torch.rand((m.batch_size, 1, 1, 1), generator=self.gen, device=dev),
            'iter': idx,
        }

    def iteration(self, idx, feed=None):
        """One "G+D step" (reference trainer.py:97-102)."""
        feed = feed if feed is not None else self.make_feed(idx)
        out = {'d': self.model.d_step(feed)}
        if idx % self.cfg.TRAIN.N_CRITIC == 0:
            out['g'] = self.model.g_step(feed)
        self.last = out
        return out

    def train(self, max_steps=None, start_point=0, log=None):
        """reference trainer.py:49-126 without the TF summaries / PNG grids / checkpoints (DESIGN.md "next" rows): the
        scalars they would log are returned by every iteration instead."""
        log = log or (lambda s: (sys.stdout.write(s + '\n'), sys.stdout.flush()))
        end = max_steps if max_steps is not None else self.cfg.TRAIN.MAX_STEPS
        t0 = time.time()
        for idx in range(start_point + 1, end):
            out = self.iteration(idx)
            if idx % self.cfg.TRAIN.SUMMARY_PERIOD == 0:
                d, g = out['d'], out.get('g', {})
                log('[%6d] D_loss %.4f G_loss %.4f wdist %.4f wdist2 %.4f gp %.4f gp2 %.4f kt %.4f (%.1fs)' % (
                    idx, float(d['D_loss']), float(g.get('G_loss', float('nan'))), float(d['wdist']), float(d['wdist2']),
                    float(d['real_gp']), float(d['real_gp2']), float(self.model.kt), time.time() - t0))
        return self.last
