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
        if idx % self.cfg.TRAIN.N_CRITIC == 0:
            d, g = self.model.dg_step(feed)              # D then G on the same feed; one graph launch under replay
            out = {'d': d, 'g': g}
        else:
            out = {'d': self.model.d_step(feed)}
        self.last = out
        return out

    def make_saver(self):
        """tf.train.Saver(max_to_keep=CHECKPOINTS_TO_KEEP) (reference trainer.py:51): every variable under its TF name, the
        two optimizers' Adam slots and step counts, the kt balance scalar and global_step (model.py:28)."""
        from ...utils.saver import Saver
        m = self.model

        def set_kt(v):
            m.kt.fill_(float(v))

        def set_step(v):
            m.global_step = int(v)
        return Saver(m.store, {'D_optim': m.D_optim, 'G_optim': m.G_optim},
                     {'kt': (lambda: m.kt.detach().cpu().numpy(), set_kt), 'global_step': (lambda: m.global_step, set_step)},
                     max_to_keep=int(getattr(self.cfg.TRAIN, 'CHECKPOINTS_TO_KEEP', 5)))

    def define_summaries(self):
        """reference trainer.py:20-47: a tf.summary.FileWriter on cfg.LOGS_DIR (utils/summary.py writes the same event-file
        format without TensorFlow)."""
        from ...utils.summary import FileWriter
        self.writer = FileWriter(self.cfg.LOGS_DIR)

    def write_summaries(self, idx, feed, out, sample_z=None):
        """The merged summary of reference trainer.py:21-45 at step idx: images `x` and `G_img` (first 3 of the batch, normalised as
        tf.summary.image does), histograms `z` / `z_sample`, and the fourteen scalars under the reference's tags.  The reference
        re-evaluates these tensors in a third sess.run AFTER the updates (with a fresh noise draw); here they are the values the
        iteration itself computed — the same quantities one update earlier — so that logging costs no extra forward pass."""
        from ...utils import summary as S
        d, g = out['d'], out.get('g')
        np_ = lambda t: t.detach().float().cpu().numpy()
        vals = [S.image('x', np_(feed['x'])), S.image('G_img', np_(d['G'])), S.histogram('z', np_(feed['z']))]
        if sample_z is not None:
            vals.append(S.histogram('z_sample', np_(sample_z)))
        if g is not None:
            vals += [S.scalar('G_loss_wass', -float(g['D_loss_fake'])), S.scalar('kl_loss', float(g['G_kl_loss'])),
                     S.scalar('G_loss', float(g['G_loss']))]
        for tag, key in (('D_loss_real', 'D_loss_real'), ('D_loss_fake', 'D_loss_fake'), ('real_gp', 'real_gp'), ('D_loss', 'D_loss'),
                         ('reg_loss', 'reg_loss'), ('wdist', 'wdist'), ('wdist2', 'wdist2'), ('d_loss_mismatch', 'D_loss_mismatch'),
                         ('real_gp2', 'real_gp2'), ('kt', 'kt'), ('balance_loss', 'balance_loss')):
            vals.append(S.scalar(tag, float(d[key])))
        self.writer.add_summary(vals, idx)
        self.writer.flush()

    def train(self, max_steps=None, start_point=None, log=None, side_effects=False, graphs=False):
        """reference trainer.py:49-126.  With side_effects=True the periodic work around the hot path is on as in the
        reference: resume from the latest checkpoint in cfg.CHECKPOINT_DIR, captions of the fixed sample batch, a PNG grid
        of `sampler` outputs every TRAIN.SAMPLE_PERIOD iterations, a checkpoint when idx % 500 == 2, and a TensorBoard event file
        in cfg.LOGS_DIR with the reference's images / histograms / scalars every SUMMARY_PERIOD iterations (write_summaries), next to
        the scalar log line; the scalars are also returned per iteration.
        graphs=True: once a critic step and a generator step have run eagerly, the iteration is captured into hipGraphs and
        replayed (bit-identical to the eager launches, including the order of the conditioning-noise draws)."""
        from ...utils.saver import load, save
        from ...utils.utils import get_balanced_factorization, save_captions, save_images
        log = log or (lambda s: (sys.stdout.write(s + '\n'), sys.stdout.flush()))
        end = max_steps if max_steps is not None else self.cfg.TRAIN.MAX_STEPS
        m = self.model
        if side_effects:
            self.saver = self.make_saver()
            sample_z = torch.randn((m.sample_num, m.z_dim), generator=self.gen, device=m.device)
            _, sample_cond, _, captions = self.dataset.test.next_batch_test(m.sample_num, 0, 1)
            sample_cond = sample_cond[0]
            save_captions(self.cfg.SAMPLE_DIR, captions)
            if getattr(self.cfg, 'LOGS_DIR', None):
                self.define_summaries()
            could_load, counter = load(self.saver, None, self.cfg.CHECKPOINT_DIR)
            if start_point is None:
                start_point = counter if could_load else 0
            log(' [*] Load SUCCESS' if could_load else ' [!] Load failed...')
        start_point = start_point or 0
        t0 = time.time()
        seen_g = False
        for idx in range(start_point + 1, end):
            feed = self.make_feed(idx)
            out = self.iteration(idx, feed)
            seen_g = seen_g or 'g' in out
            if graphs and seen_g and m._graphs is None:
                m.enable_graphs(feed)
            if idx % self.cfg.TRAIN.SUMMARY_PERIOD == 0:
                d, g = out['d'], out.get('g', {})
                log('[%6d] D_loss %.4f G_loss %.4f wdist %.4f wdist2 %.4f gp %.4f gp2 %.4f kt %.4f (%.1fs)' % (
                    idx, float(d['D_loss']), float(g.get('G_loss', float('nan'))), float(d['wdist']), float(d['wdist2']),
                    float(d['real_gp']), float(d['real_gp2']), float(self.model.kt), time.time() - t0))
            if side_effects and getattr(self, 'writer', None) is not None and idx % self.cfg.TRAIN.SUMMARY_PERIOD == 0:
                self.write_summaries(idx, feed, out, sample_z)
            if side_effects:
                epoch = idx // max(self.dataset.train.num_examples // m.batch_size, 1)
                if idx % self.cfg.TRAIN.SAMPLE_PERIOD == 0:
                    samples = m.sampler(sample_z, sample_cond)
                    save_images(samples, get_balanced_factorization(samples.shape[0]),
                                '{}train_{:02d}_{:04d}.png'.format(self.cfg.SAMPLE_DIR, epoch, idx))
                if idx % 500 == 2:
                    save(self.saver, None, self.cfg.CHECKPOINT_DIR, idx)
        return self.last
