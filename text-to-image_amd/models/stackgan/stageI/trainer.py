"""ConditionalGanTrainer (Stage-I) — reference models/stackgan/stageI/trainer.py:11-165: sigmoid cross-entropy losses
(real label 0.9), KL term of the conditioning augmentation, two Adam optimizers on ONE learning-rate placeholder
(D_LR * 0.5 ** (epoch // 100)), both under tf.GraphKeys.UPDATE_OPS, D update then G update every iteration."""
import os
import sys
import time

import torch

from .... import autograd as A
from .... import stacked as ST
from .... import kernels as K
from .... import optim
from ....utils.ops import update_ops

# the generator step's three critic evaluations as one stacked pass (stacked.py); T2I_CGAN_STACK_G=0: fake pass + [match | mismatch] pass
_STACK_G = os.environ.get('T2I_CGAN_STACK_G', '1') != '0'


class ConditionalGanTrainer(object):
    REAL_LABEL = 0.9                     # trainer.py:26 (Stage-II overrides with 0.95)

    def __init__(self, sess, model, dataset, cfg):
        self.sess, self.model, self.dataset, self.cfg = sess, model, dataset, cfg     # sess unused (no TF session)
        self.lr = float(cfg.TRAIN.D_LR)
        self.gen = torch.Generator(device=model.device).manual_seed(1234)
        self.batched = os.environ.get('T2I_GANCLS_BATCHED', '1') != '0'      # the critic's passes of one sess.run as one stacked batch (models/gancls)
        self.define_losses()

    def define_losses(self):
        t = self.cfg.TRAIN
        self.alpha, self.kl_coeff = float(t.COEFF.ALPHA_MISMATCH_LOSS), float(t.COEFF.KL)
        self.D_optim = optim.AdamTF(self.model.d_arena, float(t.D_BETA_DECAY), 0.999)
        self.G_optim = optim.AdamTF(self.model.g_arena, float(t.G_BETA_DECAY), 0.999)

    @staticmethod
    def kl_loss(mean, log_sigma):
        return (-log_sigma + 0.5 * (-1.0 + torch.exp(2.0 * log_sigma) + mean * mean)).mean()

    # what the model's generator consumes differs per stage (Stage-II feeds the Stage-I image): one hook
    def _generate(self, feed, which):
        m = self.model
        return m.generator(feed['z'], feed['phi_inputs'], reuse=True, noise=feed.get('ca_noise_' + which))

    def d_losses(self, feed):
        """What sess.run([D_optim, ...]) evaluates before the update (trainer.py:19-41).  Gradients -> d_arena."""
        m = self.model
        x, xw, phi = feed['inputs'], feed['wrong_inputs'], feed['phi_inputs']
        with update_ops():      # D_optim sits under control_dependencies(UPDATE_OPS): every BN moving average moves
            with torch.no_grad():
                G, _, _ = self._generate(feed, 'd')
            if self.batched:          # fake | match | mismatch stacked along the batch axis, batch-norm statistics per pass
                _, logits = m.discriminator(torch.cat([G, x, xw], 0), torch.cat([phi, phi, phi], 0), reuse=True, _prob=False, groups=3)
                B = x.shape[0]
                lv = logits.detach().reshape(3, B)
                heads, outs = [lv[0], lv[1], lv[2]], [logits]
                seed = torch.empty(3 * B, dtype=torch.float32, device=logits.device)
            else:
                _, l_fake = m.discriminator(G, phi, reuse=True, _prob=False)
                _, l_match = m.discriminator(x, phi, reuse=True, _prob=False)
                _, l_mis = m.discriminator(xw, phi, reuse=True, _prob=False)
                heads, outs, seed = [l_fake.detach().reshape(-1), l_match.detach().reshape(-1), l_mis.detach().reshape(-1)], [l_fake, l_match, l_mis], None
        # the three heads (trainer.py:24-33) in one launch: loss scalars + d D_loss / d logits as the seeds of the backward pass
        losses, seeds, _ = K.sigmoid_ce_head(heads, [0.0, self.REAL_LABEL, 0.0], [1.0 - self.alpha, 1.0, self.alpha], want_prob=False, seeds_into=seed)
        m.d_arena.zero_grad()
        if m.dp is not None and not getattr(self, '_capturing', False):
            m.dp.arm(m.d_arena)
        grads = [seed.view_as(outs[0])] if seed is not None else [s_.view_as(l_) for s_, l_ in zip(seeds, outs)]
        torch.autograd.backward(outs, grads, inputs=list(m.d_vars.values()))
        A.side_join()
        # the three critic outputs (fake, match, mismatch logits) ride along for the D summary's histograms (trainer.py:59-61)
        return dict(D_loss=losses[0], D_real_match_loss=losses[2], D_real_mismatch_loss=losses[3], D_synthetic_loss=losses[1], G=G,
                    D_logits=heads)

    def g_losses(self, feed):
        m = self.model
        x, xw, phi = feed['inputs'], feed['wrong_inputs'], feed['phi_inputs']
        with update_ops():
            G, mean, log_sigma = self._generate(feed, 'g')
            # G_optim also sits under ALL update ops of the graph: the match / mismatch critic passes run in this
            # sess.run too, only to move their batch-norm moving averages (trainer.py:50-55)
            if self.batched and _STACK_G and x.is_cuda:
                # round 6: fake | match | mismatch as ONE stacked pass of three evaluations (stacked.py): every conv once on 3B rows, per-evaluation
                # batch-norm statistics, the moving averages move once per evaluation in this order (as the three calls did); only the fake rows
                # carry a gradient, so the backward runs on B rows as before
                with m.store.frozen(m.d_scope):
                    _, ls = m.discriminator(ST.Stacked(G, torch.cat([x, xw], 0)), ST.Stacked(phi, torch.cat([phi, phi], 0)), reuse=True, _prob=False,
                                            groups=3)
                l_fake = ls.main
            else:
                with m.store.frozen(m.d_scope):
                    _, l_fake = m.discriminator(G, phi, reuse=True, _prob=False)
                with torch.no_grad():
                    if self.batched:
                        m.discriminator(torch.cat([x, xw], 0), torch.cat([phi, phi], 0), reuse=True, _prob=False, groups=2)
                    else:
                        m.discriminator(x, phi, reuse=True, _prob=False)
                        m.discriminator(xw, phi, reuse=True, _prob=False)
        # G_loss = CE(fake, 1) + kl_coeff * KL (trainer.py:35-41): the CE head gives its value and d/d logits; the KL term stays a
        # differentiable tensor expression, so the two are seeded together
        losses, seeds, _ = K.sigmoid_ce_head([l_fake.detach().reshape(-1)], [1.0], [1.0], want_prob=False)
        G_gan_loss = losses[1]
        G_kl_loss = self.kl_loss(mean, log_sigma)
        m.g_arena.zero_grad()
        if m.dp is not None and not getattr(self, '_capturing', False):
            m.dp.arm(m.g_arena)
        torch.autograd.backward([l_fake, G_kl_loss], [seeds[0].view_as(l_fake), torch.full_like(G_kl_loss, self.kl_coeff)],
                                inputs=list(m.g_vars.values()))
        A.side_join()
        G_loss = G_gan_loss + self.kl_coeff * G_kl_loss.detach()
        return dict(G_loss=G_loss, G_gan_loss=G_gan_loss, G_kl_loss=G_kl_loss.detach(), G=G.detach())

    # ---- device-only halves of an iteration (graph-capturable: Adam reads its step size from device memory) -----------------
    def _d_body(self, feed):
        m = self.model
        d = self.d_losses(feed)
        scale = m.dp.allreduce_arena(m.d_arena) if m.dp is not None else 1.0
        self.D_optim.apply(grad_scale=scale)
        return d

    def _g_body(self, feed):
        m = self.model
        g = self.g_losses(feed)
        scale = m.dp.allreduce_arena(m.g_arena) if m.dp is not None else 1.0
        self.G_optim.apply(grad_scale=scale)
        return g

    NOISE_KEYS = ('ca_noise_d', 'ca_noise_g')

    def enable_graphs(self, feed):
        """Capture the two halves into hipGraphs and replay them from then on.  Call after at least one eager iteration with
        the same shapes.  The conditioning-augmentation noise lives in static buffers that are re-drawn before every replay
        (the reference draws it inside the graph on every run).  With data parallelism each half is cut at its exchange
        step — [losses + backward] | all-reduce of the gradient arena, issued eagerly | [Adam] — as in models/wgancls."""
        from ....graphs import StepGraphs
        m = self.model
        feed = dict(feed)
        for k in self.NOISE_KEYS:
            if feed.get(k) is None:
                feed[k] = torch.empty(feed['z'].shape[0], self._noise_dim(k), device=m.device)
        self._graphs = StepGraphs(feed, ('inputs', 'wrong_inputs', 'phi_inputs', 'z') + tuple(self.NOISE_KEYS),
                                  filters=(m.d_arena.flat, m.g_arena.flat))
        self._draw_noise(feed)
        self._graphs.load(feed)
        if m.dp is None:
            self._graphs.capture('d', self._d_body)
            self._graphs.capture('g', self._g_body)
            return
        scale = 1.0 / m.dp.world
        self._capturing = True
        try:
            self._graphs.capture('d', self.d_losses, capture_error_mode='thread_local')
            self._graphs.capture('d_upd', lambda f: self.D_optim.apply(grad_scale=scale), capture_error_mode='thread_local', refresh=False)
            self._graphs.capture('g', self.g_losses, capture_error_mode='thread_local')
            self._graphs.capture('g_upd', lambda f: self.G_optim.apply(grad_scale=scale), capture_error_mode='thread_local', refresh=False)
        finally:
            self._capturing = False

    def _noise_dim(self, key):
        return self.model.compressed_embed_dim

    def _draw_noise(self, feed):
        for k in self.NOISE_KEYS:
            if feed.get(k) is None or feed[k] is self._graphs.static.get(k):
                K.trunc_normal_(self._graphs.static[k])

    def iteration(self, feed, epoch=0):
        lr = self.lr * (0.5 ** (epoch // 100))                       # trainer.py:119,127
        graphs = getattr(self, '_graphs', None)
        if graphs is not None:
            self._draw_noise(feed)
            graphs.load(feed)
            dp = self.model.dp
            self.D_optim.prepare(lr)
            d = graphs.replay('d')
            if dp is not None:
                dp.allreduce_arena(self.model.d_arena)
                graphs.replay('d_upd')
            self.G_optim.prepare(lr)
            g = graphs.replay('g')
            if dp is not None:
                dp.allreduce_arena(self.model.g_arena)
                graphs.replay('g_upd')
            return {'d': d, 'g': g}
        self.D_optim.prepare(lr)
        d = self._d_body(feed)
        self.G_optim.prepare(lr)
        g = self._g_body(feed)
        return {'d': d, 'g': g}

    def make_feed(self):
        m = self.model
        images, wrong_images, embed, _, _ = self.dataset.train.next_batch(m.batch_size, 4, embeddings=True, wrong_img=True)
        return {'inputs': images, 'wrong_inputs': wrong_images, 'phi_inputs': embed,
                'z': torch.randn((m.batch_size, m.z_dim), generator=self.gen, device=m.device)}

    def define_summaries(self):
        """reference trainer.py:58-87: the FileWriter on cfg.LOGS_DIR (utils/summary.py: TensorBoard event files without TensorFlow)."""
        from ....utils.summary import FileWriter
        self.writer = FileWriter(self.cfg.LOGS_DIR)

    def write_summaries(self, counter, out):
        """The two merged summaries the reference adds per update (trainer.py:72-85,134-141).  D: histograms of the three critic
        outputs (the sigmoid of the logits: model.D_synthetic / D_real_match / D_real_mismatch), scalars of the three loss terms and
        d_loss; G: image g_sum, scalars g_loss / g_gan_loss / g_kl_loss (z_sum is defined by the reference but merged into neither) —
        from the values the iteration computed, as the reference fetches them in the same sess.run as the optimizer steps."""
        from ....utils import summary as S
        np_ = lambda t: t.detach().float().cpu().numpy()
        d, g = out['d'], out['g']
        prob = [np_(torch.sigmoid(l_)) for l_ in d['D_logits']]           # fake, match, mismatch
        self.writer.add_summary([S.histogram('d_real_mismatch_sum', prob[2]), S.histogram('d_real_match_sum', prob[1]),
                                 S.histogram('d_synthetic_sum', prob[0]), S.scalar('d_synthetic_sum_loss', float(d['D_synthetic_loss'])),
                                 S.scalar('d_real_mismatch_sum_loss', float(d['D_real_mismatch_loss'])),
                                 S.scalar('d_real_match_sum_loss', float(d['D_real_match_loss'])), S.scalar('d_loss', float(d['D_loss']))], counter)
        self.writer.add_summary([S.image('g_sum', np_(g['G'])), S.scalar('g_loss', float(g['G_loss'])),
                                 S.scalar('g_gan_loss', float(g['G_gan_loss'])), S.scalar('g_kl_loss', float(g['G_kl_loss']))], counter)
        self.writer.flush()

    def train(self, max_updates=None, log=None, summaries=False):
        """summaries=True: an event file in cfg.LOGS_DIR with the reference's per-update summaries (write_summaries)."""
        log = log or (lambda s: (sys.stdout.write(s + '\n'), sys.stdout.flush()))
        if summaries and getattr(self.cfg, 'LOGS_DIR', None):
            self.define_summaries()
        t0, counter = time.time(), 1
        for epoch in range(self.cfg.TRAIN.EPOCH):
            updates_per_epoch = self.dataset.train.num_examples // self.model.batch_size
            for idx in range(updates_per_epoch):
                out = self.iteration(self.make_feed(), epoch)
                if getattr(self, 'writer', None) is not None:
                    self.write_summaries(counter, out)
                log('Epoch: [%2d] [%4d/%4d] time: %4.4f, d_loss: %.8f, g_loss: %.8f' % (
                    epoch, idx, updates_per_epoch, time.time() - t0, float(out['d']['D_loss']), float(out['g']['G_loss'])))
                counter += 1
                if max_updates is not None and counter > max_updates:
                    return
