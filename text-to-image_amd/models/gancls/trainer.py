"""GanClsTrainer — reference models/gancls/trainer.py:12-164: losses, the two Adam optimizers (both under UPDATE_OPS) and
the D-then-G update order of every iteration."""
import os
import sys
import time

import torch

from ... import autograd as A
from ... import stacked as ST
from ... import kernels as K
from ... import optim
from ...utils.ops import update_ops

# the generator step's three critic evaluations as one stacked pass (stacked.py); T2I_CGAN_STACK_G=0: fake pass + [match | mismatch] pass
_STACK_G = os.environ.get('T2I_CGAN_STACK_G', '1') != '0'


class GanClsTrainer(object):
    def __init__(self, sess, model, dataset, cfg):
        self.sess, self.model, self.dataset, self.cfg = sess, model, dataset, cfg     # sess unused (no TF session)
        self.gen = torch.Generator(device=model.device).manual_seed(1234)
        self.batched = os.environ.get('T2I_GANCLS_BATCHED', '1') != '0'      # the critic's passes of one sess.run as one stacked batch
        # The D run and the G run of one iteration evaluate the generator on the SAME feed (trainer.py:115-134: same z, same phi), the generator
        # has no noise input (model.py:111-192) and D_optim does not touch its variables: the two evaluations are the same numbers.  iteration()
        # therefore evaluates it ONCE (with the autograd graph the G run needs), hands the critic step a detached view, and lets the batch norms'
        # moving averages take the batch statistics twice, as the two runs under UPDATE_OPS do (update_ops(times=2)).  Single process only: under
        # data parallelism the two halves live in separately captured graph segments.  T2I_GANCLS_SHARE_G=0: evaluate twice.
        self.share_g = os.environ.get('T2I_GANCLS_SHARE_G', '1') != '0'
        self._shared_G = None
        self.define_losses()

    def define_losses(self):
        t = self.cfg.TRAIN
        self.alpha = float(t.COEFF.ALPHA_MISMATCH_LOSS)
        self.D_optim = optim.AdamTF(self.model.d_arena, float(t.D_BETA_DECAY), 0.999)
        self.G_optim = optim.AdamTF(self.model.g_arena, float(t.G_BETA_DECAY), 0.999)

    def d_losses(self, feed, keep_g=False):
        """What sess.run([D_optim, ...]) evaluates before the update (trainer.py:20-34,115-123).  Gradients -> d_arena.
        keep_g (iteration() only): the generator is evaluated with its autograd graph and its moving averages move twice; the result is kept
        for g_losses(G=...) of the same iteration (see __init__)."""
        m = self.model
        x, xw, phi, z = feed['inputs'], feed['wrong_inputs'], feed['phi_inputs'], feed['z']
        if keep_g:
            with update_ops(times=2):
                self._shared_G = m.generator(z, phi, reuse=True)
            G = self._shared_G.detach()
        with update_ops():      # D_optim is built under control_dependencies(UPDATE_OPS): every BN moving average moves
            if not keep_g:
                with torch.no_grad():
                    G = m.generator(z, phi, reuse=True)
            # the critic's three passes (fake / match / mismatch, model.py:48-51) as ONE stacked batch: per-sample layers run once on 3B
            # samples, every batch norm keeps its statistics per pass (discriminator(groups=3)); T2I_GANCLS_BATCHED=0: three calls
            if self.batched:
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
        # the three heads of trainer.py:20-34 in ONE launch (labels 0 / 0.9 one-sided smoothing / 0; D_loss = match + alpha mismatch +
        # (1 - alpha) fake): the loss scalars, d D_loss / d logits as the seeds of the backward pass, and the sigmoid outputs
        losses, seeds, probs = K.sigmoid_ce_head(heads, [0.0, 0.9, 0.0], [1.0 - self.alpha, 1.0, self.alpha], seeds_into=seed)
        m.d_arena.zero_grad()
        if m.dp is not None and not getattr(self, '_capturing', False):
            m.dp.arm(m.d_arena)
        grads = [seed.view_as(outs[0])] if seed is not None else [s_.view_as(l_) for s_, l_ in zip(seeds, outs)]
        torch.autograd.backward(outs, grads, inputs=list(m.d_vars.values()))
        A.side_join()
        shape = (x.shape[0], 1, 1, 1)
        return dict(D_loss=losses[0], D_real_match_loss=losses[2], D_real_mismatch_loss=losses[3], D_synthetic_loss=losses[1], G=G,
                    D_synthetic=probs[0].view(shape), D_real_match=probs[1].view(shape), D_real_mismatch=probs[2].view(shape))

    def g_losses(self, feed, G=None):
        """G: the generator output d_losses(keep_g=True) evaluated for this iteration (with its autograd graph), or None: evaluate it here."""
        m = self.model
        x, xw, phi, z = feed['inputs'], feed['wrong_inputs'], feed['phi_inputs'], feed['z']
        with update_ops():
            if G is None:
                G = m.generator(z, phi, reuse=True)
            # G_optim also sits under ALL update ops of the graph: the match / mismatch critic passes run in this
            # sess.run too, only to move their batch-norm moving averages (trainer.py:46-51)
            if self.batched and _STACK_G and x.is_cuda:
                # round 6: fake | match | mismatch as ONE stacked pass of three evaluations (stacked.py): every conv once on 3B rows, per-evaluation
                # batch-norm statistics, the moving averages move once per evaluation in this order (as the three calls did); only the fake rows
                # carry a gradient, so the backward runs on B rows as before
                with m.store.frozen('d_net'):
                    _, ls = m.discriminator(ST.Stacked(G, torch.cat([x, xw], 0)), ST.Stacked(phi, torch.cat([phi, phi], 0)), reuse=True, _prob=False,
                                            groups=3)
                l_fake = ls.main
            else:
                with m.store.frozen('d_net'):
                    _, l_fake = m.discriminator(G, phi, reuse=True, _prob=False)
                with torch.no_grad():
                    if self.batched:
                        m.discriminator(torch.cat([x, xw], 0), torch.cat([phi, phi], 0), reuse=True, _prob=False, groups=2)
                    else:
                        m.discriminator(x, phi, reuse=True, _prob=False)
                        m.discriminator(xw, phi, reuse=True, _prob=False)
        losses, seeds, _ = K.sigmoid_ce_head([l_fake.detach().reshape(-1)], [1.0], [1.0], want_prob=False)      # G_loss: label 1 (trainer.py:36)
        m.g_arena.zero_grad()
        if m.dp is not None and not getattr(self, '_capturing', False):
            m.dp.arm(m.g_arena)
        torch.autograd.backward([l_fake], [seeds[0].view_as(l_fake)], inputs=list(m.g_vars.values()))
        A.side_join()
        return dict(G_loss=losses[0], G=G.detach())

    # ---- device-only halves of an iteration (graph-capturable) ------------------------------------------------------------
    def _d_body(self, feed, keep_g=False, refresh=None):
        m = self.model
        d = self.d_losses(feed, keep_g=keep_g)
        scale = m.dp.allreduce_arena(m.d_arena) if m.dp is not None else 1.0
        self.D_optim.apply(grad_scale=scale, refresh=refresh)
        return d

    def _g_body(self, feed, G=None):
        m = self.model
        g = self.g_losses(feed, G=G)
        scale = m.dp.allreduce_arena(m.g_arena) if m.dp is not None else 1.0
        self.G_optim.apply(grad_scale=scale)
        return g

    def _dg_body(self, feed):
        """Both halves with ONE generator evaluation (see __init__); the critic's cached filter images are regenerated behind its Adam step,
        because the generator half of the same capture reads the updated filters."""
        d = self._d_body(feed, keep_g=True, refresh=True)
        G, self._shared_G = self._shared_G, None
        return d, self._g_body(feed, G=G)

    def enable_graphs(self, feed):
        """Capture the iteration into hipGraphs and replay it from then on (call after one eager iteration): one graph for both halves when
        the generator evaluation is shared (single process), else one per half.  With data parallelism each half is cut at its exchange step —
        [losses + backward] | all-reduce of the gradient arena, issued eagerly | [Adam] — as in models/wgancls and models/stackgan: no
        collective is ever captured."""
        from ...graphs import StepGraphs
        m = self.model
        self._graphs = StepGraphs(feed, ('inputs', 'wrong_inputs', 'phi_inputs', 'z'), filters=(m.d_arena.flat, m.g_arena.flat))
        if m.dp is None:
            if self.share_g:
                self._graphs.capture('dg', self._dg_body)
            else:
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

    def iteration(self, feed):
        lr_d, lr_g = float(self.cfg.TRAIN.D_LR), float(self.cfg.TRAIN.G_LR)
        graphs = getattr(self, '_graphs', None)
        dp = self.model.dp
        if graphs is not None:
            graphs.load(feed)
            if 'dg' in graphs.graphs:
                self.D_optim.prepare(lr_d)
                self.G_optim.prepare(lr_g)
                d, g = graphs.replay('dg')
                return {'d': d, 'g': g}
            self.D_optim.prepare(lr_d)
            d = graphs.replay('d')
            if dp is not None:
                dp.allreduce_arena(self.model.d_arena)
                graphs.replay('d_upd')
            self.G_optim.prepare(lr_g)
            g = graphs.replay('g')
            if dp is not None:
                dp.allreduce_arena(self.model.g_arena)
                graphs.replay('g_upd')
            return {'d': d, 'g': g}
        if self.share_g and dp is None:
            self.D_optim.prepare(lr_d)
            self.G_optim.prepare(lr_g)
            d, g = self._dg_body(feed)
            return {'d': d, 'g': g}
        self.D_optim.prepare(lr_d)
        d = self._d_body(feed)
        self.G_optim.prepare(lr_g)
        return {'d': d, 'g': self._g_body(feed)}

    def make_feed(self):
        m = self.model
        images, wrong_images, embed, _, _ = self.dataset.train.next_batch(m.batch_size, 4, embeddings=True, wrong_img=True)
        return {'inputs': images, 'wrong_inputs': wrong_images, 'phi_inputs': embed,
                'z': torch.randn((m.batch_size, m.z_dim), generator=self.gen, device=m.device)}

    def define_summaries(self):
        """reference trainer.py:53-75: the FileWriter on cfg.LOGS_DIR (utils/summary.py: TensorBoard event files without TensorFlow)."""
        from ...utils.summary import FileWriter
        self.writer = FileWriter(self.cfg.LOGS_DIR)

    def write_summaries(self, counter, feed, out):
        """The two merged summaries the reference adds per update (trainer.py:66-72,115-134) — D: histograms of the three critic
        outputs and of the three loss terms, scalar d_loss; G: image g_sum, scalar g_loss (the reference defines a histogram z_sum but
        merges it into neither, so the event file has no 'z' tag) — from the values the iteration computed (the reference fetches
        them in the same sess.run as the optimizer steps)."""
        from ...utils import summary as S
        np_ = lambda t: t.detach().float().cpu().numpy()
        d, g = out['d'], out['g']
        self.writer.add_summary([S.histogram('d_real_mismatch_sum', np_(d['D_real_mismatch'])), S.histogram('d_real_match_sum', np_(d['D_real_match'])),
                                 S.histogram('d_synthetic_sum', np_(d['D_synthetic'])), S.histogram('d_synthetic_sum_loss', np_(d['D_synthetic_loss'])),
                                 S.histogram('d_real_mismatch_sum_loss', np_(d['D_real_mismatch_loss'])),
                                 S.histogram('d_real_match_sum_loss', np_(d['D_real_match_loss'])), S.scalar('d_loss', float(d['D_loss']))], counter)
        self.writer.add_summary([S.image('g_sum', np_(g['G'])), S.scalar('g_loss', float(g['G_loss']))], counter)
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
                feed = self.make_feed()
                out = self.iteration(feed)
                if getattr(self, 'writer', None) is not None:
                    self.write_summaries(counter, feed, out)
                if counter % 10 == 0:
                    log('Epoch: [%2d] [%4d/%4d] time: %4.4f, d_loss: %.8f, g_loss: %.8f' % (
                        epoch, idx, updates_per_epoch, time.time() - t0, float(out['d']['D_loss']), float(out['g']['G_loss'])))
                counter += 1
                if max_updates is not None and counter > max_updates:
                    return out
        return out
