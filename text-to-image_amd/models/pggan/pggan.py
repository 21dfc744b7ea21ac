"""PGGAN — the reference's conditional progressive-growing GAN (reference models/pggan/pggan.py:12-394) on libt2i_hip.so
kernels.

Same class, constructor arguments and method names as the reference.  One `PGGAN` object = one (stage, trans) of the
schedule in train_pggan.py:17-69: output size 4 * 2^(stage-1); with `trans` the new resolution is faded in with
alpha = iter / steps next to the up-scaled previous `to_rgb` (generator) / the pooled-input `from_rgb` (critic).
The losses reuse the hot path's WGAN-GP machinery (gradient penalty through a double-differentiable critic); new
here: per-sample layer norm (generator only), 2x2 average pool and nearest x2 upscale (a pair of adjoint kernels, closed
under differentiation — the critic is differentiated twice), the fade-in mix.  Reference specifics kept: penalty
coefficient 200, no kt term, G = -D_fake + 5 KL, Adam(2e-6, beta1=0, beta2=0.99) hard-coded (pggan.py:104-110; the
`learning_rate` placeholder is fed but unused), eps of x_hat drawn in-graph (pggan.py:68 overrides the placeholder),
`to_rgb` = k2 s1 SAME 9-channel relu conv + 1x1, every kernel He-initialised by utils/ops.py's defaults.
`fmap_base` / `fmap_max` / the three sizes generalise the hard-coded 1024 / 512 / 128 / 1024 / 128 (defaults = the
reference's) so that the golden step can be tiny.
alpha: the reference assigns `alpha_tra = iter / steps` under a control dependency of D_optim only (pggan.py:76-77,112);
whether the critic step's own forward sees the new or the previous value is a TF scheduling race.  Here alpha is set
from `iter` BEFORE the critic step and kept for the generator step and the sampler — the assign-first order."""
import sys
import time

import torch

from ... import autograd as A
from ... import kernels as K
from ... import optim
from ... import scope as S
from ...utils.ops import concat_tile, conv2d, fc, layer_norm, lerp, lrelu_act, pool, relu, upscale


class PGGAN(object):
    def __init__(self, batch_size, steps, check_dir_write, check_dir_read, dataset, sample_path, log_dir, stage, trans,
                 build_model=True, device=None, seed=0, store=None, fmap_base=1024, fmap_max=512, z_dim=128, embed_dim=1024,
                 compr_embed_dim=128, dp=None):
        self.batch_size, self.steps = batch_size, steps
        self.check_dir_write, self.check_dir_read = check_dir_write, check_dir_read
        self.dataset, self.sample_path, self.log_dir = dataset, sample_path, log_dir
        self.stage, self.trans = stage, trans
        self.z_dim, self.embed_dim, self.compr_embed_dim = z_dim, embed_dim, compr_embed_dim
        self.fmap_base, self.fmap_max = fmap_base, fmap_max
        self.out_size = self.output_size = 4 * pow(2, stage - 1)
        self.channel = 3
        self.sample_num = 64
        self.lr = 0.00005
        self.lr_inp = self.lr
        self.store = S.set_default_store(store or S.VariableStore(device=device, seed=seed))
        self.device = self.store.device
        self.alpha_tra = 0.0                      # tf.Variable(0.0, trainable=False, name='alpha_tra')
        self._alpha_dev = torch.zeros(1, device=self.device)      # ... kept in device memory: graph-replayable fade-in
        self._graphs = None
        # data parallelism (BASELINE config 5 "DP=8"; the reference is single-device): an optional dp.DataParallel — replicas
        # at local batch `batch_size`, critic and generator gradients all-reduced over RCCL, bucketed and overlapped with the
        # backward when eager, exchanged between captured graph segments under replay (same contract as models/wgancls)
        self.dp = dp
        self._capturing = False
        if build_model:
            self.build_model()
            self.define_losses()

    # ---- graph ---------------------------------------------------------------------------------------------------------
    def build_model(self):
        """pggan.py:44-82: variable creation by a launch-free dry pass, generator first."""
        B, dev = self.batch_size, self.device
        with K.dry_run(), torch.no_grad():
            z = torch.empty(B, self.z_dim, device=dev)
            cond = torch.empty(B, self.embed_dim, device=dev)
            G, _, _ = self.generator(z, cond, stages=self.stage, t=self.trans)
            self.discriminator(G, cond, reuse=False, stages=self.stage, t=self.trans)
        self.d_vars = S.trainable_variables('d_net')
        self.g_vars = S.trainable_variables('g_net')
        self.d_arena = optim.Arena(self.d_vars)
        self.g_arena = optim.Arena(self.g_vars)
        self.d_arena.enable_sinks()
        self.g_arena.enable_sinks()

    def get_gradient_penalty(self, x, y):
        with A.input_grads_only():
            grad_y, = torch.autograd.grad(y.sum(), [x], create_graph=True)
        return self._penalty(grad_y)

    get_gradient_penalty2 = get_gradient_penalty

    @staticmethod
    def _penalty(grad_y):
        slopes = A.GpSlopesFn.apply(grad_y)
        return torch.mean(torch.clamp(slopes - 1.0, min=0.0) ** 2)

    def define_losses(self):
        """pggan.py:84-130 (the optimizers; the loss expressions are in d_losses / g_losses)."""
        self.gp_coeff, self.kl_coeff = 200.0, 5.0
        self.D_optimizer = optim.AdamTF(self.d_arena, 0.0, 0.99)
        self.G_optimizer = optim.AdamTF(self.g_arena, 0.0, 0.99)
        self.adam_lr = 0.000002

    def _noise(self, feed, key, like):
        n = feed.get(key)
        if n is None:
            n = K.trunc_normal_(torch.empty_like(like))
        return n

    def d_losses(self, feed):
        """What sess.run([D_optim, D_loss]) evaluates before the update (pggan.py:62-73,84-100).  Gradients -> d_arena."""
        x, xm, cond, z = feed['x'], feed['x_mismatch'], feed['cond'], feed['z']
        B, st, t = x.shape[0], self.stage, self.trans
        eps = feed.get('eps_graph')
        if eps is None:        # tf.random_uniform([B,1,1,1]) in the graph: the fed `epsilon` placeholder is shadowed
            eps = torch.rand(B, device=x.device)
        with torch.no_grad():
            self._ca = self._noise(feed, 'ca_noise_d', cond[:, :self.compr_embed_dim])
            G, _, _ = self.generator(z, cond, stages=st, t=t, reuse=True)
            x_hat = K.interp(eps.reshape(B, 1, 1, 1).contiguous(), G, x)
        # the critic has no batch coupling: D(G), D(x), D(x_mismatch) as one pass over 3B samples
        logits = self.discriminator(torch.cat([G, x, xm], 0), torch.cat([cond, cond, cond], 0), reuse=True, stages=st, t=t).view(3, B)
        Dg_logit, Dx_logit, Dxmi_logit = logits[0], logits[1], logits[2]
        x_hat.requires_grad_(True)
        cond_inp = (cond + 0.0).requires_grad_(True)
        Dx_hat_logit = self.discriminator(x_hat, cond_inp, reuse=True, stages=st, t=t)
        with A.input_grads_only():
            gx, gc = torch.autograd.grad(Dx_hat_logit.sum(), [x_hat, cond_inp], create_graph=True)
        real_gp, real_gp2 = self._penalty(gx), self._penalty(gc)
        D_loss_real, D_loss_fake, D_loss_mismatch = Dx_logit.mean(), Dg_logit.mean(), Dxmi_logit.mean()
        wdist, wdist2 = D_loss_real - D_loss_fake, D_loss_real - D_loss_mismatch
        D_loss = -wdist - wdist2 + self.gp_coeff * (real_gp + real_gp2)
        self.d_arena.zero_grad()
        if self.dp is not None and not self._capturing:
            self.dp.arm(self.d_arena)           # bucketed all-reduce overlaps the rest of this backward
        D_loss.backward(inputs=list(self.d_vars.values()))
        A.side_join()
        return dict(D_loss=D_loss.detach(), wdist=wdist.detach(), wdist2=wdist2.detach(), real_gp=real_gp.detach(),
                    real_gp2=real_gp2.detach(), reg_loss=(Dxmi_logit.detach() ** 2).mean(), G=G, Dx_hat_logit=Dx_hat_logit.detach(),
                    D_loss_real=D_loss_real.detach(), D_loss_fake=D_loss_fake.detach(), D_loss_mismatch=D_loss_mismatch.detach())

    def g_losses(self, feed):
        cond, z = feed['cond'], feed['z']
        self._ca = self._noise(feed, 'ca_noise_g', cond[:, :self.compr_embed_dim])
        G, mean, log_sigma = self.generator(z, cond, stages=self.stage, t=self.trans, reuse=True)
        with self.store.frozen('d_net'):
            Dg_logit = self.discriminator(G, cond, reuse=True, stages=self.stage, t=self.trans)
        G_kl_loss = self.kl_std_normal_loss(mean, log_sigma)
        G_loss = -Dg_logit.mean() + self.kl_coeff * G_kl_loss
        self.g_arena.zero_grad()
        if self.dp is not None and not self._capturing:
            self.dp.arm(self.g_arena)
        G_loss.backward(inputs=list(self.g_vars.values()))
        A.side_join()
        return dict(G_loss=G_loss.detach(), G_kl_loss=G_kl_loss.detach(), G=G.detach(), D_loss_fake=Dg_logit.detach().mean())

    def set_alpha(self, value):
        self.alpha_tra = float(value)
        self._alpha_dev.fill_(self.alpha_tra)

    def _d_body(self, feed):
        d = self.d_losses(feed)
        scale = self.dp.allreduce_arena(self.d_arena) if self.dp is not None else 1.0
        self.D_optimizer.apply(grad_scale=scale)
        return d

    def _g_body(self, feed):
        g = self.g_losses(feed)
        scale = self.dp.allreduce_arena(self.g_arena) if self.dp is not None else 1.0
        self.G_optimizer.apply(grad_scale=scale)
        return g

    def enable_graphs(self, feed):
        """Capture the two halves of the iteration into hipGraphs (call after one eager iteration).  alpha then lives in
        device memory, and the in-graph random draws of the reference (eps of x_hat, the conditioning noise) become static
        buffers that are re-drawn before every replay."""
        from ...graphs import StepGraphs
        B, dev = feed['x'].shape[0], self.device
        feed = dict(feed)
        for k, shape in (('eps_graph', (B,)), ('ca_noise_d', (B, self.compr_embed_dim)), ('ca_noise_g', (B, self.compr_embed_dim))):
            if feed.get(k) is None:
                feed[k] = torch.empty(shape, device=dev)
        self._graphs = StepGraphs(feed, ('x', 'x_mismatch', 'cond', 'z', 'eps_graph', 'ca_noise_d', 'ca_noise_g'),
                                  filters=(self.d_arena.flat, self.g_arena.flat))
        self._redraw(feed)
        self._graphs.load(feed)
        if self.dp is None:
            self._graphs.capture('d', self._d_body)
            self._graphs.capture('g', self._g_body)
            return
        # data parallelism: each half is cut at its exchange step — [losses + backward] | all-reduce (eager, never captured)
        # | [Adam]; thread-local capture mode because the process group's watchdog thread polls events meanwhile
        scale = 1.0 / self.dp.world
        self._capturing = True
        try:
            self._graphs.capture('d', self.d_losses, capture_error_mode='thread_local')
            self._graphs.capture('d_upd', lambda f: self.D_optimizer.apply(grad_scale=scale), capture_error_mode='thread_local', refresh=False)
            self._graphs.capture('g', self.g_losses, capture_error_mode='thread_local')
            self._graphs.capture('g_upd', lambda f: self.G_optimizer.apply(grad_scale=scale), capture_error_mode='thread_local', refresh=False)
        finally:
            self._capturing = False

    def _redraw(self, feed):
        st = self._graphs.static
        if feed.get('eps_graph') is None or feed['eps_graph'] is st['eps_graph']:
            st['eps_graph'].uniform_(0.0, 1.0)
        for k in ('ca_noise_d', 'ca_noise_g'):
            if feed.get(k) is None or feed[k] is st[k]:
                K.trunc_normal_(st[k])

    def iteration(self, idx, feed):
        """One D update then one G update (pggan.py:196-197)."""
        self.set_alpha(float(idx) / float(self.steps))           # alpha_assign (see the module docstring)
        if self._graphs is not None:
            self._redraw(feed)
            self._graphs.load(feed)
            self.D_optimizer.prepare(self.adam_lr)
            d = self._graphs.replay('d')
            if self.dp is not None:
                self.dp.allreduce_arena(self.d_arena)
                self._graphs.replay('d_upd')
            self.G_optimizer.prepare(self.adam_lr)
            g = self._graphs.replay('g')
            if self.dp is not None:
                self.dp.allreduce_arena(self.g_arena)
                self._graphs.replay('g_upd')
            return {'d': d, 'g': g}
        self.D_optimizer.prepare(self.adam_lr)
        d = self._d_body(feed)
        self.G_optimizer.prepare(self.adam_lr)
        return {'d': d, 'g': self._g_body(feed)}

    def sampler(self, z_sample, cond_sample):
        with torch.no_grad():
            self._ca = None
            return self.generator(z_sample, cond_sample, reuse=True, stages=self.stage, t=self.trans)[0]

    # ---- networks ------------------------------------------------------------------------------------------------------
    def discriminator(self, inp, cond, stages, t, reuse=False):
        """-> logits [B]  (pggan.py:251-281)"""
        alpha_trans = self._alpha_dev
        act = lrelu_act()
        with S.variable_scope('d_net', reuse=reuse):
            x_iden = None
            if t:
                x_iden = self.from_rgb(pool(inp, 2), stages - 2)
            x = self.from_rgb(inp, stages - 1)
            for i in range(stages - 1, 0, -1):
                with S.variable_scope(self.get_conv_scope_name(i), reuse=reuse):
                    x = conv2d(x, f=self.get_dnf(i), ks=(3, 3), s=(1, 1), act=act)
                    x = conv2d(x, f=self.get_dnf(i - 1), ks=(3, 3), s=(1, 1), act=act)
                    x = pool(x, 2)
                if i == stages - 1 and t:
                    x = lerp(x_iden, x, alpha_trans)              # alpha * x + (1 - alpha) * x_iden
            with S.variable_scope(self.get_conv_scope_name(0), reuse=reuse):
                cond_compress = fc(cond, units=self.compr_embed_dim, act=act)
                concat = self.concat_cond4(x, cond_compress)
                x_b1 = conv2d(concat, f=self.get_dnf(0), ks=(3, 3), s=(1, 1), act=act)
                x_b1 = conv2d(x_b1, f=self.get_dnf(0), ks=(4, 4), s=(1, 1), padding='VALID', act=act)
                output_b1 = fc(x_b1.reshape(x_b1.shape[0], -1), units=1)      # dense on the [B,1,1,C] map
            return output_b1.reshape(-1)

    def generator(self, z_var, cond_inp, stages, t, reuse=False, cond_noise=True):
        """-> (image NHWC, mean, log_sigma)  (pggan.py:283-316)"""
        alpha_trans = self._alpha_dev
        with S.variable_scope('g_net', reuse=reuse):
            with S.variable_scope(self.get_conv_scope_name(0), reuse=reuse):
                mean_lr, log_sigma_lr = self.generate_conditionals(cond_inp)
                cond = self.sample_normal_conditional(mean_lr, log_sigma_lr, cond_noise)
                x = torch.cat([z_var, cond], 1)
                x = fc(x, units=4 * 4 * self.get_nf(0))
                x = layer_norm(x)
                x = x.reshape(-1, 4, 4, self.get_nf(0))
                x = conv2d(x, f=self.get_nf(0), ks=(3, 3), s=(1, 1))
                x = layer_norm(x, act=relu)
                x = conv2d(x, f=self.get_nf(0), ks=(3, 3), s=(1, 1))
                x = layer_norm(x, act=relu)
            x_iden = None
            for i in range(1, stages):
                if (i == stages - 1) and t:
                    x_iden = self.to_rgb(x, stages - 2)
                    x_iden = upscale(x_iden, 2)
                with S.variable_scope(self.get_conv_scope_name(i), reuse=reuse):
                    x = upscale(x, 2)
                    x = conv2d(x, f=self.get_nf(i), ks=(3, 3), s=(1, 1))
                    x = layer_norm(x, act=relu)
                    x = conv2d(x, f=self.get_nf(i), ks=(3, 3), s=(1, 1))
                    x = layer_norm(x, act=relu)
            x = self.to_rgb(x, stages - 1)
            if t:
                x = lerp(x_iden, x, alpha_trans)                  # (1 - alpha) * x_iden + alpha * x
            return x, mean_lr, log_sigma_lr

    def concat_cond4(self, x, cond):
        return concat_tile(x, cond)

    def get_rgb_name(self, stage):
        return 'rgb_stage_%d' % stage

    def get_conv_scope_name(self, stage):
        return 'conv_stage_%d' % stage

    def get_dnf(self, stage):
        return min(self.fmap_base // (2 ** stage) * 2, self.fmap_max)

    def get_nf(self, stage):
        return min(self.fmap_base // (2 ** stage) * 4, self.fmap_max)

    def from_rgb(self, x, stage):
        with S.variable_scope(self.get_rgb_name(stage), reuse=S.default_store().reuse()):
            return conv2d(x, f=self.get_dnf(stage), ks=(1, 1), s=(1, 1), act=lrelu_act())

    def to_rgb(self, x, stage):
        with S.variable_scope(self.get_rgb_name(stage), reuse=S.default_store().reuse()):
            x = conv2d(x, f=9, ks=(2, 2), s=(1, 1), act=relu)
            return conv2d(x, f=3, ks=(1, 1), s=(1, 1))

    def generate_conditionals(self, embeddings, units=None):
        units = units or self.compr_embed_dim
        with K.f32_outputs():          # conditioning statistics stay fp32 in every storage mode
            return fc(embeddings, units, act=lrelu_act()), fc(embeddings, units, act=lrelu_act())

    def sample_normal_conditional(self, mean, log_sigma, cond_noise=True):
        if not cond_noise:
            return mean
        eps = getattr(self, '_ca', None)
        if eps is None or eps.shape != mean.shape:
            eps = K.trunc_normal_(torch.empty_like(mean))
        return mean + torch.exp(log_sigma) * eps

    def kl_std_normal_loss(self, mean, log_sigma):
        return torch.mean(-log_sigma + 0.5 * (-1.0 + torch.exp(2.0 * log_sigma) + mean * mean))

    def get_variables_up_to_stage(self, stages):
        """Names of the variables a stage's checkpoint holds (pggan.py:380-386): the stage's rgb layers + every conv stage
        below it, critic then generator."""
        d = list(self.store.global_variables('d_net/%s/' % self.get_rgb_name(stages - 1)))
        g = list(self.store.global_variables('g_net/%s/' % self.get_rgb_name(stages - 1)))
        for stage in range(stages):
            d += list(self.store.global_variables('d_net/%s/' % self.get_conv_scope_name(stage)))
            g += list(self.store.global_variables('g_net/%s/' % self.get_conv_scope_name(stage)))
        return d + g

    # ---- training loop (pggan.py:147-247) ---------------------------------------------------------------------------------
    def make_feed(self, gen):
        images, wrong_images, embed, _, _ = self.dataset.train.next_batch(self.batch_size, 4, wrong_img=True, embeddings=True)
        return {'x': images, 'x_mismatch': wrong_images, 'cond': embed,
                'z': torch.randn((self.batch_size, self.z_dim), generator=gen, device=self.device)}

    def define_summaries(self):
        """reference pggan.py:132-156,165: the FileWriter on log_dir (utils/summary.py: TensorBoard event files without TensorFlow)."""
        from ...utils.summary import FileWriter
        self.writer = FileWriter(self.log_dir)

    def write_summaries(self, idx, feed, out, sample_z=None):
        """The merged summary of reference pggan.py:132-156 at step idx (written every 20 steps, pggan.py:220-222): images `x` and
        `G_img`, histograms `z` / `z_sample`, and the twelve scalars under the reference's tags.  The reference evaluates them in a
        third sess.run after the two updates; here they are the values the iteration itself computed, as in models/wgancls."""
        from ...utils import summary as S
        d, g = out['d'], out['g']
        np_ = lambda t: t.detach().float().cpu().numpy()
        vals = [S.image('x', np_(feed['x'])), S.image('G_img', np_(g['G'])), S.histogram('z', np_(feed['z']))]
        if sample_z is not None:
            vals.append(S.histogram('z_sample', np_(sample_z)))
        vals += [S.scalar('G_loss_wass', -float(g['D_loss_fake'])), S.scalar('kl_loss', float(g['G_kl_loss'])), S.scalar('G_loss', float(g['G_loss']))]
        for tag, key in (('D_loss_real', 'D_loss_real'), ('D_loss_fake', 'D_loss_fake'), ('real_gp', 'real_gp'), ('D_loss', 'D_loss'),
                         ('reg_loss', 'reg_loss'), ('wdist', 'wdist'), ('wdist2', 'wdist2'), ('d_loss_mismatch', 'D_loss_mismatch'),
                         ('real_gp2', 'real_gp2')):
            vals.append(S.scalar(tag, float(d[key])))
        self.writer.add_summary(vals, idx)
        self.writer.flush()

    def train(self, max_steps=None, log=None, side_effects=False, summaries=False):
        """Stage schedule semantics of pggan.py:147-247: a transition stage restores the previous stage's variables
        (`get_variables_up_to_stage(stage - 1)`) from check_dir_read, a stabilisation stage its own; new variables keep
        their fresh initialisation; checkpoints of `get_variables_up_to_stage(stage)` go to check_dir_write."""
        from ...utils.saver import Saver, load, save
        from ...utils.utils import get_balanced_factorization, save_captions, save_images
        log = log or (lambda s: (sys.stdout.write(s + '\n'), sys.stdout.flush()))
        saver = Saver(self.store, var_list=self.get_variables_up_to_stage(self.stage), max_to_keep=2)
        if side_effects and self.stage != 1:
            src = Saver(self.store, var_list=self.get_variables_up_to_stage(self.stage - 1)) if self.trans else saver
            could_load, _ = load(src, None, self.check_dir_read)
            if not could_load:
                raise RuntimeError('Could not load previous stage during transition' if self.trans else 'Could not load current stage')
        gen = torch.Generator(device=self.device).manual_seed(1234)
        if side_effects:
            sample_z = torch.randn((self.sample_num, self.z_dim), generator=gen, device=self.device)
            _, sample_cond, _, captions = self.dataset.test.next_batch_test(self.sample_num, 0, 1)
            sample_cond = sample_cond[0]
            save_captions(self.sample_path, captions)
        end = min(self.steps, max_steps) if max_steps is not None else self.steps
        t0 = time.time()
        out = None
        if summaries and self.log_dir:
            self.define_summaries()
        for idx in range(1, end):
            feed = self.make_feed(gen)
            out = self.iteration(idx, feed)
            if idx % 20 == 0 and getattr(self, 'writer', None) is not None:
                self.write_summaries(idx, feed, out, sample_z if side_effects else None)
            if idx % 20 == 0:
                epoch = idx // max(self.dataset.train.num_examples // self.batch_size, 1)
                log('Epoch: [%2d] [%4d] time: %4.4f, d_loss: %.8f, g_loss: %.8f' % (
                    epoch, idx, time.time() - t0, float(out['d']['D_loss']), float(out['g']['G_loss'])))
            if side_effects and idx % 2000 == 0:
                samples = torch.clamp(self.sampler(sample_z, sample_cond), -1.0, 1.0)
                save_images(samples, get_balanced_factorization(samples.shape[0]), '{}train_{:02d}_{:04d}.png'.format(self.sample_path, 0, idx))
            if side_effects and (idx % 2000 == 0 or idx == end - 1):      # end - 1 == steps - 1 unless max_steps truncates
                save(saver, None, self.check_dir_write, idx)
        return out
