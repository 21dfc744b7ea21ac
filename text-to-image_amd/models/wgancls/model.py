"""WGanCls — the reference's conditional WGAN-GP (reference models/wgancls/model.py:1-225) on libt2i_hip.so kernels.

Same class, constructor, method names and argument meaning as the reference.  The TF-1 graph/session split becomes:
  build_model()    creates every variable (a launch-free dry run of G and D, so it also works without a GPU) and the
                   two parameter arenas; the placeholders of model.py:36-46 become the keys of the ``feed`` dict
  define_losses()  creates kt and the two Adam optimizers (model.py:72-106)
  d_step(feed) / g_step(feed)
                   what ``sess.run([D_optim, kt_optim, D_loss])`` / ``sess.run([G_optim, G_loss])`` evaluate
                   (reference trainer.py:97,101): losses and gradients at pre-update values, then the updates.
Scheduling differences that do not change the mathematics: the three gradient-free critic passes
D(G), D(x), D(x_mismatch) run as ONE batched pass of 3B samples (the critic has no batch norm, samples are independent
and weights shared, model.py:49-51), and the zero-valued branches of the double backward are never launched.
"""
import os

import torch

from ... import autograd as A
from ... import kernels as K
from ... import optim
from ... import scope as S
from ... import stacked as ST
from ...utils import roctx
from ...utils.ops import (NCHW, add, batch_norm, concat_tile, conv2d, conv2d_transpose, fc, fork, lrelu_act, relu,
                          reshape_to_map, tanh, to_nchw, to_nhwc, update_ops)


# capture mode of the data-parallel graph segments: thread-local, because the process group's watchdog thread polls events
# while the capture is open (T2I_DP_CAPTURE_MODE=global for diagnostics with a backend that has no such thread)
_CAPTURE_MODE = os.environ.get('T2I_DP_CAPTURE_MODE', 'thread_local')
# one-graph iteration (single GPU): issue the G step's generator forward on a second stream beside the critic step
_OVERLAP_G_FORWARD = os.environ.get('T2I_OVERLAP_G_FORWARD', '1') != '0'
# the one-graph iteration trusts the critic's filter images its previous replay regenerated behind the critic's Adam step instead
# of regenerating them again at its head (dg_step keeps them current across outside writers): T2I_TRUST_IMAGES=0 restores the refresh
_TRUST_IMAGES = os.environ.get('T2I_TRUST_IMAGES', '1') != '0'
# single GPU: the critic's four passes of a step as ONE stacked pass of 4B images (stacked.py); T2I_STACK_XHAT=0: the 3B + B form
_STACK_XHAT = os.environ.get('T2I_STACK_XHAT', '1') != '0'
# single GPU, D + G iteration: the generator's two evaluations (critic step: no gradient, its own noise; generator step: under UPDATE_OPS)
# as ONE stacked pass of 2B rows with per-evaluation batch-norm statistics (stacked.py); T2I_PAIR_G=0: two passes of B
_PAIR_G = os.environ.get('T2I_PAIR_G', '1') != '0'


class WGanCls(object):
    def __init__(self, cfg, build_model=True, device=None, seed=0, dp=None):
        """
        Args:
          cfg: Config specifying all the parameters of the model (reference cfg/flowers.yml keys).
          device/seed/dp: where the variables live, the initializer seed, an optional dp.DataParallel (RCCL)
        """
        self.cfg = cfg
        m, t = cfg.MODEL, cfg.TRAIN
        self.batch_size, self.sample_num = t.BATCH_SIZE, t.SAMPLE_NUM
        self.output_size = m.OUTPUT_SIZE
        self.z_dim, self.embed_dim, self.compressed_embed_dim = m.Z_DIM, m.EMBED_DIM, m.COMPRESSED_EMBED_DIM
        self.gf_dim, self.df_dim = m.GF_DIM, m.DF_DIM
        self.image_dims = [m.IMAGE_SHAPE.H, m.IMAGE_SHAPE.W, m.IMAGE_SHAPE.D]
        if self.output_size != 64:
            raise ValueError('the reference tiles the text code over a fixed 4x4 map (model.py:154): OUTPUT_SIZE must be 64')
        self.store = S.set_default_store(S.VariableStore(device=device, seed=seed))
        self.device = self.store.device
        self.dp = dp
        self.dp_cut_eager = os.environ.get('T2I_DP_CUT_EAGER') == '1'
        self.dp_schedule = None
        self.global_step = 0
        self._graphs = None
        self._keep_cut = False
        self._d_cut = self._g_cut = None
        self._capturing = False
        self._consts = {}
        self._kl = None
        # per-network arithmetic: {'g_net': (math, storage)} — layers of that network are created under kernels.math_scope
        # the critic's four passes of a step as one stacked pass of 4B images (stacked.py); False: the 3B + B form of rounds 1-5
        self.stack_xhat = _STACK_XHAT
        self.pair_g = _PAIR_G                 # (needs stack_xhat: the pair writes both images into the stacked critic input's buffer)
        self.net_math = {}
        K.forget_scopes()          # a scoped model that lived in this process before leaves no twin policy behind for this one
        if os.environ.get('T2I_G_MATH'):
            gm = os.environ['T2I_G_MATH'].split(',')     # "f32" or "f32,bf16" (forward arithmetic[, backward arithmetic])
            self.net_math['g_net'] = (gm[0], 'f32') + tuple(gm[1:2])

        if build_model:
            self.build_model()
            self.define_losses()

    # ------------------------------------------------------------------------------------------------------------------
    def build_model(self):
        """Variable creation = one dry (launch-free) pass of generator and discriminator on empty tensors."""
        B, dev = self.batch_size, self.device
        with K.dry_run(), torch.no_grad():
            z = torch.empty(B, self.z_dim, device=dev)
            cond = torch.empty(B, self.embed_dim, device=dev)
            G, _, _ = self.generator(z, cond, reuse=False)
            self.discriminator(G, cond, reuse=False)
        self.d_vars = S.trainable_variables('d_net')
        self.g_vars = S.trainable_variables('g_net')
        self.d_arena = optim.Arena(self.d_vars)
        self.g_arena = optim.Arena(self.g_vars)
        # gradients are accumulated by the kernels' epilogues straight into the optimizer arenas (autograd.SINKS); with
        # data parallelism the bucket overlap follows autograd.NOTIFY instead of AccumulateGrad hooks (dp.py)
        self.d_arena.enable_sinks()
        self.g_arena.enable_sinks()

    def get_gradient_penalty(self, x, y):
        """reference model.py:62-65: one-sided penalty on the per-sample gradient norm of y wrt x."""
        with A.input_grads_only():
            grad_y, = torch.autograd.grad(y.sum(), [x], create_graph=True)
        return self._penalty(grad_y)

    def get_gradient_penalty2(self, x, y):
        """reference model.py:67-70 (same, for the rank-2 text embedding)."""
        return self.get_gradient_penalty(x, y)

    def _const_like(self, t, value):
        """A cached constant tensor shaped like `t` (backward seeds): filled once, so nothing is launched per step."""
        key = (tuple(t.shape), float(value), t.device)
        c = self._consts.get(key)
        if c is None:
            c = torch.full(tuple(t.shape), float(value), dtype=torch.float32, device=t.device)
            # a tensor first created while a graph is being captured lives in that graph's pool and is filled by a node of
            # THAT graph only: caching it would hand other graphs (and eager steps) memory nobody ever filled for them
            if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
                self._consts[key] = c
        return c

    def _ones_like(self, t):
        return self._const_like(t, 1.0)

    @staticmethod
    def _penalty(grad_y):
        slopes = A.GpSlopesFn.apply(grad_y)                       # wave-reduced per-sample L2 norm
        return torch.mean(torch.clamp(slopes - 1.0, min=0.0) ** 2)  # [B] scalars

    def define_losses(self):
        self.kl_coeff = float(self.cfg.TRAIN.COEFF.KL)
        self.lambda1 = float(self.cfg.TRAIN.COEFF.LAMBDA)   # read but unused by the reference too (model.py:75,91)
        self.gp_coeff = 150.0                               # hard-coded in the reference (model.py:91)
        self.kt = torch.tensor(0.7, dtype=torch.float32, device=self.device)
        self.kt_lr = 0.001
        b1, b2 = float(self.cfg.TRAIN.BETA1), float(self.cfg.TRAIN.BETA2)
        self.D_optim = optim.AdamTF(self.d_arena, b1, b2)
        self.G_optim = optim.AdamTF(self.g_arena, b1, b2)

    # ------------------------------------------------------------------------------------------------------------------
    def _ca_noise(self, feed, key, like):
        n = feed.get(key)
        if n is None:    # tf.truncated_normal(tf.shape(mean)) resampled per run (model.py:119)
            n = torch.empty_like(like)
            K.trunc_normal_(n)
        return n

    # Where the data-parallel graph schedule cuts the two backward passes so that the exchange of the gradients that are final
    # first starts while the rest of the backward still runs ("launch as each bwd-filter completes"):
    #   critic: at the input of Conv_3.  Backward order is Conv_9 ... Conv_3 | Conv_2, Conv_1, Conv: the first part leaves
    #           105.5 of the arena's 116 MB final (Conv_7 42 MB, Conv_3 34 MB, Conv_6 19 MB are produced first); the gradient
    #           penalty's second-order contributions to ALL layers also fall into the first part (the double-backward chain runs
    #           Conv -> Conv_9 before the main pass runs back), so after it only the main-pass terms of Conv..Conv_2 are missing.
    #   generator: at the 4x4 -> 8x8 boundary (output of the first bottleneck).  First part: the critic's input gradient and the
    #           generator from out_conv back to Conv2d_transpose (57 MB of 90.6); rest: the 4x4 bottleneck, dense_2, the two
    #           conditioning heads (+ the KL term's gradient, which only reaches those heads).
    _CUT_D = 'd_net/Conv_3/weights'                 # first variable (creation order = arena order) of the critic's FIRST backward part
    _CUT_G = 'g_net/Conv2d_transpose/weights'       # ... of the generator's

    def _cut_ranges(self, arena, first_var):
        """([(start, end)] of the part that is final after the first half of the cut backward, [(start, end)] of the rest)."""
        if first_var not in arena.offsets:
            raise KeyError('cut schedule: %r is not a variable of this arena (a model variant needs its own _CUT_D / _CUT_G)' % first_var)
        # the cut is "everything created from first_var on": only valid when arena order == creation order
        offs = [arena.offsets[n][0] for n in arena.names]
        if any(b <= a for a, b in zip(offs, offs[1:])):
            raise RuntimeError('cut schedule: arena offsets are not monotone in creation order; use the uncut schedule')
        o = arena.offsets[first_var][0]
        return [(o, arena.numel)], [(0, o)]

    def _d_cut_ranges(self):
        """The critic arena's two exchange ranges of the cut schedule; the stacked step finishes the whole arena at once."""
        if self.stack_xhat and self.device.type == 'cuda':
            return [(0, self.d_arena.numel)], []
        return self._cut_ranges(self.d_arena, self._CUT_D)

    @staticmethod
    def _split_vars(variables, first_var):
        names = list(variables)
        if first_var not in variables:
            raise KeyError('cut schedule: %r is not among the variables %s...' % (first_var, names[:3]))
        i = names.index(first_var)
        return [variables[n] for n in names[i:]], [variables[n] for n in names[:i]]

    def _stack_buffers(self, B, like):
        """The stacked critic step's persistent inputs for batch B: [G | x | x_mismatch | x_hat] images, the text embedding four
        times, the upstream gradient of the stacked first-order pass ([dD_loss/dlogits (3B) | ones (B)]) and a dummy slope vector."""
        bufs = getattr(self, '_stk', None)
        if bufs is None or bufs['B'] != B:
            if like.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('the stacked critic step allocates its input buffers on its first eager iteration; run one before capturing')
            dev = like.device
            seed4 = torch.zeros(4 * B, dtype=torch.float32, device=dev)
            seed4[3 * B:] = 1.0                  # d sum(D(x_hat)) / d logits: set once, the head rewrites only the first 3B entries
            # images: [G of the generator step | G of the critic step | x | x_mismatch | x_hat]; the critic's stacked input is the last 4B rows
            img5 = torch.zeros((5 * B,) + tuple(self.image_dims), dtype=torch.float32, device=dev)
            bufs = self._stk = {'B': B, 'img5': img5, 'inp4': img5[B:],
                                'cond4': torch.zeros((4 * B, self.embed_dim), dtype=torch.float32, device=dev), 'seed4': seed4,
                                'zeros': torch.zeros(B, dtype=torch.float32, device=dev)}
        return bufs

    def _d_losses_stacked(self, feed, cut=False, have_g=False):
        """d_losses with the critic's four passes as one stacked pass (stacked.py): forward and first-order input-gradient chain on
        4B images; per layer ONE filter-gradient launch over all 4B rows inside the gradient penalty's double backward (the scored
        rows' activations x the loss gradient + the penalty's tangent x its chain), the double backward itself on the x_hat rows.
        Same mathematics as the two-pass form — per-sample arithmetic is unchanged, the sums over the batch are grouped differently
        (so results agree to rounding, not bit for bit, with T2I_STACK_XHAT=0).
        cut (data-parallel graph schedule): the stacked step does not cut the critic's backward — the large filter gradients are the
        LAST things its double backward produces, whichever way it is cut — so the first "part" is the whole step, d_backward_rest()
        is empty and the whole arena leaves in one exchange, which the generator's forward hides (_d_cut_ranges, dg_step)."""
        x, xm, cond, z, eps = feed['x'], feed['x_mismatch'], feed['cond'], feed['z'], feed['epsilon']
        B = x.shape[0]
        R = 3 * B
        bufs = self._stack_buffers(B, x)
        inp4, cond4, seed4 = bufs['inp4'], bufs['cond4'], bufs['seed4']
        del ST._DEFERRED[:]                       # (records of a step that was abandoned half-way)
        if not (x.is_cuda and torch.cuda.is_current_stream_capturing()):
            ST.prepare_side(self.device)
        with torch.no_grad():
            if have_g:                                      # _g_forward_pair has already put this step's G into its slot
                G = inp4[:B]
            else:
                self._noise = self._ca_noise(feed, 'ca_noise_d', cond[:, :self.compressed_embed_dim])
                with K.output_into(inp4[:B]):               # the generator's last kernel writes G into its slot
                    G, _, _ = self.generator(z, cond, reuse=True)
                if G.data_ptr() != inp4.data_ptr():
                    K.axpby(G.contiguous(), 1.0, out=inp4[:B])
                    G = inp4[:B]
            # (copies into the image buffer go through t2i_axpby, not Tensor.copy_: the paired generator's outputs are views of this
            # buffer, and a tensor-library in-place op on any other view of it would invalidate them for autograd)
            for src, slot in ((x, inp4[B:2 * B]), (xm, inp4[2 * B:R])):
                if src.data_ptr() != slot.data_ptr():   # (the captured graphs' static inputs ARE these slots: nothing to copy on replay)
                    K.axpby(src.contiguous(), 1.0, out=slot)
            K.interp(eps, inp4[:B], inp4[B:2 * B], out=inp4[R:])
            cond4.view(4, B, -1).copy_(cond.unsqueeze(0).expand(4, B, cond.shape[1]))
        x_hat = inp4[R:].detach().requires_grad_(True)
        cond_hat = cond4[R:].detach().requires_grad_(True)
        logits = self.discriminator(ST.Stacked(inp4[:R], x_hat), ST.Stacked(cond4[:R], cond_hat), reuse=True)
        lm, lh = logits.main, logits.hat                         # [3B,1,1,1] (fake | real | mismatch), [B,1,1,1]
        # dD_loss/dlogits depends on kt alone: the head is asked for it before the slopes exist, so that the loss gradient of the scored
        # rows and the penalty's first-order chain can run down the layers together
        K.wgan_d_head(lm.detach().reshape(-1), bufs['zeros'], bufs['zeros'], self.kt, self.gp_coeff, seed_l_into=seed4[:R])
        self.d_arena.zero_grad()
        if self.dp is not None and not self._capturing:
            self.dp.arm(self.d_arena)          # bucketed all-reduce overlaps the rest of this step (bias sinks are written from here on)
        with ST.first_order_pass(R):
            gx, gc = torch.autograd.grad([lm, lh], [x_hat, cond_hat], grad_outputs=[seed4[:R].view_as(lm), seed4[R:].view_as(lh)],
                                         create_graph=True)
        slopes1, slopes2 = A.GpSlopesFn.apply(gx), A.GpSlopesFn.apply(gc)
        scal, _, seed_s1, seed_s2 = K.wgan_d_head(lm.detach().reshape(-1), slopes1.detach(), slopes2.detach(), self.kt, self.gp_coeff)
        torch.autograd.backward([slopes1, slopes2], [seed_s1, seed_s2], inputs=list(self.d_vars.values()))
        ST.flush_deferred()                       # (filter gradients the double backward did not reach: none on this model)
        ST.join()                                 # ... and the ones issued on the second stream (T2I_STACK_SIDE)
        A.side_join()
        out = {k: scal[i] for i, k in enumerate(K.D_HEAD_KEYS)}
        i0 = K.D_HEAD_KEYS.index('wdist')
        out['wd_sums'] = scal[i0:i0 + 2].clone() if self.dp is not None else scal[i0:i0 + 2]      # (see d_losses)
        out.update(G=G, Dx_hat_logit=lh.detach(), grad_x_hat=gx.detach(), grad_cond=gc.detach())
        if cut:
            self._d_rest = None
        return out

    def d_losses(self, feed, cut=False, have_g=False):
        """Everything `sess.run([D_optim, kt_optim, D_loss])` evaluates before the updates.  Returns a dict of scalar
        tensors; leaves the critic gradients in the arena (self.d_arena.grad).
        cut=True (data-parallel graph schedule): only the FIRST part of the backward is run (down to the input of Conv_3); the
        caller starts the exchange of that part's gradients and then runs d_backward_rest()."""
        if self.stack_xhat and feed['x'].is_cuda:
            return self._d_losses_stacked(feed, cut, have_g)
        x, xm, cond, z, eps = feed['x'], feed['x_mismatch'], feed['cond'], feed['z'], feed['epsilon']
        B = x.shape[0]
        with torch.no_grad():
            # G in training mode (batch statistics), moving averages NOT updated here (D_optim is outside UPDATE_OPS)
            self._noise = self._ca_noise(feed, 'ca_noise_d', cond[:, :self.compressed_embed_dim])
            G, _, _ = self.generator(z, cond, reuse=True)
            x_hat = K.interp(eps, G, x)
        # D(G), D(x), D(x_mismatch): one batched pass (shared weights, no batch coupling in the critic)
        # (the cut tensor is recorded only when asked for: a reference kept on the model outlives the step and, in the one-graph
        # capture of the single-GPU iteration, is released on another stream than it was made on)
        self._keep_cut = cut
        logits = self.discriminator(torch.cat([G, x, xm], 0), torch.cat([cond, cond, cond], 0), reuse=True).view(3, B)
        self._keep_cut = False
        d_cut, self._d_cut = self._d_cut, None         # of the batched pass (the x_hat pass below gets no first-order gradient)
        Dg_logit, Dx_logit, Dxmi_logit = logits[0], logits[1], logits[2]
        x_hat.requires_grad_(True)
        cond_inp = (cond + 0.0).requires_grad_(True)
        Dx_hat_logit = self.discriminator(x_hat, cond_inp, reuse=True)
        with A.input_grads_only():
            # d(sum of logits)/d(inputs): seeded with ones directly (no .sum() node, no ones_like fill in its backward)
            gx, gc = torch.autograd.grad(Dx_hat_logit, [x_hat, cond_inp], grad_outputs=self._ones_like(Dx_hat_logit),
                                         create_graph=True)
        slopes1, slopes2 = A.GpSlopesFn.apply(gx), A.GpSlopesFn.apply(gc)
        # loss head (model.py:72-92) in one launch: every logged scalar + the seeds dD_loss/dlogits, dD_loss/dslopes
        scal, seed_l, seed_s1, seed_s2 = K.wgan_d_head(logits.detach().reshape(-1), slopes1.detach(), slopes2.detach(), self.kt,
                                                        self.gp_coeff)
        self.d_arena.zero_grad()
        if self.dp is not None and not self._capturing:
            self.dp.arm(self.d_arena)          # bucketed all-reduce overlaps the rest of this backward
        if cut:
            first, rest = self._split_vars(self.d_vars, self._CUT_D)
            torch.autograd.backward([logits, slopes1, slopes2], [seed_l.view_as(logits), seed_s1, seed_s2], inputs=[d_cut] + first)
            self._d_rest = (d_cut, rest)
        else:
            torch.autograd.backward([logits, slopes1, slopes2], [seed_l.view_as(logits), seed_s1, seed_s2],
                                    inputs=list(self.d_vars.values()))
        A.side_join()                          # filter gradients issued on the side stream are in the arena
        out = {k: scal[i] for i, k in enumerate(K.D_HEAD_KEYS)}
        # (wdist, wdist2): what the kt step needs.  Under data parallelism these two batch means are summed over the ranks
        # next to the gradient arena (balance_loss is quadratic in them, so the per-rank kt gradients must not be averaged);
        # a private copy, because the exchange works in place and the logged scalars should stay this rank's own
        i0 = K.D_HEAD_KEYS.index('wdist')
        out['wd_sums'] = scal[i0:i0 + 2].clone() if self.dp is not None else scal[i0:i0 + 2]
        out.update(G=G, Dx_hat_logit=Dx_hat_logit.detach(), grad_x_hat=gx.detach(), grad_cond=gc.detach())
        return out

    def d_backward_rest(self):
        """Second part of a cut critic backward: from the gradient at the input of Conv_3 through Conv_2, Conv_1, Conv."""
        if self._d_rest is None:               # the stacked step: nothing was left behind the "cut"
            return
        d_cut, rest = self._d_rest
        self._d_rest = None
        g, d_cut.grad = d_cut.grad, None
        torch.autograd.backward([d_cut], [g], inputs=rest)
        A.side_join()

    def _d_update(self, out, scale):
        """Adam on the critic arena + the kt step; `scale` turns rank-summed gradients (and batch means) into the mean."""
        self.D_optim.apply(grad_scale=scale, refresh=True if self.dp is None else None)   # the generator half reads these filters next
        K.kt_sgd(self.kt, out['wd_sums'], scale, self.kt_lr)          # GradientDescentOptimizer(0.001) on balance_loss (model.py:100)

    def _d_body(self, feed, have_g=False):
        """Device work of the critic step (graph-capturable): losses, backward, [all-reduce], Adam, kt."""
        out = self.d_losses(feed, have_g=have_g)
        scale = 1.0
        if self.dp is not None:
            scale = self.dp.allreduce_arena(self.d_arena, extra=out['wd_sums'])
        self._d_update(out, scale)
        return out

    def d_step(self, feed):
        with roctx.range('wgancls.d_step'):
            return self._d_step(feed)

    def _d_step(self, feed):
        self.D_optim.prepare(float(feed['learning_rate_d']))
        if self._graphs is not None:
            self._load_static(feed, noise=('ca_noise_d',))
            self._graphs['d'].replay()
            out = self._graphs['d_out']
            if self.dp is not None:            # the exchange step runs between the two captured halves
                self.dp.allreduce_arena(self.d_arena, extra=out['wd_sums'])
                self._graphs['d_upd'].replay()
            K.filter_cache_invalidate(external=False)        # the replay rewrote filters (and cached transforms) behind the host's back
        else:
            out = self._d_body(feed)
        self.global_step += 1
        return out

    def _g_forward(self, feed, keep_cut=False):
        """The generator's forward of the G step: depends on the generator's variables only, so under data parallelism
        it can run while the critic's gradients are still being exchanged (dg_step).  keep_cut: remember the tensor at the
        4x4 -> 8x8 boundary for a cut backward (g_losses(cut=True))."""
        cond, z = feed['cond'], feed['z']
        self._noise = self._ca_noise(feed, 'ca_noise_g', cond[:, :self.compressed_embed_dim])
        self._keep_cut = keep_cut
        with update_ops():   # G_optim runs under control_dependencies(UPDATE_OPS) (model.py:102)
            G, mean, log_sigma = self.generator(z, cond, reuse=True)
        self._keep_cut = False
        G_kl = self._kl if self._kl is not None else self.kl_std_normal_loss(mean, log_sigma).reshape(1)
        return G, G_kl

    def _pairing(self, feed):
        return self.pair_g and self.stack_xhat and self.dp is None and feed['z'].is_cuda

    def _g_forward_pair(self, feed):
        """Both generator evaluations of a D + G iteration as ONE stacked pass of 2B rows (stacked.py): the generator step's (leading
        rows: autograd graph, UPDATE_OPS, noise `ca_noise_g`) and the critic step's (rows behind: no gradient, noise `ca_noise_d`,
        moving averages untouched).  Same weights, z and text embedding; the conditioning heads are evaluated once (they do not depend
        on the noise), every batch norm keeps per-evaluation statistics, and the last kernel writes both images into the stacked
        critic input's buffer.  -> ((G, G_kl) for g_losses(fwd=...), and the critic step's G is in its slot: d_losses(have_g=True))."""
        cond, z = feed['cond'], feed['z']
        B = z.shape[0]
        bufs = self._stack_buffers(B, feed['x'])
        like = cond[:, :self.compressed_embed_dim]
        noise_d = self._ca_noise(feed, 'ca_noise_d', like)      # (the order an unpaired iteration draws them in)
        noise_g = self._ca_noise(feed, 'ca_noise_g', like)
        img5 = bufs['img5']
        with update_ops(), ST.moving_groups(1), K.output_into(img5[:2 * B]):       # only the generator step's evaluation moves the moving averages
            Gs, mean, log_sigma = self.generator(z, cond, reuse=True, pair=(noise_g, noise_d))
        if Gs.hat.data_ptr() != img5[B:].data_ptr():             # (the last kernel did not take the buffer: copy the critic step's image)
            with torch.no_grad():
                K.axpby(Gs.hat.contiguous(), 1.0, out=img5[B:2 * B])
        G_kl = self._kl if self._kl is not None else self.kl_std_normal_loss(mean, log_sigma).reshape(1)
        return Gs.main, G_kl

    def g_losses(self, feed, fwd=None, cut=False):
        """cut=True: only the first part of the backward (the critic's input gradient and the generator back to the 4x4 -> 8x8
        boundary); g_backward_rest() runs the remainder."""
        cond = feed['cond']
        G, G_kl = fwd if fwd is not None else self._g_forward(feed, keep_cut=cut)
        g_cut, self._g_cut = (self._g_cut, None) if cut else (None, None)
        with self.store.frozen('d_net'):
            Dg_logit = self.discriminator(G, cond, reuse=True)
        # G_loss = -mean(D(G)) + kl_coeff * KL (model.py:90-92): the KL value came out of the fused conditioning-augmentation
        # kernel; the backward is seeded with dG_loss/dlogit = -1/B and dG_loss/dKL = kl_coeff
        B = Dg_logit.numel()
        self.g_arena.zero_grad()
        if self.dp is not None and not self._capturing:
            self.dp.arm(self.g_arena)
        if cut:
            first, rest = self._split_vars(self.g_vars, self._CUT_G)
            torch.autograd.backward([Dg_logit], [self._const_like(Dg_logit, -1.0 / B)], inputs=[g_cut] + first)
            self._g_rest = (g_cut, G_kl, rest)
        else:
            torch.autograd.backward([Dg_logit, G_kl], [self._const_like(Dg_logit, -1.0 / B), self._const_like(G_kl, self.kl_coeff)],
                                    inputs=list(self.g_vars.values()))
        A.side_join()
        with torch.no_grad():
            D_loss_fake = Dg_logit.detach().mean()
            G_kl_loss = G_kl.detach().reshape(())
            G_loss = -D_loss_fake + self.kl_coeff * G_kl_loss
        return dict(G_loss=G_loss, G_kl_loss=G_kl_loss, D_loss_fake=D_loss_fake, G=G.detach())

    def g_backward_rest(self):
        """Second part of a cut generator backward: the 4x4 bottleneck, dense_2 and the conditioning heads, seeded with the gradient at
        the cut and with dG_loss/dKL (the KL term only reaches the heads)."""
        g_cut, G_kl, rest = self._g_rest
        self._g_rest = None
        g, g_cut.grad = g_cut.grad, None
        torch.autograd.backward([g_cut, G_kl], [g, self._const_like(G_kl, self.kl_coeff)], inputs=rest)
        A.side_join()

    def _refresh_filters(self):
        """Head of a captured graph: one batched regeneration per arena of the cached filter images (kernels.filter_cache_refresh)."""
        K.filter_cache_refresh(self.d_arena.flat)
        K.filter_cache_refresh(self.g_arena.flat)

    def _prepare_ahead(self):
        """The second stream and its workspace lane (sized like the main lane) — allocated OUTSIDE any capture."""
        if getattr(self, '_ahead', None) is None:
            self._ahead = torch.cuda.Stream(device=self.device)
        K.stream_lane(self._ahead, self.device)              # its convolutions get their own scratch

    def _g_forward_ahead(self, feed):
        """_g_forward issued on a second stream, forked from the current one: the G step's generator forward reads only the
        generator's variables and the feed, so it can run BESIDE the critic step (whose own generator pass is a no_grad
        evaluation that updates nothing) instead of after it.  Its ~70 launches are small (B x 4x4..32x32 maps, batch-norm
        reductions) and leave most of the chip idle; next to the critic's large GEMMs they are nearly free.  The caller joins
        with `torch.cuda.current_stream().wait_stream(self._ahead)` before the critic reads G."""
        self._ahead.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._ahead):
            fwd = self._g_forward(feed)
        return fwd

    def _g_body(self, feed, fwd=None, ahead=True):
        if fwd is not None and ahead:                        # issued ahead on the second stream: join, and tell the allocator
            cur = torch.cuda.current_stream()
            cur.wait_stream(self._ahead)
            for t in fwd:
                t.record_stream(cur)
        out = self.g_losses(feed, fwd)
        scale = 1.0
        if self.dp is not None:
            scale = self.dp.allreduce_arena(self.g_arena)
        self.G_optim.apply(grad_scale=scale)
        return out

    def g_step(self, feed):
        with roctx.range('wgancls.g_step'):
            return self._g_step(feed)

    def _g_step(self, feed):
        self.G_optim.prepare(float(feed['learning_rate_g']))
        if self._graphs is not None:
            if not self._graphs['loaded']:
                self._load_static(feed, noise=('ca_noise_g',))
            elif feed.get('ca_noise_g') is None:       # inputs were loaded by d_step on this feed: only the G step's own draw is due
                K.trunc_normal_(self._graphs['static']['ca_noise_g'])
            self._graphs['loaded'] = False
            self._graphs['g'].replay()
            if self.dp is not None:
                self.dp.allreduce_arena(self.g_arena)
                self._graphs['g_upd'].replay()
            K.filter_cache_invalidate(external=False)
            return self._graphs['g_out']
        return self._g_body(feed)

    def dg_step(self, feed):
        """d_step followed by g_step on the same feed (reference trainer.py:97-102).  Under graph replay the two halves
        are ONE graph launch where that is possible (single GPU): a graph launch costs ~0.15 ms of idle GPU on this stack
        (tools/dp_phase_times.py); with data parallelism the critic's Adam segment and the generator half share a graph."""
        g = self._graphs
        if g is None:
            if self.dp is not None and self.dp_cut_eager:
                return self._dg_cut_eager(feed)
            if self._pairing(feed):
                return self._dg_pair_eager(feed)
            return self.d_step(feed), self.g_step(feed)
        self.D_optim.prepare(float(feed['learning_rate_d']))
        self.G_optim.prepare(float(feed['learning_rate_g']))
        self._load_static(feed)
        g['loaded'] = False
        if self.dp is None:
            if g.get('trust') and g.get('epoch') != K.filter_epoch():
                # the graph does not regenerate the critic's filter images at its head (enable_graphs): before the first replay, and
                # whenever someone outside the training step wrote filters since the last one, they are regenerated here
                g['dref'].replay()
                g['epoch'] = K.filter_epoch()
            with roctx.range('wgancls.iteration: d_step + g_step (one hipGraph replay)'):
                g['dg'].replay()
        else:
            # five graph launches, four collectives; each backward is cut once so that the bulk of its gradients is on the wire
            # while the rest of the backward (and, for the critic, the generator's forward) still runs — see enable_graphs
            dA, dB = self._d_cut_ranges()
            gA, gB = self._cut_ranges(self.g_arena, self._CUT_G)
            with roctx.range('d_step segment A: critic losses + backward to Conv_3'):
                g['d_a'].replay()                  # critic losses + backward down to the input of Conv_3
            self.dp.start_allreduce(self.d_arena, extra=g['dg_out'][0]['wd_sums'], ranges=dA)
            with roctx.range('segment B: generator forward + rest of critic backward'):
                g['gf_d_b'].replay()               # generator forward (needs no critic variable) + the rest of the critic backward
            self.dp.start_allreduce(self.d_arena, ranges=dB)
            self.dp.finish_allreduce(self.d_arena)
            with roctx.range('g_step segment A: critic Adam + kt, D(G), backward to the 8x8 boundary'):
                g['dupd_g_a'].replay()             # critic Adam + kt; critic on G; backward down to the generator's 4x4 -> 8x8 boundary
            self.dp.start_allreduce(self.g_arena, ranges=gA)
            with roctx.range('g_step segment B: rest of generator backward'):
                g['g_b'].replay()                  # the rest of the generator backward
            self.dp.start_allreduce(self.g_arena, ranges=gB)
            self.dp.finish_allreduce(self.g_arena)
            with roctx.range('generator Adam'):
                g['g_upd'].replay()
        K.filter_cache_invalidate(external=False)
        self.global_step += 1
        return g['dg_out']

    def _dg_pair_eager(self, feed):
        """d_step then g_step with the generator's two evaluations as one stacked pass (_g_forward_pair), launched eagerly: the order
        of the one-graph iteration (enable_graphs)."""
        self.D_optim.prepare(float(feed['learning_rate_d']))
        self.G_optim.prepare(float(feed['learning_rate_g']))
        with roctx.range('wgancls.iteration: paired generator forward, d_step, g_step'):
            fwd = self._g_forward_pair(feed)
            d_out = self._d_body(feed, have_g=True)
            g_out = self._g_body(feed, fwd, ahead=False)
        self.global_step += 1
        return d_out, g_out

    def _dg_cut_eager(self, feed):
        """The segment sequence of the data-parallel graph schedule (enable_graphs), launched eagerly: the same calls in the same
        order, the exchanges between them.  Selected with `dp_cut_eager` (T2I_DP_CUT_EAGER=1); it is what the CPU gloo test and
        the 2-ranks-on-1-GPU exactness test run without graphs, and a fallback where capture is not available.  (The default
        eager data-parallel step is the bucket-overlap schedule driven by autograd.NOTIFY, dp.py.)"""
        self.D_optim.prepare(float(feed['learning_rate_d']))
        self.G_optim.prepare(float(feed['learning_rate_g']))
        scale = 1.0 / self.dp.world
        dA, dB = self._d_cut_ranges()
        gA, gB = self._cut_ranges(self.g_arena, self._CUT_G)
        self._capturing = True                      # the exchanges are issued here, not by armed hooks
        try:
            d_out = self.d_losses(feed, cut=True)
            self.dp.start_allreduce(self.d_arena, extra=d_out['wd_sums'], ranges=dA)
            fwd = self._g_forward(feed, keep_cut=True)
            self.d_backward_rest()
            self.dp.start_allreduce(self.d_arena, ranges=dB)
            self.dp.finish_allreduce(self.d_arena)
            self._d_update(d_out, scale)
            g_out = self.g_losses(feed, fwd=fwd, cut=True)
            self.dp.start_allreduce(self.g_arena, ranges=gA)
            self.g_backward_rest()
            self.dp.start_allreduce(self.g_arena, ranges=gB)
            self.dp.finish_allreduce(self.g_arena)
            self.G_optim.apply(grad_scale=scale)
        finally:
            self._capturing = False
        self.global_step += 1
        return d_out, g_out

    # ---- hipGraph capture of the two halves of the iteration ---------------------------------------------------------------
    _STATIC_KEYS = ('x', 'x_mismatch', 'cond', 'z', 'epsilon', 'ca_noise_d', 'ca_noise_g')

    def _load_static(self, feed, noise=('ca_noise_d', 'ca_noise_g')):
        """Copy this step's inputs into the captured graphs' static buffers.  Conditioning-augmentation noise the feed does
        not carry is RE-DRAWN in place, in the order the eager step would draw it (`noise`: which of the two draws this
        step makes) — the reference resamples tf.truncated_normal on every run (model.py:119)."""
        for k, buf in self._graphs['static'].items():
            src = feed.get(k)
            if src is None:
                if k not in ('ca_noise_d', 'ca_noise_g'):
                    raise KeyError('feed lacks %r, which the captured graphs read' % k)
                if k in noise:
                    K.trunc_normal_(buf)
            elif src.data_ptr() != buf.data_ptr():
                if k in ('x', 'x_mismatch') and buf._base is not None and src.dtype == buf.dtype and src.is_cuda:
                    K.axpby(src.reshape(buf.shape).contiguous(), 1.0, out=buf)     # a slot of the stacked image buffer (see _d_losses_stacked)
                else:
                    buf.copy_(src.reshape(buf.shape))
        self._graphs['loaded'] = True

    def static_inputs(self):
        """The captured graphs' input buffers (None before enable_graphs).  A data pipeline that writes its batch straight into
        them — or a feed that simply hands them back — makes `_load_static` a no-op: it copies only tensors that live elsewhere."""
        return dict(self._graphs['static']) if self._graphs is not None else None

    def enable_graphs(self, feed):
        """Capture the device work of d_step and g_step into two hipGraphs and replay them from then on: the step's
        ~1000 launches become two graph launches (the host was within 25% of being the bottleneck: 15.9 ms to issue an
        iteration that runs 21 ms).  Shapes are static; per-step scalars (Adam's lr_t, kt) live in device memory.  Call
        after at least one eager iteration with the same shapes (workspace and kernel attributes are then settled).
        With data parallelism each half is cut at the exchange step: [losses + backward] | RCCL all-reduce of the gradient
        arena, issued eagerly | [Adam (+ kt)] — four graph launches and two collectives per iteration, no collective is
        ever captured."""
        static = {k: feed[k].clone() for k in self._STATIC_KEYS if feed.get(k) is not None}
        if self.stack_xhat and getattr(self, '_stk', None) is not None and self._stk['B'] == feed['x'].shape[0]:
            # the stacked critic step reads x and x_mismatch from their slots of [G | x | x_mismatch | x_hat]: the graphs' static inputs
            # ARE those slots, so that a feed written into static_inputs() is never copied again
            B_ = feed['x'].shape[0]
            for k, sl in (('x', slice(B_, 2 * B_)), ('x_mismatch', slice(2 * B_, 3 * B_))):
                slot = self._stk['inp4'][sl]
                slot.copy_(feed[k])
                static[k] = slot
        for k in ('ca_noise_d', 'ca_noise_g'):
            if k not in static:   # re-drawn in place before every replay (_load_static); nothing is drawn here, so the
                static[k] = torch.zeros(feed['cond'].shape[0], self.compressed_embed_dim, device=self.device)   # RNG stream stays the eager one
        if self.dp is None and _OVERLAP_G_FORWARD:
            self._prepare_ahead()
        torch.cuda.synchronize(self.device)
        gd, gg = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        if self.dp is None:
            from ...graphs import capture_mode
            _plain_mode = capture_mode('global')     # 'thread_local' once any process group (and its watchdog thread) exists in this process
            # every graph starts with ONE batched regeneration of the cached filter images (a graph contains every transform
            # it depends on; filled lazily they are ~60 small launches)
            with torch.cuda.graph(gd, capture_error_mode=_plain_mode):
                self._refresh_filters()
                d_out = self._d_body(static)
            with torch.cuda.graph(gg, pool=gd.pool(), capture_error_mode=_plain_mode):
                self._refresh_filters()
                g_out = self._g_body(static)
            trust = _TRUST_IMAGES and K.filter_cache_enabled()        # nothing to trust (and nothing to capture below) without the cache
            gref = None
            if trust:
                gref = torch.cuda.CUDAGraph()            # the critic's filter images alone (dg_step: after an outside write); captured,
                with torch.cuda.graph(gref, pool=gd.pool(), capture_error_mode=_plain_mode):   # because a capture regenerates every image the cache holds of the range
                    K.filter_cache_refresh(self.d_arena.flat)
            gdg = torch.cuda.CUDAGraph()                 # both halves in one launch, same outputs' addresses not needed:
            with torch.cuda.graph(gdg, pool=gd.pool(), capture_error_mode=_plain_mode):  # dg_step returns this capture's own output tensors
                if trust:
                    # the critic's images were regenerated behind its Adam step by the previous replay (_d_update: refresh=True) —
                    # by whichever step function ran last, in fact: all of them leave the critic's images current — so only the
                    # generator's are due here; dg_step vouches for the critic's (t2i_filter_cache_assume)
                    K.filter_cache_assume(self.d_arena.flat)
                    K.filter_cache_refresh(self.g_arena.flat)
                else:
                    self._refresh_filters()
                if self._pairing(static):
                    fwd = self._g_forward_pair(static)       # both generator evaluations as one stacked pass of 2B rows
                    d_out2 = self._d_body(static, have_g=True)
                    g_out2 = self._g_body(static, fwd, ahead=False)
                else:
                    ahead = self._g_forward_ahead(static) if _OVERLAP_G_FORWARD else None   # beside the critic step, not after it
                    d_out2 = self._d_body(static)
                    g_out2 = self._g_body(static, ahead)
            self._graphs = {'d': gd, 'g': gg, 'dg': gdg, 'd_out': d_out, 'g_out': g_out, 'dg_out': (d_out2, g_out2),
                            'static': static, 'loaded': False, 'trust': trust, 'epoch': None, 'dref': gref}
            return
        # thread-local capture mode: the process group's watchdog thread polls events while we capture
        gdu, ggu = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        scale = 1.0 / self.dp.world
        self._capturing = True
        try:
            with torch.cuda.graph(gd, capture_error_mode=_CAPTURE_MODE):
                d_out = self.d_losses(static)
            with torch.cuda.graph(gdu, pool=gd.pool(), capture_error_mode=_CAPTURE_MODE):
                self._d_update(d_out, scale)
            with torch.cuda.graph(gg, pool=gd.pool(), capture_error_mode=_CAPTURE_MODE):
                g_out = self.g_losses(static)
            with torch.cuda.graph(ggu, pool=gd.pool(), capture_error_mode=_CAPTURE_MODE):
                self.G_optim.apply(grad_scale=scale)
            # dg_step (the trainer's iteration): each backward is cut once.  [critic losses + first part of its backward] |
            # exchange of that part starts | [generator forward + rest of the critic backward] | exchange of the rest, wait |
            # [critic Adam + kt; critic on G(z); first part of the generator step's backward] | exchange starts | [rest] |
            # exchange, wait | [generator Adam].  The autograd graph recorded in one capture is consumed by a later one (all
            # captures share one private pool and are replayed in capture order).
            gda, ggfdb, gduga, ggb = (torch.cuda.CUDAGraph() for _ in range(4))
            with torch.cuda.graph(gda, pool=gd.pool(), capture_error_mode=_CAPTURE_MODE):
                d_out2 = self.d_losses(static, cut=True)
            with torch.cuda.graph(ggfdb, pool=gd.pool(), capture_error_mode=_CAPTURE_MODE):
                fwd = self._g_forward(static, keep_cut=True)
                self.d_backward_rest()
            with torch.cuda.graph(gduga, pool=gd.pool(), capture_error_mode=_CAPTURE_MODE):
                self._d_update(d_out2, scale)
                g_out2 = self.g_losses(static, fwd=fwd, cut=True)
            with torch.cuda.graph(ggb, pool=gd.pool(), capture_error_mode=_CAPTURE_MODE):
                self.g_backward_rest()
            del fwd
        finally:
            self._capturing = False
        self._graphs = {'d': gd, 'g': gg, 'd_upd': gdu, 'g_upd': ggu, 'd_a': gda, 'gf_d_b': ggfdb, 'dupd_g_a': gduga, 'g_b': ggb,
                        'd_out': d_out, 'g_out': g_out, 'dg_out': (d_out2, g_out2), 'static': static, 'loaded': False}
        if self.stack_xhat:
            self.dp_schedule = ('5 graphs + 3 eager exchanges/iteration: the stacked critic step (4B rows, one filter-gradient launch per layer '
                                'inside the double backward) finishes its 116 MB arena at once, which leaves while the generator forward '
                                'runs; the generator\'s backward is cut at the 4x4->8x8 boundary: 57 of 91 MB leave before the 4x4 layers')
        else:
            self.dp_schedule = ('5 graphs + 4 eager exchanges/iteration: both backward passes cut once (critic at the input of Conv_3: '
                                '105 of 116 MB leave while the generator forward and the rest of the backward run; generator at the '
                                '4x4->8x8 boundary: 57 of 91 MB leave before the 4x4 layers)')

    def sampler(self, z_sample, cond_sample):
        """eval-mode generator on fixed samples (reference model.py:57)"""
        with torch.no_grad():
            img, _, _ = self.generator(z_sample, cond_sample, reuse=True, is_training=False)
        return img

    # ------------------------------------------------------------------------------------------------------------------
    def generate_conditionals(self, embeddings):
        """Conditioning augmentation statistics (reference model.py:108-115): two lrelu-activated 1024->128 dense
        layers, `g_net/dense` (mu) and `g_net/dense_1` (log sigma)."""
        act = lrelu_act(0.2)
        flat = embeddings.reshape(embeddings.shape[0], -1)
        with K.f32_outputs():      # [B, 128] each, consumed by the fp32 conditioning-augmentation kernel (bf16 storage would round them for nothing)
            return fc(flat, self.compressed_embed_dim, act=act), fc(flat, self.compressed_embed_dim, act=act)

    def sample_normal_conditional(self, mean, log_sigma, cond_noise=True):
        """c = mu + exp(log sigma) * eps, eps ~ truncated N(0,1) (reference model.py:117-122).  The draw comes from the
        feed (`ca_noise_d` / `ca_noise_g`) when given so that runs are reproducible; [B,128] scalar math stays in torch."""
        self._kl = None
        if not cond_noise:
            return mean
        eps = getattr(self, '_noise', None)
        if eps is None or eps.shape != mean.shape:
            eps = torch.empty_like(mean)
            K.trunc_normal_(eps)
        code, self._kl = A.CaSampleKlFn.apply(mean, log_sigma, eps)     # one launch; the KL term rides along
        return code

    def kl_std_normal_loss(self, mean, log_sigma):
        """KL(N(mu, sigma) || N(0, 1)) averaged over batch and features (reference model.py:124-127)."""
        return torch.mean(0.5 * (torch.exp(2.0 * log_sigma) + mean * mean - 1.0) - log_sigma)

    # -- critic: 4 stride-2 convs, a bottleneck residual, text conditioning, 3 head convs (reference model.py:129-161) --
    def discriminator(self, inputs, embed, reuse=False):
        nf, act, fmt = self.df_dim, lrelu_act(0.2), NCHW
        h = to_nchw(inputs)
        with S.variable_scope('d_net', reuse=reuse):
            for mult in (1, 2, 4):                                             # d_net/Conv, Conv_1, Conv_2
                h = conv2d(h, nf * mult, ks=(4, 4), s=(2, 2), act=act, df=fmt)
            if self._keep_cut:                                                 # where the data-parallel schedule cuts the backward (_CUT_D)
                self._d_cut = h
            trunk = conv2d(h, nf * 8, ks=(4, 4), s=(2, 2), df=fmt)             # Conv_3, linear
            trunk, trunk_ = fork(trunk, df=fmt)                                # (two consumers; a stacked pass sums their gradients in one launch)
            r = conv2d(trunk_, nf * 2, ks=(1, 1), s=(1, 1), padding='valid', act=act, df=fmt)   # Conv_4
            r = conv2d(r, nf * 4, ks=(3, 3), s=(1, 1), act=act, df=fmt)        # Conv_5
            r = conv2d(r, nf * 8, ks=(3, 3), s=(1, 1), df=fmt)                 # Conv_6
            joined = add(trunk, r, act=act, df=fmt)
            text = fc(embed, self.compressed_embed_dim, act=act)               # d_net/dense
            h = concat_tile(joined, text, df=fmt)                              # [B,4,4,8nf+128]
            h = conv2d(h, nf * 8, ks=(3, 3), s=(1, 1), padding='same', act=act, df=fmt)    # Conv_7
            h = conv2d(h, nf * 8, ks=(1, 1), s=(1, 1), padding='valid', act=act, df=fmt)   # Conv_8
            return conv2d(h, 1, ks=(4, 4), s=(4, 4), padding='valid', df=fmt)              # Conv_9 -> [B,1,1,1]

    # -- generator (reference model.py:163-225) ---------------------------------------------------------------------------
    def _g_bottleneck(self, x, mid, out, train, fmt):
        """1x1 -> BN/ReLU -> 3x3 -> BN/ReLU -> 3x3 -> BN, added to x, ReLU (model.py:184-191 and :200-207)."""
        # stats=train: the conv epilogue hands the batch norm its column sums (no separate statistics pass)
        r = batch_norm(conv2d(x, mid, ks=(1, 1), s=(1, 1), padding='valid', df=fmt, stats=train), train=train, act=relu, df=fmt)
        r = batch_norm(conv2d(r, mid, ks=(3, 3), s=(1, 1), df=fmt, stats=train), train=train, act=relu, df=fmt)
        r = batch_norm(conv2d(r, out, ks=(3, 3), s=(1, 1), df=fmt, stats=train), train=train, act=None, df=fmt)
        return add(x, r, act=relu, df=fmt)

    def _g_upsample(self, x, nf, train, fmt, act):
        """k4s2 transposed conv -> 3x3 conv -> BN(+act) (model.py:194-196, 210-216)."""
        u = conv2d(conv2d_transpose(x, nf, ks=(4, 4), s=(2, 2), df=fmt), nf, ks=(3, 3), s=(1, 1), df=fmt, stats=train)
        return batch_norm(u, train=train, act=act, df=fmt)

    def generator(self, z, embed, reuse=False, is_training=True, df=NCHW, cond_noise=True, pair=None):
        """pair = (noise of the leading evaluation, noise of the second): both evaluations of a D + G iteration as one stacked pass
        (_g_forward_pair); the returned image is then a stacked.Stacked."""
        # config 3's compliant mode: the generator's layers in their own arithmetic / storage (self.net_math['g_net'])
        with K.math_scope(*self.net_math.get('g_net', (None, None))):
            return self._generator(z, embed, reuse, is_training, df, cond_noise, pair)

    def _pair_code(self, z, mean, log_sigma, noise_a, noise_b):
        """The generator's input code for two evaluations that differ in their conditioning noise only: [z | c_a] with the autograd
        graph (and the KL term), [z | c_b] without."""
        code_a, self._kl = A.CaSampleKlFn.apply(mean, log_sigma, noise_a)
        with torch.no_grad():
            code_b, _ = K.ca_kl_fwd(mean.detach(), log_sigma.detach(), noise_b)
            hat = torch.cat([z, code_b], 1)
        return ST.Stacked(torch.cat([z, code_a], 1), hat)

    def _generator(self, z, embed, reuse, is_training, df, cond_noise, pair=None):
        nf, grid = self.gf_dim, self.output_size // 16
        with S.variable_scope('g_net', reuse=reuse):
            mean, log_sigma = self.generate_conditionals(embed)
            if pair is not None:
                code = self._pair_code(z, mean, log_sigma, *pair)
            else:
                code = torch.cat([z, self.sample_normal_conditional(mean, log_sigma, cond_noise)], 1)
            h = batch_norm(fc(code, nf * 8 * grid * grid), train=is_training, df=df)      # dense_2 + rank-2 BatchNorm
            h = reshape_to_map(h, nf * 8, grid, grid, df)                                 # [B,4,4,8nf]
            h = self._g_bottleneck(h, nf * 2, nf * 8, is_training, df)
            if self._keep_cut:                                                            # the 4x4 -> 8x8 boundary (_CUT_G)
                self._g_cut = h
            h = self._g_upsample(h, nf * 4, is_training, df, act=None)                    # 8x8
            h = self._g_bottleneck(h, nf, nf * 4, is_training, df)
            h = self._g_upsample(h, nf * 2, is_training, df, act=relu)                    # 16x16
            h = self._g_upsample(h, nf, is_training, df, act=relu)                        # 32x32
            rgb = conv2d_transpose(h, self.image_dims[-1], ks=(4, 4), s=(2, 2), df=df)    # 64x64
            img = conv2d(rgb, self.image_dims[-1], ks=(3, 3), s=(1, 1), act=tanh, df=df)  # tanh fused in the epilogue
            return (to_nhwc(img) if df == NCHW else img), mean, log_sigma
