"""StackGAN Stage-I — reference models/stackgan/stageI/model.py:5-171 on libt2i_hip.so kernels.

Same class name, constructor and methods as the reference.  NHWC throughout; `utils.ops` layers (tf.contrib.layers
variable names `Conv_3/weights`, `BatchNorm_2/gamma`) plus `tf.layers.dense` (`dense_1/kernel`); N(0, 0.02) kernels,
gamma ~ N(1, 0.02).  Zero new kernels: every layer is one of the hot path's conv / conv^T / dense / batch-norm Functions.
The discriminator is batch-normalised, so its three passes (fake / match / mismatch, model.py:47-51) keep separate batch
statistics and are not merged into one batch."""
import torch

from .... import kernels as K
from .... import optim
from .... import scope as S
from ....utils.ops import add, batch_norm, concat_tile, conv2d, conv2d_transpose, lrelu_act, relu, tanh
from ....utils.tf_layers import dense


class ConditionalGan(object):
    def __init__(self, cfg, build_model=True, device=None, seed=0, dp=None, store=None):
        self.name = 'ConditionalGAN/StageI'
        self.g_scope, self.d_scope = 'g_net', 'd_net'
        m, t = cfg.MODEL, cfg.TRAIN
        self.cfg = cfg
        self.batch_size, self.sample_num = t.BATCH_SIZE, t.SAMPLE_NUM
        self.output_size = m.OUTPUT_SIZE
        self.z_dim, self.embed_dim, self.compressed_embed_dim = m.Z_DIM, m.EMBED_DIM, m.COMPRESSED_EMBED_DIM
        self.gf_dim, self.df_dim = m.GF_DIM, m.DF_DIM
        self.image_dims = [m.IMAGE_SHAPE.H, m.IMAGE_SHAPE.W, m.IMAGE_SHAPE.D]
        if self.output_size != 64:
            raise ValueError('the reference tiles the text code over a fixed 4x4 map (stageI/model.py:111): OUTPUT_SIZE must be 64')
        self.w_init = S.normal_init(0.02)                           # tf.random_normal_initializer(stddev=0.02)
        self.batch_norm_init = {'gamma': S.normal_init(0.02, 1.0)}   # gamma ~ N(1, 0.02)
        self.store = S.set_default_store(store or S.VariableStore(device=device, seed=seed))
        self.device = self.store.device
        self.dp = dp
        # per-network arithmetic (kernels.math_scope): {'g_net' | 'd_net': (math, storage, backward math)} — that network's layers are created under it
        # (kernels.FWD_F32_BWD_BF16: forward GEMMs in fp32, input- and filter-gradient GEMMs in bf16); {} = the process-wide setting
        self.net_math = {}
        if build_model:
            self.build_model()

    def build_model(self):
        """Variable creation by a launch-free dry pass (the placeholders of model.py:38-45 become feed-dict keys)."""
        B, dev = self.batch_size, self.device
        with K.dry_run(), torch.no_grad():
            z = torch.empty(B, self.z_dim, device=dev)
            phi = torch.empty(B, self.embed_dim, device=dev)
            G, _, _ = self.generator(z, phi, reuse=False)
            self.discriminator(G, phi, reuse=False)
        self.d_vars = S.trainable_variables('d_net')
        self.g_vars = S.trainable_variables('g_net')
        self.d_arena = optim.Arena(self.d_vars)
        self.g_arena = optim.Arena(self.g_vars)
        self.d_arena.enable_sinks()
        self.g_arena.enable_sinks()

    def sampler(self, z_sample, embed_sample):
        with torch.no_grad():
            return self.generator(z_sample, embed_sample, is_training=False, reuse=True)[0]

    # ---- conditioning augmentation (model.py:59-75) ---------------------------------------------------------------------
    def generate_conditionals(self, embeddings):
        act = lrelu_act(0.2)
        embeddings = embeddings.reshape(embeddings.shape[0], -1)
        with K.f32_outputs():          # [B, 128] statistics of the conditioning augmentation: fp32 in every storage mode
            mean = dense(embeddings, self.compressed_embed_dim, activation=act, kernel_initializer=self.w_init)
            log_sigma = dense(embeddings, self.compressed_embed_dim, activation=act, kernel_initializer=self.w_init)
        return mean, log_sigma

    def sample_normal_conditional(self, mean, log_sigma, cond_noise=True, noise=None):
        """noise: the truncated-normal draw (a feed key here so that runs are reproducible); None -> drawn on the device."""
        if cond_noise:
            if noise is None:
                noise = K.trunc_normal_(torch.empty_like(mean))
            return mean + torch.exp(log_sigma) * noise
        return mean

    # ---- networks ------------------------------------------------------------------------------------------------------
    def discriminator(self, inputs, embed, is_training=True, reuse=False, _prob=True, groups=1):
        """-> (sigmoid(logits), logits), logits [B,1,1,1]  (model.py:77-121)"""
        nf, act, bn_init, s16 = self.df_dim, lrelu_act(0.2), self.batch_norm_init, self.output_size // 16
        with K.math_scope(*self.net_math.get('d_net', (None, None))), S.variable_scope('d_net', reuse=reuse):
            h = conv2d(inputs, nf, ks=(4, 4), s=(2, 2), act=act, init=self.w_init)
            for mult, a in ((2, act), (4, act), (8, None)):
                h = conv2d(h, nf * mult, ks=(4, 4), s=(2, 2), init=self.w_init)
                h = batch_norm(h, train=is_training, init=bn_init, act=a, groups=groups)
            trunk = h
            r = conv2d(trunk, nf * 2, ks=(1, 1), s=(1, 1), padding='valid', init=self.w_init)
            r = batch_norm(r, train=is_training, init=bn_init, act=act, groups=groups)
            r = conv2d(r, nf * 2, ks=(3, 3), s=(1, 1), init=self.w_init)
            r = batch_norm(r, train=is_training, init=bn_init, act=act, groups=groups)
            r = conv2d(r, nf * 8, ks=(3, 3), s=(1, 1), init=self.w_init)
            r = batch_norm(r, train=is_training, init=bn_init, groups=groups)
            joined = add(trunk, r, act=act)
            text = dense(embed, self.compressed_embed_dim, activation=act)          # tf.layers default: glorot-uniform
            h = concat_tile(joined, text)
            h = conv2d(h, nf * 8, ks=(1, 1), s=(1, 1), padding='valid', init=self.w_init)
            h = batch_norm(h, train=is_training, init=bn_init, act=act, groups=groups)
            logits = conv2d(h, 1, ks=(s16, s16), s=(s16, s16), padding='valid', init=self.w_init)
            # _prob=False / groups: see models/gancls/model.py (the trainer stacks the critic passes of one sess.run along the batch axis)
            return (torch.sigmoid(logits) if _prob else None), logits

    def _bottleneck(self, x, mid, out, train):
        bn_init = self.batch_norm_init
        r = batch_norm(conv2d(x, mid, ks=(1, 1), s=(1, 1), padding='valid', init=self.w_init), train=train, init=bn_init, act=relu)
        r = batch_norm(conv2d(r, mid, ks=(3, 3), s=(1, 1), init=self.w_init), train=train, init=bn_init, act=relu)
        r = batch_norm(conv2d(r, out, ks=(3, 3), s=(1, 1), init=self.w_init), train=train, init=bn_init)
        return add(x, r, act=relu)

    def _upsample(self, x, nf, train, act):
        u = conv2d_transpose(x, nf, ks=(4, 4), s=(2, 2), init=self.w_init)
        u = conv2d(u, nf, ks=(3, 3), s=(1, 1), init=self.w_init)
        return batch_norm(u, train=train, init=self.batch_norm_init, act=act)

    def generator(self, z, embed, is_training=True, reuse=False, cond_noise=True, noise=None):
        """-> (image NHWC in [-1,1], mean, log_sigma)  (model.py:123-171)"""
        nf, s16 = self.gf_dim, self.output_size // 16
        with K.math_scope(*self.net_math.get('g_net', (None, None))), S.variable_scope('g_net', reuse=reuse):
            mean, log_sigma = self.generate_conditionals(embed)
            code = self.sample_normal_conditional(mean, log_sigma, cond_noise, noise)
            h = dense(torch.cat([z, code], 1), nf * 8 * s16 * s16, kernel_initializer=self.w_init)
            h = batch_norm(h, train=is_training, init=self.batch_norm_init)
            h = h.reshape(-1, s16, s16, nf * 8)                     # NHWC reshape: free
            h = self._bottleneck(h, nf * 2, nf * 8, is_training)
            h = self._upsample(h, nf * 4, is_training, act=None)
            h = self._bottleneck(h, nf, nf * 4, is_training)
            h = self._upsample(h, nf * 2, is_training, act=relu)
            h = self._upsample(h, nf, is_training, act=relu)
            rgb = conv2d_transpose(h, self.image_dims[-1], ks=(4, 4), s=(2, 2), init=self.w_init)
            return conv2d(rgb, self.image_dims[-1], ks=(3, 3), s=(1, 1), act=tanh, init=self.w_init), mean, log_sigma
