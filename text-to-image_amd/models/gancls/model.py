"""GanCls — the reference's GAN-CLS variant (reference models/gancls/model.py:5-192) on libt2i_hip.so kernels.

Same layer stack as wgancls' generator minus the conditioning augmentation, a batch-normalised discriminator, NHWC,
`tf.layers` variable names, N(0,0.02) kernels and gamma ~ N(1,0.02).  Because the discriminator has batch norm, its three
passes (fake / match / mismatch, model.py:48-51) keep SEPARATE batch statistics and cannot be merged into one batch."""
import torch

from ... import kernels as K
from ... import optim
from ... import scope as S
from ...utils.ops import add, batch_norm, concat_tile, lrelu_act, relu, tanh
from ...utils.tf_layers import conv2d, conv2d_transpose, dense


class GanCls(object):
    def __init__(self, cfg, build_model=True, device=None, seed=0, dp=None):
        self.name = 'GANL_CLS'
        m, t = cfg.MODEL, cfg.TRAIN
        self.cfg = cfg
        self.batch_size, self.sample_num = t.BATCH_SIZE, t.SAMPLE_NUM
        self.output_size = m.OUTPUT_SIZE
        self.z_dim, self.embed_dim, self.compressed_embed_dim = m.Z_DIM, m.EMBED_DIM, m.COMPRESSED_EMBED_DIM
        self.gf_dim, self.df_dim = m.GF_DIM, m.DF_DIM
        self.image_dims = [m.IMAGE_SHAPE.H, m.IMAGE_SHAPE.W, m.IMAGE_SHAPE.D]
        if self.output_size != 64:
            raise ValueError('the reference tiles the text code over a fixed 4x4 map (model.py:95): OUTPUT_SIZE must be 64')
        self.w_init = S.normal_init(0.02)                          # tf.random_normal_initializer(stddev=0.02)
        self.batch_norm_init = {'gamma': S.normal_init(0.02, 1.0)}  # gamma ~ N(1, 0.02)
        self.store = S.set_default_store(S.VariableStore(device=device, seed=seed))
        self.device = self.store.device
        self.dp = dp
        if build_model:
            self.build_model()

    def build_model(self):
        """Variable creation by a launch-free dry pass (the placeholders of model.py:38-45 become feed-dict keys)."""
        B, dev = self.batch_size, self.device
        with K.dry_run(), torch.no_grad():
            z = torch.empty(B, self.z_dim, device=dev)
            phi = torch.empty(B, self.embed_dim, device=dev)
            G = self.generator(z, phi, reuse=False)
            self.discriminator(G, phi, reuse=False)
        self.d_vars = S.trainable_variables('d_net')
        self.g_vars = S.trainable_variables('g_net')
        self.d_arena = optim.Arena(self.d_vars)
        self.g_arena = optim.Arena(self.g_vars)
        # gradients are accumulated by the kernels' epilogues straight into the optimizer arenas (autograd.SINKS); with
        # data parallelism the bucket overlap follows autograd.NOTIFY instead of AccumulateGrad hooks (dp.py)
        self.d_arena.enable_sinks()
        self.g_arena.enable_sinks()

    def sampler(self, z_sample, phi_sample):
        with torch.no_grad():
            return self.generator(z_sample, phi_sample, is_training=False, reuse=True)

    def discriminator(self, inputs, embed, is_training=True, reuse=False, _prob=True, groups=1):
        """-> (sigmoid(logits), logits), logits [B,1,1,1]  (reference model.py:54-109).  _prob=False (the trainer): the first element is
        None — the loss head kernel (kernels.sigmoid_ce_head) returns the probabilities with the losses, one launch for all passes.
        groups > 1 (the trainer): `inputs` / `embed` hold that many passes of the reference graph stacked along the batch axis (fake | match |
        mismatch, model.py:48-51).  Convolutions, activations and the text projection are per-sample, so they run once on the stacked batch;
        every batch norm keeps SEPARATE statistics per pass (ops.batch_norm(groups=...)), moving averages move once per pass in stacking
        order — the same values as `groups` sequential calls, in a third of the launches."""
        nf, act, bn_init, s16 = self.df_dim, lrelu_act(0.2), self.batch_norm_init, self.output_size / 16
        with S.variable_scope('d_net', reuse=reuse):
            h = conv2d(inputs, nf, (4, 4), (2, 2), 'same', activation=act, kernel_initializer=self.w_init)
            for mult, a in ((2, act), (4, act), (8, None)):        # conv2d_1..3 + BatchNorm..BatchNorm_2
                h = conv2d(h, nf * mult, (4, 4), (2, 2), 'same', kernel_initializer=self.w_init)
                h = batch_norm(h, train=is_training, init=bn_init, act=a, groups=groups)
            trunk = h
            r = conv2d(trunk, nf * 2, (1, 1), (1, 1), 'valid', kernel_initializer=self.w_init)
            r = batch_norm(r, train=is_training, init=bn_init, act=act, groups=groups)
            r = conv2d(r, nf * 2, (3, 3), (1, 1), 'same', kernel_initializer=self.w_init)
            r = batch_norm(r, train=is_training, init=bn_init, act=act, groups=groups)
            r = conv2d(r, nf * 8, (3, 3), (1, 1), 'same', kernel_initializer=self.w_init)
            r = batch_norm(r, train=is_training, init=bn_init, act=None, groups=groups)
            joined = add(trunk, r, act=act)
            text = dense(embed, self.compressed_embed_dim, activation=act)            # glorot-uniform (tf.layers default)
            h = concat_tile(joined, text)
            h = conv2d(h, nf * 8, (1, 1), (1, 1), 'valid', kernel_initializer=self.w_init)
            h = batch_norm(h, train=is_training, init=bn_init, act=act, groups=groups)
            logits = conv2d(h, 1, (s16, s16), (s16, s16), 'valid', kernel_initializer=self.w_init)
            return (torch.sigmoid(logits) if _prob else None), logits

    def _bottleneck(self, x, mid, out, train):
        bn_init = self.batch_norm_init
        r = batch_norm(conv2d(x, mid, (1, 1), (1, 1), 'valid', kernel_initializer=self.w_init), train=train, init=bn_init, act=relu)
        r = batch_norm(conv2d(r, mid, (3, 3), (1, 1), 'same', kernel_initializer=self.w_init), train=train, init=bn_init, act=relu)
        r = batch_norm(conv2d(r, out, (3, 3), (1, 1), 'same', kernel_initializer=self.w_init), train=train, init=bn_init, act=None)
        return add(x, r, act=relu)

    def _upsample(self, x, nf, train, act):
        u = conv2d_transpose(x, nf, (4, 4), (2, 2), 'same', kernel_initializer=self.w_init)
        u = conv2d(u, nf, (3, 3), (1, 1), 'same', kernel_initializer=self.w_init)
        return batch_norm(u, train=train, init=self.batch_norm_init, act=act)

    def generator(self, z, embed, is_training=True, reuse=False):
        """-> image NHWC in [-1,1]  (reference model.py:111-192)"""
        nf = self.gf_dim
        with S.variable_scope('g_net', reuse=reuse):
            code = torch.cat([z, dense(embed, self.compressed_embed_dim)], 1)
            h = dense(code, nf * 8 * 16, kernel_initializer=self.w_init)
            h = batch_norm(h, train=is_training, init=self.batch_norm_init)
            h = h.reshape(-1, 4, 4, nf * 8)                       # NHWC reshape: free
            h = self._bottleneck(h, nf * 2, nf * 8, is_training)
            h = self._upsample(h, nf * 4, is_training, act=None)
            h = self._bottleneck(h, nf, nf * 4, is_training)
            h = self._upsample(h, nf * 2, is_training, act=relu)
            h = self._upsample(h, nf, is_training, act=relu)
            rgb = conv2d_transpose(h, self.image_dims[-1], (4, 4), (2, 2), 'same', kernel_initializer=self.w_init)
            return conv2d(rgb, self.image_dims[-1], (3, 3), (1, 1), 'same', activation=tanh, kernel_initializer=self.w_init)
