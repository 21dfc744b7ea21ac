"""StackGAN Stage-II — reference models/stackgan/stageII/model.py:7-201 on libt2i_hip.so kernels.

256x256 generator conditioned on the Stage-I 64x64 image and the text embedding; 256x256 discriminator.  Reference
quirks kept on purpose:
  * the Stage-I generator runs inside the Stage-II graph in TRAINING mode (stageII/model.py:50 uses the default
    is_training=True): frozen weights (not in any var_list), batch statistics, and — because both optimizers sit under
    UPDATE_OPS — its moving averages keep moving during Stage-II training;
  * the discriminator's residual join is `tf.add(net, net)` (stageII/model.py:117): the branch doubled, the trunk dropped;
  * `generator_residual_layer` uses 4x4 stride-1 SAME convolutions (asymmetric padding 1 top/left, 2 bottom/right);
  * layers without an explicit `init` take utils/ops.py's He initializer, the others N(0, 0.02).
No new kernels: the k4 s1 geometry is the same implicit-GEMM kernel with pad_t = pad_l = 1 (parity fixture `k4s1_same`)."""
import torch

from .... import kernels as K
from .... import optim
from .... import scope as S
from ....utils.ops import add, batch_norm, concat_tile, conv2d, conv2d_transpose, lrelu_act, relu, tanh
from ....utils.tf_layers import dense


class ConditionalGan(object):
    def __init__(self, stagei, cfg, build_model=True, dp=None):
        self.name = 'ConditionalGAN/StageII'
        self.g_scope, self.d_scope = 'stageII_g_net', 'stageII_d_net'
        self.stagei = stagei
        m, t = cfg.MODEL, cfg.TRAIN
        self.cfg = cfg
        self.batch_size, self.sample_num = t.BATCH_SIZE, t.SAMPLE_NUM
        self.output_size = m.OUTPUT_SIZE
        self.z_dim, self.embed_dim, self.compressed_embed_dim = m.Z_DIM, m.EMBED_DIM, m.COMPRESSED_EMBED_DIM
        self.gf_dim, self.df_dim = m.GF_DIM, m.DF_DIM
        self.image_dims = [m.IMAGE_SHAPE.H, m.IMAGE_SHAPE.W, m.IMAGE_SHAPE.D]
        if self.output_size != 256:
            raise ValueError('the Stage-II generator always produces 256x256 (64 -> 16 -> four x2 upsamplings): OUTPUT_SIZE must be 256')
        self.w_init = S.normal_init(0.02)
        self.batch_norm_init = {'gamma': S.normal_init(0.02, 1.0)}
        self.store = S.set_default_store(stagei.store)               # one variable store, like one TF graph
        self.device = self.store.device
        self.dp = dp
        # per-network arithmetic (kernels.math_scope): {'g_net' | 'd_net': (math, storage, backward math)} — that network's layers are created under it
        # (kernels.FWD_F32_BWD_BF16: forward GEMMs in fp32, input- and filter-gradient GEMMs in bf16); {} = the process-wide setting
        self.net_math = {}
        if build_model:
            self.build_model()

    def build_model(self):
        """stageII/model.py:39-63: Stage-I generator variables first (`g_net/...`), then stageII_g_net, stageII_d_net."""
        B, dev = self.batch_size, self.device
        with K.dry_run(), torch.no_grad():
            z = torch.empty(B, self.z_dim, device=dev)
            phi = torch.empty(B, self.embed_dim, device=dev)
            if 'g_net/dense/kernel' not in self.store.vars:          # Stage-I built with build_model=False
                img64, _, _ = self.stagei.generator(z, phi, reuse=False)
            else:
                img64, _, _ = self.stagei.generator(z, phi, reuse=True)
            G, _, _ = self.generator(img64, phi, reuse=False)
            self.discriminator(G, phi, reuse=False)
        self.d_vars = S.trainable_variables('stageII_d_net')
        self.g_vars = S.trainable_variables('stageII_g_net')
        self.d_arena = optim.Arena(self.d_vars)
        self.g_arena = optim.Arena(self.g_vars)
        self.d_arena.enable_sinks()
        self.g_arena.enable_sinks()

    def sampler(self, z_sample, embed_sample):
        with torch.no_grad():
            img64, _, _ = self.stagei.generator(z_sample, embed_sample, is_training=False, reuse=True)
            return self.generator(img64, embed_sample, is_training=False, reuse=True)[0]

    def generate_conditionals(self, embeddings):
        act = lrelu_act(0.2)
        embeddings = embeddings.reshape(embeddings.shape[0], -1)
        with K.f32_outputs():          # [B, 128] statistics of the conditioning augmentation: fp32 in every storage mode
            mean = dense(embeddings, self.compressed_embed_dim, activation=act, kernel_initializer=self.w_init)
            log_sigma = dense(embeddings, self.compressed_embed_dim, activation=act, kernel_initializer=self.w_init)
        return mean, log_sigma

    def sample_normal_conditional(self, mean, log_sigma, cond_noise, noise=None):
        if cond_noise:
            if noise is None:
                noise = K.trunc_normal_(torch.empty_like(mean))
            return mean + torch.exp(log_sigma) * noise
        return mean

    # ---- discriminator (stageII/model.py:78-133) --------------------------------------------------------------------------
    def discriminator(self, inputs, embed, is_training=True, reuse=False, _prob=True, groups=1):
        nf, act, bn_init, s16 = self.df_dim, lrelu_act(0.2), self.batch_norm_init, self.output_size // 64
        with K.math_scope(*self.net_math.get('d_net', (None, None))), S.variable_scope('stageII_d_net', reuse=reuse):
            h = conv2d(inputs, nf, ks=(4, 4), s=(2, 2), act=act, init=self.w_init)
            for mult in (2, 4, 8, 16, 32):
                h = conv2d(h, nf * mult, ks=(4, 4), s=(2, 2), init=self.w_init)
                h = batch_norm(h, train=is_training, init=bn_init, act=act, groups=groups)
            h = conv2d(h, nf * 16, ks=(4, 4), s=(1, 1), init=self.w_init)
            h = batch_norm(h, train=is_training, init=bn_init, act=act, groups=groups)
            h = conv2d(h, nf * 8, ks=(4, 4), s=(1, 1), init=self.w_init)
            h7 = batch_norm(h, train=is_training, init=bn_init, groups=groups)
            r = conv2d(h7, nf * 2, ks=(1, 1), s=(1, 1), init=self.w_init)
            r = batch_norm(r, train=is_training, init=bn_init, act=act, groups=groups)
            r = conv2d(r, nf * 2, ks=(3, 3), s=(1, 1), init=self.w_init)
            r = batch_norm(r, train=is_training, init=bn_init, act=act, groups=groups)
            r = conv2d(r, nf * 8, ks=(3, 3), s=(1, 1), init=self.w_init)
            r = batch_norm(r, train=is_training, init=bn_init, groups=groups)
            h8 = add(r, r, act=act)                                   # tf.add(net, net), not tf.add(net_h7, net)
            text = dense(embed, self.compressed_embed_dim, activation=act)
            h = concat_tile(h8, text)
            h = conv2d(h, nf * 8, ks=(1, 1), s=(1, 1), init=self.w_init)
            h = batch_norm(h, train=is_training, init=bn_init, act=act, groups=groups)
            logits = conv2d(h, 1, ks=(s16, s16), s=(s16, s16), init=self.w_init)
            # _prob=False / groups: see models/gancls/model.py (the trainer stacks the critic passes of one sess.run along the batch axis)
            return (torch.sigmoid(logits) if _prob else None), logits

    # ---- generator (stageII/model.py:135-201) -----------------------------------------------------------------------------
    def generator_encode_image(self, image, is_training=True):
        bn_init = self.batch_norm_init
        h = conv2d(image, self.gf_dim, ks=(3, 3), s=(1, 1), act=relu)
        h = batch_norm(conv2d(h, self.gf_dim * 2, ks=(4, 4), s=(2, 2)), train=is_training, init=bn_init, act=relu)
        return batch_norm(conv2d(h, self.gf_dim * 4, ks=(4, 4), s=(2, 2)), train=is_training, init=bn_init, act=relu)

    def generator_residual_layer(self, input_layer, is_training=True):
        bn_init = self.batch_norm_init
        h = batch_norm(conv2d(input_layer, self.gf_dim * 4, ks=(4, 4), s=(1, 1)), train=is_training, init=bn_init, act=relu)
        h = batch_norm(conv2d(h, self.gf_dim * 4, ks=(4, 4), s=(1, 1)), train=is_training, init=bn_init)
        return add(input_layer, h, act=relu)

    def generator_upsample(self, input_layer, is_training=True):
        h = input_layer
        for nf in (self.gf_dim * 2, self.gf_dim, self.gf_dim // 2, self.gf_dim // 4):
            h = conv2d_transpose(h, nf, ks=(4, 4), init=self.w_init)
            h = conv2d(h, nf, ks=(3, 3), s=(1, 1))
            h = batch_norm(h, train=is_training, init=self.batch_norm_init, act=relu)
        return conv2d(h, self.image_dims[-1], ks=(3, 3), s=(1, 1), act=tanh)

    def generator(self, image, embed, is_training=True, reuse=False, cond_noise=True, noise=None):
        """image: the Stage-I output [B,64,64,3] -> (image [B,256,256,3], mean, log_sigma)"""
        with K.math_scope(*self.net_math.get('g_net', (None, None))), S.variable_scope('stageII_g_net', reuse=reuse):
            encoded = self.generator_encode_image(image, is_training=is_training)          # [B,16,16,4*gf]
            mean, log_sigma = self.generate_conditionals(embed)
            code = self.sample_normal_conditional(mean, log_sigma, cond_noise, noise)
            h = concat_tile(encoded, code)
            h = conv2d(h, self.gf_dim * 4, ks=(3, 3), s=(1, 1))
            h = batch_norm(h, train=is_training, init=self.batch_norm_init, act=relu)
            for _ in range(4):
                h = self.generator_residual_layer(h, is_training=is_training)
            return self.generator_upsample(h, is_training=is_training), mean, log_sigma
