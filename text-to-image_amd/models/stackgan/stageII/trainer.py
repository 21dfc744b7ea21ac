"""ConditionalGanTrainer (Stage-II) — reference models/stackgan/stageII/trainer.py:11-177.  Same losses and schedule as
Stage-I with real label 0.95 (trainer.py:27); the generator input is the Stage-I generator's image, produced inside the
same run in training mode with frozen weights (stageII/model.py:50)."""
import torch

from ..stageI.trainer import ConditionalGanTrainer as _StageITrainer


class ConditionalGanTrainer(_StageITrainer):
    REAL_LABEL = 0.95
    NOISE_KEYS = ('ca_noise_d', 'ca_noise_g', 'ca_noise_d_s1', 'ca_noise_g_s1')

    def _noise_dim(self, key):
        return self.model.stagei.compressed_embed_dim if key.endswith('_s1') else self.model.compressed_embed_dim

    def __init__(self, sess, model, dataset, cfg, cfg_stage_i=None):
        self.cfg_stage_i = cfg_stage_i
        super().__init__(sess, model, dataset, cfg)

    def _generate(self, feed, which):
        m = self.model
        with torch.no_grad():      # Stage-I variables are in no var_list; under update_ops() its BN moving averages move
            img64, _, _ = m.stagei.generator(feed['z'], feed['phi_inputs'], reuse=True, noise=feed.get('ca_noise_%s_s1' % which))
        return m.generator(img64, feed['phi_inputs'], reuse=True, noise=feed.get('ca_noise_' + which))
