"""Synthetic stand-in for the reference's TextDataset (reference preprocess/dataset.py:122-184, out of scope for the
hot path: BASELINE.json asks for synthetic inputs).  Produces, already resident on the device, exactly the shapes and
ranges the real pipeline feeds the trainer: images / mismatched images in [-1,1) (dataset.py:150), 1024-d text
embeddings ~ N(0,1) (mean of `window` caption embeddings, dataset.py:98-120)."""
import torch


class _Split(object):
    def __init__(self, cfg, device, seed, num_examples):
        m = cfg.MODEL
        self.shape = (m.IMAGE_SHAPE.H, m.IMAGE_SHAPE.W, m.IMAGE_SHAPE.D)
        self.embed_dim = m.EMBED_DIM
        self.device = device
        self.num_examples = num_examples
        self.gen = torch.Generator(device=device).manual_seed(seed)

    def next_batch(self, batch_size, window=4, embeddings=True, wrong_img=True):
        """-> images, wrong_images, embed, None, None   (same tuple arity as dataset.py:122-184)"""
        img = torch.rand((batch_size,) + self.shape, generator=self.gen, device=self.device) * 2.0 - 1.0
        wrong = torch.rand((batch_size,) + self.shape, generator=self.gen, device=self.device) * 2.0 - 1.0 if wrong_img else None
        emb = torch.randn((batch_size, self.embed_dim), generator=self.gen, device=self.device) if embeddings else None
        return img, wrong, emb, None, None

    def next_batch_test(self, batch_size, start, max_captions):
        """-> images, embeddings [max_captions,B,E], None, captions   (dataset.py:186-216)"""
        img = torch.rand((batch_size,) + self.shape, generator=self.gen, device=self.device) * 2.0 - 1.0
        emb = torch.randn((max_captions, batch_size, self.embed_dim), generator=self.gen, device=self.device)
        return img, emb, None, [['synthetic caption %d' % i] for i in range(batch_size)]


class SyntheticTextDataset(object):
    def __init__(self, cfg, device, seed=1, num_examples=8192):
        self.train = _Split(cfg, device, seed, num_examples)
        self.test = _Split(cfg, device, seed + 1, num_examples // 8)
