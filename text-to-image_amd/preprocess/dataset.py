"""Dataset / TextDataset — reference preprocess/dataset.py:24-283 with the per-batch work on the GPU.

Same classes, constructor arguments, methods and return tuples as the reference.  The image store ([N,S,S,3] uint8) and
the caption embeddings ([N,5,D] float32) are resident in HBM; a batch is produced by two HBM-bound kernels of
libt2i_hip.so (t2i_crop_flip_normalize: gather by id + scale to [-1,1] + random crop + horizontal flip;
t2i_gather_mean: mean of `window` chosen caption embeddings) and never touches the host.  What stays on the host is the
reference's random *decisions* — epoch permutation, crop offsets, flips, mismatched-image ids, caption choices — drawn
from the same global NumPy / `random` generators in the same order as the reference, so that equal seeds give equal
batches (pinned bit-for-bit against the reference's own output: tests/golden/reference_dataset.npz)."""
import os
import pickle
import random

import numpy as np
import torch

from .. import kernels as K

FINAL_SIZE_TO_ORIG = {4: 4, 8: 8, 16: 16, 32: 38, 64: 76, 128: 152, 256: 304, 299: 360, 512: 600}   # dataset.py:11-21


class Dataset(object):
    def __init__(self, images, imsize, embeddings=None, filenames=None, workdir=None, labels=None, aug_flag=True,
                 class_id=None, class_range=None, device=None):
        self.device = torch.device(device) if device is not None else torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self._images = self._to_device(images, torch.uint8)                 # [N,S,S,3] uint8, resident
        self._embeddings = self._to_device(embeddings, torch.float32) if embeddings is not None else None
        self._filenames = filenames
        self.workdir = workdir
        self._labels = labels
        self._epochs_completed = -1
        self._num_examples = len(images)
        self._saveIDs = self.saveIDs()          # (consumes one np.random.shuffle, like the reference constructor)
        self._index_in_epoch = self._num_examples   # shuffle on first run
        self._aug_flag = aug_flag
        self._class_id = None if class_id is None else np.asarray(class_id)
        self._class_range = class_range
        self._imsize = imsize
        self._perm = None

    def _to_device(self, a, dtype):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=self.device, dtype=dtype).contiguous()

    images = property(lambda self: self._images)
    embeddings = property(lambda self: self._embeddings)
    filenames = property(lambda self: self._filenames)
    num_examples = property(lambda self: self._num_examples)
    epochs_completed = property(lambda self: self._epochs_completed)
    class_ids = property(lambda self: self._class_id)

    def saveIDs(self):
        """A shuffled id list for the test-time dumps (dataset.py:66-69); one np.random.shuffle, as in the reference constructor."""
        order = np.arange(self._num_examples)
        np.random.shuffle(order)
        self._saveIDs = order
        return order

    def _caption_path(self, filename, class_id):
        """flowers keeps its caption files under class_%05d/ instead of jpg/ (dataset.py:71-76)."""
        if 'jpg/' in filename:
            filename = filename.replace('jpg/', 'class_%05d/' % (class_id + 1))
        return os.path.join(self.workdir, 'text_c10', filename + '.txt')

    def readCaptions(self, filenames, class_id):
        """The non-empty lines of one image's caption file (dataset.py:71-81)."""
        with open(self._caption_path(filenames, class_id), 'r') as f:
            return [line for line in f.read().split('\n') if line]

    # ---- the reference's random decisions, in its order (dataset.py:83-96) ------------------------------------------------
    def _draw_crops(self, n, ori_size):
        """-> (row0, col0, flip) int32 arrays.  Per image: two np.random.random() then one random.random(); the reference
        slices rows with its `w1` and columns with its `h1` (dataset.py:89-91) — the second and the first draw."""
        row0, col0, flip = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        for i in range(n):
            h1 = int(np.floor((ori_size - self._imsize) * np.random.random()))
            w1 = int(np.floor((ori_size - self._imsize) * np.random.random()))
            row0[i], col0[i] = w1, h1
            flip[i] = 1 if random.random() > 0.5 else 0
        return row0, col0, flip

    def _images_for(self, ids):
        """ids: int array -> float32 [B,s,s,3] on the device: scale, then (if aug_flag) random crop + flip."""
        ids = np.asarray(ids)
        S = int(self._images.shape[1])
        if self._aug_flag:
            row0, col0, flip = self._draw_crops(len(ids), S)
            size = self._imsize
        else:
            row0 = col0 = flip = np.zeros(len(ids), np.int32)
            size = S
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)
        return K.crop_flip_normalize(self._images, dev(ids), dev(row0), dev(col0), dev(flip), size)

    def transform(self, images):
        """Reference signature: float images [B,S,S,3] in [-1,1] -> random crop + flip (host tensor math; the training
        path uses the fused kernel in _images_for instead)."""
        if not self._aug_flag:
            return images
        row0, col0, flip = self._draw_crops(images.shape[0], images.shape[1])
        out = []
        for i in range(images.shape[0]):
            c = images[i][row0[i]:row0[i] + self._imsize, col0[i]:col0[i] + self._imsize, :]
            out.append(torch.flip(c, dims=[1]) if flip[i] else c)
        return torch.stack(out)

    def sample_embeddings(self, embeddings_ids, filenames, class_id, sample_num):
        """Mean of `sample_num` of each image's caption embeddings (dataset.py:98-120).  embeddings_ids: the image ids."""
        emb = self._embeddings
        if emb.dim() == 2 or emb.shape[1] == 1:
            return emb[torch.as_tensor(np.asarray(embeddings_ids), device=self.device)].squeeze(), []
        embedding_num = emb.shape[1]
        choice, captions = [], []
        for i in range(len(embeddings_ids)):
            randix = np.random.choice(embedding_num, sample_num, replace=False)
            if sample_num == 1:
                captions.append(self.readCaptions(filenames[i], class_id[i])[int(randix)])
            choice.append(np.atleast_1d(randix))
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)
        out = K.gather_mean(emb, dev(np.asarray(embeddings_ids)), dev(np.stack(choice)))
        return out.squeeze(), captions

    # ---- batches.  The order of the random draws IS the interface (equal seeds -> the reference's batches, bit for bit):
    #      [epoch shuffle] -> crops/flips of the batch -> mismatched ids -> ONE collision offset (drawn whether or not anything
    #      collides) -> crops/flips of the mismatched images -> caption choices.
    def _advance(self, batch_size):
        """Ids of the next `batch_size` examples of the current epoch permutation; a batch that would run past the end starts a
        new epoch (fresh permutation) instead — the tail of the old one is dropped (dataset.py:124-139)."""
        stop = self._index_in_epoch + batch_size
        if stop > self._num_examples:
            if batch_size > self._num_examples:
                raise AssertionError('batch of %d from %d examples' % (batch_size, self._num_examples))
            self._epochs_completed += 1
            self._perm = np.arange(self._num_examples)
            np.random.shuffle(self._perm)
            stop = batch_size
        self._index_in_epoch = stop
        return self._perm[stop - batch_size:stop]

    def _mismatched_ids(self, ids):
        """Uniform ids for the "wrong image" of every example; the ones that landed in the example's own class are moved on by
        one shared offset in [100, 200) (dataset.py:154-159)."""
        wrong = np.random.randint(self._num_examples, size=len(ids))
        offset = np.random.randint(100, 200)
        same_class = self._class_id[ids] == self._class_id[wrong]
        return np.where(same_class, (wrong + offset) % self._num_examples, wrong)

    def _text_for(self, ids, window):
        names = [None] * len(ids) if self._filenames is None else [self._filenames[i] for i in ids]
        return self.sample_embeddings(ids, names, [self._class_id[i] for i in ids], window)

    def next_batch(self, batch_size, window=None, wrong_img=False, embeddings=False, labels=False):
        """-> [images, wrong_images | None, embeddings | None, captions | None, class ids | None]  (dataset.py:122-184)"""
        ids = self._advance(batch_size)
        images = self._images_for(ids)
        wrong = self._images_for(self._mismatched_ids(ids)) if wrong_img else None
        text, captions = self._text_for(ids, window) if (embeddings and self._embeddings is not None) else (None, None)
        classes = [self._class_id[i] for i in ids] if (labels and self._labels is not None) else None
        return [images, wrong, text, captions, classes]

    def next_batch_test(self, batch_size, start, max_captions):
        """-> [images, [embeddings of caption 0, 1, ...], save ids, captions]  (dataset.py:186-216).  A window that would run
        past the end is moved back so that it ends at the last example."""
        start = min(start, self._num_examples - batch_size)
        window = slice(start, start + batch_size)
        images = self._images_for(np.arange(window.start, window.stop))
        emb = self._embeddings[window]
        per_caption = [emb[:, j, :].squeeze() for j in range(min(max_captions, emb.shape[1]))]
        captions = []
        have_text = self._filenames is not None and self.workdir is not None and os.path.isdir(os.path.join(self.workdir, 'text_c10'))
        if have_text:
            captions = [self.readCaptions(self._filenames[i], self._class_id[i]) for i in range(window.start, window.stop)]
        return [images, per_caption, self._saveIDs[window], captions]

    def class_to_index(self):
        return {class_id: idx for idx, class_id in enumerate(np.unique(self._class_id))}


def _unpickle(path, **kw):
    with open(path, 'rb') as f:
        return pickle.load(f, **kw)


class TextDataset(object):
    """The on-disk side of the pipeline (role of reference preprocess/dataset.py:229-283): a dataset directory holds one
    sub-directory per split (`train`, `test`), each with four pickles —

        <orig>images.pickle              joblib dump, uint8 [N, orig, orig, 3]; orig = FINAL_SIZE_TO_ORIG[size] (76 for 64x64)
        char-CNN-RNN-embeddings.pickle   [N, captions per image, D] float (Python-2 pickle: bytes encoding)
        filenames.pickle                 N relative image names (caption files are found through them)
        class_info.pickle                N class ids, 1-based on disk, 0-based in memory

    `get_data(split_dir)` reads them and returns a `Dataset` whose image store and embeddings are resident on the device;
    the caller assigns the result to `.train` / `.test` (reference models/wgancls/run.py:33-40)."""

    EMBEDDINGS = 'char-CNN-RNN-embeddings.pickle'
    FILENAMES = 'filenames.pickle'
    CLASSES = 'class_info.pickle'

    def __init__(self, workdir, size, device=None):
        if size not in FINAL_SIZE_TO_ORIG:
            raise RuntimeError('Size {} not supported'.format(size))
        self.workdir, self.size, self.device = workdir, size, device
        self.image_shape = [size, size, 3]
        self.image_dim = size * size * 3
        self.image_filename = '/%dimages.pickle' % FINAL_SIZE_TO_ORIG[size]      # leading '/' like the reference attribute
        self.embedding_filename = '/' + self.EMBEDDINGS
        self.embedding_shape = None            # [D], known once a split has been read
        self.train = None
        self.test = None

    @property
    def name(self):
        return os.path.basename(os.path.normpath(self.workdir))

    def get_data(self, pickle_path, aug_flag=True):
        import joblib
        split = pickle_path.rstrip('/')
        images = np.asarray(joblib.load(split + self.image_filename))
        embeddings = np.asarray(_unpickle(os.path.join(split, self.EMBEDDINGS), encoding='bytes'))
        self.embedding_shape = [int(embeddings.shape[-1])]
        filenames = _unpickle(os.path.join(split, self.FILENAMES))
        class_id = np.asarray(_unpickle(os.path.join(split, self.CLASSES), encoding='bytes')) - 1
        return Dataset(images, self.size, embeddings=embeddings, filenames=filenames, workdir=self.workdir, labels=class_id,
                       aug_flag=aug_flag, class_id=class_id, device=self.device)
