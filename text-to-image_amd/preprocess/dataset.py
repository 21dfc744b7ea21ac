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
        self._saveIDs = np.arange(self._num_examples)
        np.random.shuffle(self._saveIDs)
        return self._saveIDs

    def readCaptions(self, filenames, class_id):
        name = filenames
        if name.find('jpg/') != -1:                                  # flowers: captions live under class_%05d/
            name = name.replace('jpg/', 'class_%05d/' % (class_id + 1))
        with open('%s/text_c10/%s.txt' % (self.workdir, name), 'r') as f:
            captions = f.read().split('\n')
        return [cap for cap in captions if len(cap) > 0]

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

    def next_batch(self, batch_size, window=None, wrong_img=False, embeddings=False, labels=False):
        """-> [images, wrong_images | None, embeddings | None, captions | None, class ids | None]  (dataset.py:122-184)"""
        start = self._index_in_epoch
        self._index_in_epoch += batch_size
        if self._index_in_epoch > self._num_examples:      # finished epoch: reshuffle, start over
            self._epochs_completed += 1
            self._perm = np.arange(self._num_examples)
            np.random.shuffle(self._perm)
            start = 0
            self._index_in_epoch = batch_size
            assert batch_size <= self._num_examples
        end = self._index_in_epoch
        current_ids = self._perm[start:end]
        ret_list = [self._images_for(current_ids)]
        if wrong_img:
            fake_ids = np.random.randint(self._num_examples, size=batch_size)
            collision_flag = (self._class_id[current_ids] == self._class_id[fake_ids])
            fake_ids[collision_flag] = (fake_ids[collision_flag] + np.random.randint(100, 200)) % self._num_examples
            ret_list.append(self._images_for(fake_ids))
        else:
            ret_list.append(None)
        if self._embeddings is not None and embeddings:
            filenames = [self._filenames[i] for i in current_ids] if self._filenames is not None else [None] * len(current_ids)
            class_id = [self._class_id[i] for i in current_ids]
            sampled_embeddings, sampled_captions = self.sample_embeddings(current_ids, filenames, class_id, window)
            ret_list.append(sampled_embeddings)
            ret_list.append(sampled_captions)
        else:
            ret_list.append(None)
            ret_list.append(None)
        if self._labels is not None and labels:
            ret_list.append([self._class_id[i] for i in current_ids])
        else:
            ret_list.append(None)
        return ret_list

    def next_batch_test(self, batch_size, start, max_captions):
        """-> [images, [embeddings of caption 0, 1, ...], save ids, captions]  (dataset.py:186-216)"""
        if (start + batch_size) > self._num_examples:
            end = self._num_examples
            start = end - batch_size
        else:
            end = start + batch_size
        ids = np.arange(start, end)
        sampled_images = self._images_for(ids)
        sampled_embeddings = self._embeddings[start:end]
        embedding_num = sampled_embeddings.shape[1]
        sampled_captions = []
        if self._filenames is not None and self.workdir is not None and os.path.isdir(os.path.join(self.workdir, 'text_c10')):
            for i in range(start, end):
                sampled_captions.append(self.readCaptions(self._filenames[i], self._class_id[i]))
        batches = [sampled_embeddings[:, i, :].squeeze() for i in range(min(max_captions, embedding_num))]
        return [sampled_images, batches, self._saveIDs[start:end], sampled_captions]

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
