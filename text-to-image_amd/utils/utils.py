"""Image-grid / caption helpers of the reference's utils/utils.py:24-58,82-109: what the
trainers call around the sampler (`save_images(samples, get_balanced_factorization(n), path)`, trainer.py:104-126).
Host-side NumPy like the reference; `merge`, `inverse_transform` and `get_balanced_factorization` are pinned against the
reference's own outputs (tests/golden/reference_utils.npz).  PNG encoding uses Pillow (the reference's scipy.misc.imsave
is gone from SciPy): bytescale to uint8 exactly as imsave did for float input (min -> 0, max -> 255)."""
import math
import os

import numpy as np


def merge(images, size):
    """Tile n images [n,h,w,c] into one (rows, cols) = `size` grid, filled row by row; cells past n stay zero.
    -> float64 [rows*h, cols*w, c] for c in (3, 4), [rows*h, cols*w] for c == 1 (role of reference utils/utils.py:30-49;
    done as one reshape/transpose of a zero-padded stack instead of a per-image paste loop)."""
    images = np.asarray(images)
    n, h, w, c = images.shape
    rows, cols = int(size[0]), int(size[1])
    if c not in (1, 3, 4):
        raise ValueError('merge: images must have 1, 3 or 4 channels, got an array of shape %s' % (images.shape,))
    if n > rows * cols:
        raise ValueError('merge: %d images do not fit a %d x %d grid' % (n, rows, cols))
    cells = np.zeros((rows * cols, h, w, c))
    cells[:n] = images
    grid = cells.reshape(rows, cols, h, w, c).transpose(0, 2, 1, 3, 4).reshape(rows * h, cols * w, c)
    return grid[:, :, 0] if c == 1 else grid


def inverse_transform(images):
    return (np.asarray(images) + 1.) / 2.


def get_balanced_factorization(x):
    """(a, b) with a * b == x, a <= b and a as large as possible: the most square grid for x sample images (role of
    reference utils/utils.py:82-93)."""
    x = int(x)
    if x < 1:
        raise ValueError('get_balanced_factorization needs a positive integer, got %d' % x)
    a = max(d for d in range(1, math.isqrt(x) + 1) if x % d == 0)
    return a, x // a


def _bytescale(img):
    """scipy.misc.imsave's float handling: linear map of [min, max] onto [0, 255]."""
    lo, hi = float(img.min()), float(img.max())
    if hi == lo:
        return np.zeros(img.shape, np.uint8)
    return np.clip(np.round((img - lo) * (255.0 / (hi - lo))), 0, 255).astype(np.uint8)


def imsave(images, size, path):
    from PIL import Image
    image = np.squeeze(merge(images, size))
    Image.fromarray(_bytescale(image)).save(path)
    return path


def save_images(images, size, image_path):
    """images: [n,h,w,c] in [-1,1] (NumPy or a device tensor) -> PNG grid (utils.py:24-27)."""
    if hasattr(images, 'detach'):
        images = images.detach().float().cpu().numpy()
    d = os.path.dirname(image_path)
    if d and not os.path.exists(d):
        os.makedirs(d)
    return imsave(inverse_transform(images), size, image_path)


def save_captions(directory, captions):
    """Write `<directory>/captions.txt`: a header line, then "<1-based index>: <first caption of the sample>" per sampled
    image; an existing file is replaced (role of reference utils/utils.py:96-109)."""
    os.makedirs(directory, exist_ok=True)
    lines = ['Captions of the sampled x:'] + ['%d: %s' % (i, cap[0]) for i, cap in enumerate(captions, 1)]
    with open(os.path.join(directory, 'captions.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
