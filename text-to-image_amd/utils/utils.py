"""Image-grid / caption helpers of the reference's utils/utils.py:24-58,82-109: what the
trainers call around the sampler (`save_images(samples, get_balanced_factorization(n), path)`, trainer.py:104-126).
Host-side NumPy like the reference; `merge`, `inverse_transform` and `get_balanced_factorization` are pinned against the
reference's own outputs (tests/golden/reference_utils.npz).  PNG encoding uses Pillow (the reference's scipy.misc.imsave
is gone from SciPy): bytescale to uint8 exactly as imsave did for float input (min -> 0, max -> 255)."""
import os

import numpy as np


def merge(images, size):
    """[n,h,w,c] -> one [size[0]*h, size[1]*w(,c)] float64 grid, row-major (utils.py:30-49)."""
    images = np.asarray(images)
    h, w = images.shape[1], images.shape[2]
    if images.shape[3] in (3, 4):
        img = np.zeros((h * size[0], w * size[1], images.shape[3]))
        for idx, image in enumerate(images):
            i, j = idx % size[1], idx // size[1]
            img[j * h:j * h + h, i * w:i * w + w, :] = image
        return img
    if images.shape[3] == 1:
        img = np.zeros((h * size[0], w * size[1]))
        for idx, image in enumerate(images):
            i, j = idx % size[1], idx // size[1]
            img[j * h:j * h + h, i * w:i * w + w] = image[:, :, 0]
        return img
    raise ValueError('in merge(x,size) x parameter must have dimensions: HxW or HxWx3 or HxWx4')


def inverse_transform(images):
    return (np.asarray(images) + 1.) / 2.


def get_balanced_factorization(x):
    """x = a*b with a <= b as close as possible (utils.py:82-93)."""
    if x <= 0:
        raise ValueError('Argument must be a strictly positive number but it is %d' % x)
    a = int(np.sqrt(x))
    if a ** 2 == x:
        return a, a
    for a in range(a, 0, -1):
        if x % a == 0:
            return a, x // a
    raise ValueError('Error finding the balanced factorization of %d' % x)


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
    """utils.py:96-109"""
    if not os.path.exists(directory):
        os.makedirs(directory)
    filepath = os.path.join(directory, 'captions.txt')
    if os.path.exists(filepath):
        os.remove(filepath)
    with open(filepath, 'w+') as f:
        f.write('Captions of the sampled x:\n')
        for idx, caption in enumerate(captions):
            f.write('{}: {}\n'.format(idx + 1, caption[0]))
