"""Golden vectors produced by THE REFERENCE ITSELF (run in the build container only; /root/reference does not travel):

    python tests/golden/make_reference_golden.py        ->  tests/golden/reference_dataset.npz, reference_utils.npz

The reference's data pipeline (preprocess/dataset.py) and image-grid helpers (utils/utils.py: merge,
inverse_transform, get_balanced_factorization) are pure Python/NumPy.  They are imported here from /root/reference with
empty stand-in modules for the imports they do not use on these code paths (tensorflow, imageio, scipy.misc,
sklearn.externals.joblib) — nothing of the reference is copied; only inputs and the outputs it computed are stored.
These are the only fixtures in this repo that pin parity against the reference's own code (everything that needs TF-1.4
stays "parity unpinned")."""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    if not os.path.isdir(REF):
        raise RuntimeError('%s is not available (fixtures can only be regenerated in the build container)' % REF)
    sk = _stub('sklearn'); ext = _stub('sklearn.externals'); jb = _stub('sklearn.externals.joblib'); sk.externals = ext; ext.joblib = jb
    tf = _stub('tensorflow'); contrib = _stub('tensorflow.contrib'); slim = _stub('tensorflow.contrib.slim')
    tf.contrib = contrib; contrib.slim = slim
    sp = _stub('scipy.misc')
    ff = types.SimpleNamespace(download=lambda: None)
    _stub('imageio', plugins=types.SimpleNamespace(ffmpeg=ff))
    sys.path.insert(0, REF)
    import importlib
    ds = importlib.import_module('preprocess.dataset')
    ut = importlib.import_module('utils.utils')
    return ds, ut


def make_dataset(ds):
    """Synthetic 76x76 uint8 images / 5 captions per image / class ids, then the reference's own next_batch /
    next_batch_test under fixed seeds; every call's outputs are recorded in call order."""
    rng = np.random.default_rng(2024)
    N, S, E, D = 37, 76, 5, 24
    images = rng.integers(0, 256, (N, S, S, 3), dtype=np.uint8)
    embeddings = rng.standard_normal((N, E, D)).astype(np.float32)
    class_id = rng.integers(0, 6, N)
    filenames = ['jpg/image_%05d' % i for i in range(N)]
    out = {'in/images': images, 'in/embeddings': embeddings, 'in/class_id': class_id, 'in/np_seed': np.array(123),
           'in/py_seed': np.array(456), 'in/imsize': np.array(64)}
    np.random.seed(123); random.seed(456)
    d = ds.Dataset(images, 64, embeddings, filenames, '/nonexistent', None, True, class_id)
    calls = [(8, 4, True), (8, 4, True), (8, 2, True), (8, 4, False), (8, 4, True), (8, 3, True), (16, 4, True)]   # crosses 2 epoch ends
    for c, (B, window, wrong) in enumerate(calls):
        img, wimg, emb, caps, lab = d.next_batch(B, window, wrong_img=wrong, embeddings=True)
        out['call%d/args' % c] = np.array([B, window, int(wrong)])
        out['call%d/images' % c] = np.asarray(img, np.float32); assert np.array_equal(out['call%d/images' % c], img)
        if wrong:
            out['call%d/wrong_images' % c] = np.asarray(wimg, np.float32); assert np.array_equal(out['call%d/wrong_images' % c], wimg)
        out['call%d/embeddings' % c] = np.asarray(emb); assert out['call%d/embeddings' % c].dtype == np.float32
    # no augmentation: images pass through at full size
    np.random.seed(7); random.seed(8)
    d2 = ds.Dataset(images, 64, embeddings, filenames, '/nonexistent', None, False, class_id)
    img, wimg, emb, _, _ = d2.next_batch(5, 4, wrong_img=True, embeddings=True)
    out['noaug/images'] = np.asarray(img, np.float32); out['noaug/wrong_images'] = np.asarray(wimg, np.float32)
    out['noaug/embeddings'] = np.asarray(emb)
    return out


def make_utils(ut):
    rng = np.random.default_rng(5)
    out = {}
    for c, (n, h, w, ch, size) in enumerate([(6, 4, 5, 3, (2, 3)), (4, 3, 3, 1, (2, 2)), (5, 2, 2, 4, (2, 3))]):
        imgs = rng.uniform(-1, 1, (n, h, w, ch)).astype(np.float32)
        out['merge%d/images' % c] = imgs; out['merge%d/size' % c] = np.array(size)
        out['merge%d/out' % c] = ut.merge(imgs, size)
        out['merge%d/inv' % c] = ut.inverse_transform(imgs)
    xs = np.array([1, 2, 3, 4, 12, 16, 17, 36, 48, 49, 64, 100, 360, 997])
    out['factor/x'] = xs
    out['factor/ab'] = np.array([ut.get_balanced_factorization(int(x)) for x in xs])
    return out


if __name__ == '__main__':
    ds, ut = import_reference()
    a = make_dataset(ds)
    np.savez_compressed(os.path.join(HERE, 'reference_dataset.npz'), **a)
    b = make_utils(ut)
    np.savez_compressed(os.path.join(HERE, 'reference_utils.npz'), **b)
    print('reference_dataset: %d arrays; reference_utils: %d arrays' % (len(a), len(b)))
    print(b['factor/ab'].tolist())
