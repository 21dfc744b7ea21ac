"""models/wgancls/run.py as the reference's entry point (reference models/wgancls/run.py:19-70): config -> directories ->
pickled TextDataset from cfg.DATASET_DIR -> mode switch -> WGanClsTrainer.train() with its side effects."""
import os
import pickle
import random

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_split(path, n, rng, emb_dim):
    import joblib
    os.makedirs(path)
    joblib.dump(list(rng.integers(0, 256, (n, 76, 76, 3), dtype=np.uint8)), os.path.join(path, '76images.pickle'))
    pickle.dump(list(rng.standard_normal((n, 5, emb_dim)).astype(np.float32)), open(os.path.join(path, 'char-CNN-RNN-embeddings.pickle'), 'wb'))
    pickle.dump(['jpg/image_%05d' % i for i in range(n)], open(os.path.join(path, 'filenames.pickle'), 'wb'))
    pickle.dump([int(c) for c in rng.integers(1, 6, n)], open(os.path.join(path, 'class_info.pickle'), 'wb'))


def _make_cfg(tmp_path, train_flag, n_train=12, n_test=9, batch=4, sample_num=9, data=True):
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'text-to-image_amd', 'models', 'wgancls', 'cfg', 'flowers.yml')))
    d = str(tmp_path)
    cfg.update(DATASET_DIR=d + '/data/flowers/', CHECKPOINT_DIR=d + '/ckpt/', LOGS_DIR=d + '/logs/', SAMPLE_DIR=d + '/samples/')
    cfg['MODEL'].update(Z_DIM=8, EMBED_DIM=32, COMPRESSED_EMBED_DIM=16, GF_DIM=8, DF_DIM=8)
    cfg['TRAIN'].update(FLAG=train_flag, BATCH_SIZE=batch, SAMPLE_NUM=sample_num, SAMPLE_PERIOD=2, SUMMARY_PERIOD=2, MAX_STEPS=7)
    if data:
        rng = np.random.default_rng(0)
        _write_split(d + '/data/flowers/train', n_train, rng, 32)
        _write_split(d + '/data/flowers/test', n_test, rng, 32)
    path = d + '/cfg.yml'
    yaml.safe_dump(cfg, open(path, 'w'))
    return path


def test_load_dataset_reads_the_pickled_splits(tmp_path, monkeypatch):
    """run.load_dataset = reference run.py:33-40: TextDataset(DATASET_DIR, 64) with .test and .train read from the pickles
    (a 12 / 9 image set written here); a missing directory is an error, not a silent switch to synthetic data."""
    import t2i_amd  # noqa: F401
    from oracle import np_dataset as OD
    from t2i_amd import kernels as K
    from t2i_amd.models.wgancls import run
    from t2i_amd.utils.config import config_from_yaml
    n = lambda t: t.cpu().numpy()
    monkeypatch.setattr(K, 'crop_flip_normalize', lambda src, ids, r0, c0, fl, size: torch.from_numpy(
        OD.crop_flip_normalize(n(src), n(ids), n(r0), n(c0), n(fl), size)))
    monkeypatch.setattr(K, 'gather_mean', lambda emb, ids, choice: torch.from_numpy(OD.gather_mean(n(emb), n(ids), n(choice))))
    cfg = config_from_yaml(_make_cfg(tmp_path, True))
    np.random.seed(0); random.seed(0)
    ds = run.load_dataset(cfg, 'cpu')
    assert ds.train.num_examples == 12 and ds.test.num_examples == 9 and ds.embedding_shape == [32] and ds.name == 'flowers'
    img, wrong, emb, _, _ = ds.train.next_batch(4, 4, embeddings=True, wrong_img=True)
    assert tuple(img.shape) == (4, 64, 64, 3) and tuple(wrong.shape) == (4, 64, 64, 3) and tuple(emb.shape) == (4, 32)
    assert float(img.min()) >= -1.0 and float(img.max()) <= 1.0
    _, cond, _, _ = ds.test.next_batch_test(9, 0, 1)
    assert len(cond) == 1 and tuple(cond[0].shape) == (9, 32)
    syn = run.load_dataset(cfg, 'cpu', synthetic=True)
    assert tuple(syn.train.next_batch(4, 4)[0].shape) == (4, 64, 64, 3)
    cfg.DATASET_DIR = str(tmp_path / 'nowhere')
    with pytest.raises(FileNotFoundError):
        run.load_dataset(cfg, 'cpu')


def test_main_honours_the_mode_flags(tmp_path):
    """TRAIN.FLAG False (the shipped yml) selects the reference's visualiser, EVAL.FLAG its evaluator: both are out of scope
    and must say so; the output directories are created first, as in the reference."""
    import t2i_amd  # noqa: F401
    from t2i_amd.models.wgancls import run
    path = _make_cfg(tmp_path, False, data=False)
    with pytest.raises(NotImplementedError, match='TRAIN.FLAG'):
        run.main(['--cfg', path])
    for d in ('ckpt', 'logs', 'samples'):
        assert os.path.isdir(str(tmp_path / d))
    cfg = yaml.safe_load(open(path)); cfg['EVAL']['FLAG'] = True
    yaml.safe_dump(cfg, open(path, 'w'))
    with pytest.raises(NotImplementedError, match='EVAL.FLAG'):
        run.main(['--cfg', path, '--train'])


@pytest.mark.gpu
@pytest.mark.parametrize('graphs', [1, 0])
def test_main_trains_with_side_effects(tmp_path, graphs):
    """`run.py --cfg <yml>` with TRAIN.FLAG: True on a pickled data set: 6 iterations (captured into hipGraphs after the
    first when graphs=1), the sampler at SAMPLE_NUM=9 > BATCH_SIZE=4 running eagerly between replays (the workspace may
    grow there: the captured graphs must keep their own buffer — kernels.workspace), captions, PNG grids at idx 2/4/6, a
    checkpoint at idx 2; a second run resumes from it."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import t2i_amd  # noqa: F401
    from t2i_amd.models.wgancls import run
    path = _make_cfg(tmp_path, True)
    np.random.seed(0); random.seed(0)
    out = run.main(['--cfg', path, '--graphs', str(graphs)])
    torch.cuda.synchronize()
    assert np.isfinite(float(out['d']['D_loss'])) and np.isfinite(float(out['g']['G_loss']))
    samples = sorted(os.listdir(str(tmp_path / 'samples')))
    assert 'captions.txt' in samples and [s for s in samples if s.endswith('.png')] == [
        'train_00_0002.png', 'train_01_0004.png', 'train_02_0006.png'], samples
    from PIL import Image
    assert Image.open(str(tmp_path / 'samples' / 'train_00_0002.png')).size == (3 * 64, 3 * 64)
    # the TensorBoard event file of reference trainer.py:20-47,104-107: version record + one merged summary at idx 2, 4, 6
    from t2i_amd.utils import summary as S
    logs_dir = str(tmp_path / 'logs')
    ev_files = [f for f in os.listdir(logs_dir) if f.startswith('events.out.tfevents.')]
    assert len(ev_files) == 1, ev_files
    ev = S.read_events(os.path.join(logs_dir, ev_files[0]))
    assert ev[0]['file_version'] == 'brain.Event:2' and [e['step'] for e in ev[1:]] == [2, 4, 6]
    tags = [v['tag'] for v in ev[1]['values']]
    assert tags[:6] == ['x/image/0', 'x/image/1', 'x/image/2', 'G_img/image/0', 'G_img/image/1', 'G_img/image/2'] and tags[6:8] == ['z', 'z_sample']
    assert tags[8:] == ['G_loss_wass', 'kl_loss', 'G_loss', 'D_loss_real', 'D_loss_fake', 'real_gp', 'D_loss', 'reg_loss', 'wdist', 'wdist2',
                        'd_loss_mismatch', 'real_gp2', 'kt', 'balance_loss'], tags
    last = {v['tag']: v for v in ev[-1]['values']}
    assert abs(last['D_loss']['simple_value'] - float(out['d']['D_loss'])) <= 1e-6 * max(1.0, abs(float(out['d']['D_loss'])))
    assert S.decode_png(last['G_img/image/0']['image']['png']).shape == (64, 64, 3) and last['z']['histo']['num'] == 4 * 8
    ck = sorted(os.listdir(str(tmp_path / 'ckpt')))
    assert ck == ['checkpoint', 'model-2.npz'], ck
    z = np.load(str(tmp_path / 'ckpt' / 'model-2.npz'))
    assert int(z['global_step']) == 2 and 'd_net/Conv_3/weights' in z.files and 'D_optim/d_net/Conv_3/weights/Adam_1' in z.files
    # resume: the loop restarts at checkpoint counter + 1 with the saved weights, Adam state, kt and global_step
    logs = []
    from t2i_amd.models.wgancls.model import WGanCls
    from t2i_amd.models.wgancls.trainer import WGanClsTrainer
    from t2i_amd.utils.config import config_from_yaml
    cfg = config_from_yaml(path)
    m = WGanCls(cfg)
    tr = WGanClsTrainer(None, m, run.load_dataset(cfg, m.device), cfg)
    tr.train(max_steps=4, log=logs.append, side_effects=True)
    assert any('Load SUCCESS' in l for l in logs) and m.global_step == 3 and m.D_optim.t == 3
