"""Checkpoints — the role of the reference's utils/saver.py:6-25 (tf.train.Saver save / restore by global step).

On-disk format: one `<prefix>-<step>.npz` per checkpoint holding every variable of the store under its TF name
(`d_net/Conv_3/weights`, `g_net/BatchNorm_4/moving_mean`, ... — conv kernels HWIO, deconv [kh,kw,Cout,Cin], dense
[in,out], i.e. exactly the key space and layouts of the reference's TF checkpoints, so arrays dumped from a real TF run
load unchanged), plus optimizer slots under `<opt>/<name>/Adam` and `/Adam_1`, the step counts `<opt>/t` and whatever
scalars the trainer registers through `extra` (wgancls: `kt` and `global_step`); and a `checkpoint` text file naming the latest one (what tf.train.get_checkpoint_state reads).  `load`
returns (found, counter) with the counter parsed from the file name like the reference does."""
import os
import re

import numpy as np
import torch


class Saver(object):
    """var_list: name prefixes to include (None = every variable), like tf.train.Saver(var_list)."""

    def __init__(self, store, optimizers=None, extra=None, var_list=None, max_to_keep=5):
        self.store, self.optimizers, self.extra = store, optimizers or {}, extra or {}
        self.var_list, self.max_to_keep = var_list, max_to_keep
        self._kept = []

    def _selected(self):
        for n, v in self.store.vars.items():
            if self.var_list is None or any(n.startswith(p) for p in self.var_list):
                yield n, v

    def state(self):
        out = {n: v.detach().cpu().numpy() for n, v in self._selected()}
        for oname, opt in self.optimizers.items():
            a = opt.arena
            for n in a.names:
                o, k = a.offsets[n]
                out['%s/%s/Adam' % (oname, n)] = opt.m[o:o + k].view(a.vars[n].shape).cpu().numpy()
                out['%s/%s/Adam_1' % (oname, n)] = opt.v[o:o + k].view(a.vars[n].shape).cpu().numpy()
            out['%s/t' % oname] = np.array(opt.t)
        for k, get in self.extra.items():
            out[k] = np.asarray(get[0]())
        return out

    def restore(self, path):
        z = np.load(path)
        with torch.no_grad():
            for n, v in self._selected():
                if n not in z.files:
                    raise KeyError('checkpoint %s has no variable %s' % (path, n))
                if tuple(z[n].shape) != tuple(v.shape):
                    raise ValueError('checkpoint %s: %s has shape %s, variable has %s' % (path, n, z[n].shape, tuple(v.shape)))
                v.copy_(torch.from_numpy(z[n]).to(v.device))
            for oname, opt in self.optimizers.items():
                a = opt.arena
                for n in a.names:
                    o, k = a.offsets[n]
                    if '%s/%s/Adam' % (oname, n) in z.files:
                        opt.m[o:o + k].copy_(torch.from_numpy(z['%s/%s/Adam' % (oname, n)]).reshape(-1).to(opt.m.device))
                        opt.v[o:o + k].copy_(torch.from_numpy(z['%s/%s/Adam_1' % (oname, n)]).reshape(-1).to(opt.v.device))
                if '%s/t' % oname in z.files:
                    opt.t = int(z['%s/t' % oname])
                if hasattr(opt, 'moments_loaded'):
                    opt.moments_loaded()
            for k, get in self.extra.items():
                if k in z.files:
                    get[1](z[k])
        from .. import kernels as K
        K.filter_cache_invalidate()


_CKPT_RE = re.compile(r'^model-(\d+)\.npz$')


def _existing(checkpoint_dir):
    """Checkpoints already in the directory, oldest step first (what max_to_keep prunes against after a resume)."""
    found = []
    if os.path.isdir(checkpoint_dir):
        for f in os.listdir(checkpoint_dir):
            m = _CKPT_RE.match(f)
            if m:
                found.append((int(m.group(1)), os.path.join(checkpoint_dir, f)))
    return [p for _, p in sorted(found)]


def save(saver, sess, checkpoint_dir, step):
    """reference utils/saver.py:6-10 (sess is unused: there is no TF session).  The archive is written to a temporary
    name and renamed into place, so a crash mid-save never leaves a truncated newest checkpoint; the `checkpoint` state
    file is updated only after the archive exists."""
    if not os.path.exists(checkpoint_dir):
        os.makedirs(checkpoint_dir)
    name = 'model-%d.npz' % step
    path = os.path.join(checkpoint_dir, name)
    tmp = path + '.tmp'
    with open(tmp, 'wb') as f:
        np.savez(f, **saver.state())
    os.replace(tmp, path)
    saver._kept = [p for p in _existing(checkpoint_dir) if p != path] + [path]
    while len(saver._kept) > saver.max_to_keep:
        old = saver._kept.pop(0)
        if os.path.exists(old):
            os.remove(old)
    state_tmp = os.path.join(checkpoint_dir, 'checkpoint.tmp')
    with open(state_tmp, 'w') as f:
        f.write('model_checkpoint_path: "%s"\n' % name)
    os.replace(state_tmp, os.path.join(checkpoint_dir, 'checkpoint'))
    return path


def load(saver, sess, checkpoint_dir):
    """reference utils/saver.py:13-25 -> (could_load, counter)"""
    print(' [*] Reading checkpoints from %s...' % checkpoint_dir)
    state = os.path.join(checkpoint_dir, 'checkpoint')
    if os.path.exists(state):
        m = re.search(r'model_checkpoint_path: "([^"]+)"', open(state).read())
        if m and os.path.exists(os.path.join(checkpoint_dir, m.group(1))):
            ckpt_name = os.path.basename(m.group(1))
            saver.restore(os.path.join(checkpoint_dir, ckpt_name))
            saver._kept = _existing(checkpoint_dir)          # pruning continues across the resume
            counter = int(next(re.finditer(r'(\d+)(?!.*\d)', ckpt_name)).group(0))
            print(' [*] Success to read {}'.format(ckpt_name))
            return True, counter
    print(' [*] Failed to find checkpoints')
    return False, 0
