#!/usr/bin/env python
"""Where config 3's error comes from: per-layer relative L2 of the bf16 generator / critic forward against the float64 oracle at
B = 64 (mask-pinned, tests/test_step_b64_gpu.py's set-up), and the generator-step gradient errors, for a few placements of fp32
arithmetic.  usage: python tools/bf16_error_table.py [--variants base last3_f32 ...]   (GPU + ~1 min of CPU oracle per variant)

Variants (which conv descriptors are forced to fp32 math while the rest of the step runs in bf16 math + bf16 storage):
  base        none, bf16 activation tensors (config 3 as benchmarked)
  base_f32store   none, fp32 activation tensors (bf16 operand rounding only)
  hw32_f32    every conv / deconv whose larger map side is >= 32 (the generator's last three GEMM layers, the critic's first two)
  hw16_f32    ... >= 16
  g_f32       the generator's layers in fp32 math on fp32 tensors, the critic in bf16 math on bf16 tensors
  g_fwd_f32   ... only the generator's FORWARD GEMMs in fp32 math; its input- and filter-gradient GEMMs in bf16 math (fp32 tensors)
  f32         all (the fp32 path: the floor of the comparison)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def rel_l2(got, ref):
    got = got.detach().double().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    ref = ref.detach().double().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref, np.float64)
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))


G_POINTS = ['mu head (lrelu)', 'log-sigma head (lrelu)', 'G4a 1x1 (bn, relu)', 'G4b 3x3 (bn, relu)', 'G4 join 4x4x1024 (relu)', 'G6a 1x1 8x8 (bn, relu)',
            'G6b 3x3 (bn, relu)', 'G6 join 8x8x512 (relu)', 'G7 16x16x256 (bn, relu)', 'G8 32x32x128 (bn, relu)']
D_POINTS = ['D1 32x32x128', 'D2 16x16x256', 'D3 8x8x512', 'D5 1x1 4x4x256', 'D6 3x3 4x4x512', 'D join 4x4x1024', 'text code 128', 'D10 3x3 4x4x1024',
            'D11 1x1 4x4x1024']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--variants', nargs='*', default=['base', 'base_f32store', 'hw32_f32', 'hw16_f32'])
    a = ap.parse_args()
    import bench
    import t2i_amd  # noqa: F401
    from branches import record_branches
    from oracle import torch_step as T
    from t2i_amd import kernels as K
    from t2i_amd.models.wgancls import model as _wm
    from t2i_amd.models.wgancls.model import WGanCls
    from test_step_b64_gpu import N_D, N_G, _split_d_masks, _to_oracle_layout
    _wm._STACK_XHAT = False          # the per-layer value tables below index the two-pass launch order ([G | x | x_mis], then x_hat)
    B = 64
    dev = torch.device('cuda')
    ocfg = T.Cfg(batch=B)
    P = {n: v.double() for n, v in T.init_variables(ocfg, seed=0).items()}
    feed = {k: v.double() for k, v in T.synthetic_feed(ocfg, seed=1).items()}
    m = WGanCls(bench.make_cfg(B), device=dev)
    m.store.load({n: v.numpy() for n, v in P.items()})
    f = {k: v.float().to(dev) for k, v in feed.items()}
    f['epsilon'] = f.pop('eps'); f['learning_rate_d'] = 1e-4; f['learning_rate_g'] = 1e-4
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    real_desc, real_ddesc = K.conv_desc, K.deconv_desc

    def force(pred):
        def cd(Bn, H, W, Cin, Cout, KH, KW, SH, SW, padding, math=None):
            return real_desc(Bn, H, W, Cin, Cout, KH, KW, SH, SW, padding, math=(K.MATH_F32 if pred(max(H, W) ) else math))

        def dd(Bn, H, W, Cin, Cout, KH, KW, SH, SW, padding, math=None):
            return real_ddesc(Bn, H, W, Cin, Cout, KH, KW, SH, SW, padding, math=(K.MATH_F32 if pred(max(H, W) * max(SH, SW)) else math))
        K.conv_desc, K.deconv_desc = cd, dd

    for var in a.variants:
        K.set_math('f32' if var == 'f32' else 'bf16')
        K.set_storage('bf16' if var in ('base', 'g_f32', 'g_fwd_f32') else 'f32')
        m.net_math = {'g_net': ('f32', 'f32')} if var == 'g_f32' else {'g_net': ('f32', 'f32', 'bf16')} if var == 'g_fwd_f32' else {}      # mixed-math variants keep fp32 tensors (a fp32 conv cannot read a bf16 tensor)
        if var == 'hw32_f32':
            force(lambda s: s >= 32)
        elif var == 'hw16_f32':
            force(lambda s: s >= 16)
        try:
            vals = []
            import contextlib

            @contextlib.contextmanager
            def record_values(out):          # the same taps as record_branches, keeping the activation outputs
                saved = {}

                def wrap(name, act_pos, first=False):
                    fn = getattr(K, name)
                    saved[name] = fn

                    def tapped(*aa, **kw):
                        res = fn(*aa, **kw)
                        y = res[0] if first else res
                        act = aa[act_pos] if len(aa) > act_pos else kw.get('act', K.ACT_NONE)
                        if act in (K.ACT_LRELU, K.ACT_RELU):
                            out.append(y.detach().float().cpu())
                        return res
                    setattr(K, name, tapped)
                try:
                    wrap('conv_fwd', 5); wrap('conv_fwd_stats', 5); wrap('conv_bwd_data', 5)
                    wrap('bn_train_fwd_grouped', 6, first=True); wrap('bn_apply', 3); wrap('add_act', 2); wrap('act_fwd', 1)
                    yield out
                finally:
                    for n, fn in saved.items():
                        setattr(K, n, fn)
            rec = []
            with record_values(vals), record_branches(rec):
                d = m.d_losses(f)
                torch.cuda.synchronize()
            masks = _split_d_masks(rec, B)
            tapes = {k: T.MaskTape(masks[k], keep_values=True) for k in masks}
            with torch.no_grad():
                G, _, _ = T.generator(P, ocfg, feed['z'], feed['cond'], feed['ca_noise_d'], train=True, tape=tapes['G'])
                T.discriminator(P, ocfg, feed['x'], feed['cond'], tapes['Dx'])
                T.discriminator(P, ocfg, feed['eps'] * G + (1.0 - feed['eps']) * feed['x'], feed['cond'], tapes['Dxh'])
            print('== variant %s: forward, relative L2 of the activation outputs against float64 (mask-pinned)' % var)
            gv = [_to_oracle_layout(v) for v in vals[:N_G]]
            for i, name in enumerate(G_POINTS):
                print('  G  %-32s %.3e' % (name, rel_l2(gv[i], tapes['G'].values[i])))
            print('  G  %-32s %.3e' % ('tanh output 64x64x3', rel_l2(d['G'], G)))
            d3 = [_to_oracle_layout(v) for v in vals[N_G:N_G + N_D]]
            for i, name in enumerate(D_POINTS):
                print('  D(x)      %-25s %.3e' % (name, rel_l2(d3[i][B:2 * B], tapes['Dx'].values[i])))
            dh = [_to_oracle_layout(v) for v in vals[N_G + N_D:]]
            for i, name in enumerate(D_POINTS):
                print('  D(x_hat)  %-25s %.3e' % (name, rel_l2(dh[i], tapes['Dxh'].values[i])))
            ref = T.d_step(P, ocfg, feed, 0.7, masks=masks)
            print('  D(x_hat) logit %.3e   grad_x_hat %.3e' % (rel_l2(d['Dx_hat_logit'], ref['Dx_hat']), rel_l2(d['grad_x_hat'], ref['grad_x_hat'])))
            worst = max((rel_l2(m.d_arena.grad_of(n), ref['grads'][n]), n) for n in m.d_vars if float(ref['grads'][n].abs().max()) > 1e-9 and 'Conv_9/biases' not in n)
            print('  critic-step gradients: worst relative L2 %.3e (%s)' % worst)
            rec = []
            with record_branches(rec):
                g = m.g_losses(f)
                torch.cuda.synchronize()
            rec = [_to_oracle_layout(x) for x in rec]
            gref = T.g_step(P, ocfg, feed, masks={'G': rec[:N_G], 'Dg': rec[N_G:]})
            print('  generator step: G %.3e ; gradients, relative L2 per tensor:' % rel_l2(g['G'], gref['G']))
            errs = []
            for n in m.g_vars:
                r = gref['grads'][n]
                if float(r.abs().max()) < 1e-9:
                    continue
                errs.append((rel_l2(m.g_arena.grad_of(n), r), n))
            for e, n in errs:
                print('     %-44s %.3e' % (n, e))
            print('  generator-step gradients: worst %.3e, median %.3e' % (max(errs)[0], sorted(e for e, _ in errs)[len(errs) // 2]))
        finally:
            K.conv_desc, K.deconv_desc = real_desc, real_ddesc
            K.set_storage('f32')
            K.set_math('f32')


if __name__ == '__main__':
    main()
