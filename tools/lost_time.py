#!/usr/bin/env python
"""Rank conv launches of one wgancls iteration by time lost against a target rate (from tools/bench_conv.py logs).
usage: lost_time.py conv_B64.log conv_B192.log [target_TF]"""
import sys

def parse(path):
    out = {}
    for line in open(path):
        p = line.split()
        if len(p) >= 13 and p[0] not in ('layer', 'TOTAL') and '|' in line:
            name = p[0]; gf = float(line.split("|")[0].split()[-1])
            f = line.split('|')
            out[name] = (gf, float(f[1].split()[0]), float(f[2].split()[0]), float(f[3].split()[0]))
    return out

b64, b192 = parse(sys.argv[1]), parse(sys.argv[2])
target = float(sys.argv[3]) if len(sys.argv) > 3 else 125.0
rows = []
for name, (gf, tf, td, tw) in b64.items():
    ideal = gf / target * 1e3   # us
    if name.startswith('D'):
        mult = {'fwd': 3, 'bwdD': 2, 'bwdF': 1}
        if name == 'D1':
            mult = {'fwd': 3, 'bwdD': 2, 'bwdF': 1}
    elif name.endswith('dc'):     # generator deconv: forward IS the bwd_data kernel
        mult = {'fwd': 1, 'bwdD': 2, 'bwdF': 1}
    else:
        mult = {'fwd': 2, 'bwdD': 1, 'bwdF': 1}
    for mode, t in (('fwd', tf), ('bwdD', td), ('bwdF', tw)):
        rows.append((mult[mode] * (t - ideal), name, 64, mode, mult[mode], t, gf / t * 1e-3 * 1e0))
for name, (gf, tf, td, tw) in b192.items():
    ideal = gf / target * 1e3
    for mode, t, m in (('fwd', tf, 1), ('bwdD', td, 0 if name == 'D1' else 1), ('bwdF', tw, 1)):
        if m:
            rows.append((m * (t - ideal), name, 192, mode, m, t, gf / t * 1e-3))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows); ttime = sum(r[4] * r[5] for r in rows)
print('conv time / iteration (from microbench): %.2f ms; lost vs %.0f TF/s: %.2f ms' % (ttime / 1e3, target, tot / 1e3))
for lost, name, B, mode, m, t, tfs in rows[:28]:
    print('%-6s B=%-3d %-5s x%d  %7.1f us  %6.1f TF/s  lost %7.1f us' % (name, B, mode, m, t, tfs * 1e3, lost))
