#!/usr/bin/env python
"""Group a rocprofv3 kernel trace by (kernel, grid): usage trace_groups.py DIR ITERS [TOP] — per-iteration time per launch shape."""
import csv
import glob
import sys
from collections import defaultdict

d, iters = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = (glob.glob(d + '/*kernel_trace.csv') + glob.glob(d + '/*/*kernel_trace.csv'))[0]
g = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('t2i::', '')
    key = (name[:44], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    e = g[key]
    e[0] += 1
    e[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(e[1] for e in g.values())
print('total %.1f us/iter' % (tot / iters))
for k, e in sorted(g.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%-44s grid %7s %5s %4s  calls/iter %5.1f  avg %8.1f us  %8.1f us/iter  %5.2f%%' % (k[0], k[1], k[2], k[3], e[0] / iters, e[1] / e[0], e[1] / iters, 100 * e[1] / tot))
