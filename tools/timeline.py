#!/usr/bin/env python
"""One iteration of bench.py as the GPU saw it: every dispatch of the LAST complete iteration in a rocprofv3 kernel trace, with
its grid, duration and the idle gap before it.  usage: python tools/timeline.py <rocprof out dir> <iterations in the trace> [top]
Prints totals (busy, idle, span), the per-kernel busy/idle table and the `top` longest dispatches."""
import collections
import csv
import glob
import os
import re
import sys


def main():
    d, iters = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    f = [p for p in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)][0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    n = len(rows) // iters
    it = rows[-n:]                                   # the last iteration (every iteration issues the same launches)
    t0 = int(it[0]['Start_Timestamp'])
    span = int(it[-1]['End_Timestamp']) - t0
    busy = collections.OrderedDict()
    prev_end, idle_total, busy_total = None, 0, 0
    recs = []
    for r in it:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('t2i::', '')
        gap = 0 if prev_end is None else max(0, s - prev_end)
        prev_end = max(prev_end or 0, e)
        grid = int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) // max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 256))))
        b = busy.setdefault(name, [0, 0, 0])
        b[0] += 1; b[1] += e - s; b[2] += gap
        idle_total += gap; busy_total += e - s
        recs.append((e - s, gap, name, grid, s - t0))
    if os.environ.get('T2I_TIMELINE_SEQUENCE'):       # every dispatch in issue order: start us, duration us, gap us, grid, kernel
        with open(os.environ['T2I_TIMELINE_SEQUENCE'], 'w') as fh:
            for dur, gap, name, grid, at in recs:
                fh.write('%9.1f %7.1f %5.1f %6d  %s\n' % (at / 1e3, dur / 1e3, gap / 1e3, grid, name[:90]))
    print('last iteration: %d dispatches, span %.3f ms, kernels %.3f ms, idle gaps %.3f ms (avg %.2f us)' % (
        len(it), span / 1e6, busy_total / 1e6, idle_total / 1e6, idle_total / 1e3 / len(it)))
    short = [r for r in recs if r[0] < 8000]
    print('dispatches shorter than 8 us: %d, %.3f ms in total (VERDICT round 1, item 6: the launch-bound tail)' % (len(short), sum(r[0] for r in short) / 1e6))
    print('%-46s %6s %10s %10s %9s' % ('kernel', 'calls', 'busy us', 'gap-before', 'avg us'))
    for name, (c, b, g) in sorted(busy.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
        print('%-46s %6d %10.1f %10.1f %9.1f' % (name[:46], c, b / 1e3, g / 1e3, b / 1e3 / c))
    print('\nlongest dispatches:')
    for dur, gap, name, grid, at in sorted(recs, reverse=True)[:top]:
        print('  %8.1f us  (gap %5.1f)  at %8.1f us  grid %6d  %s' % (dur / 1e3, gap / 1e3, at / 1e3, grid, name[:60]))
    if os.environ.get('T2I_TIMELINE_ALL'):
        print('\nall dispatches in order:')
        for dur, gap, name, grid, at in recs:
            print('  at %8.1f  %7.1f us  gap %5.1f  grid %6d  %s' % (at / 1e3, dur / 1e3, gap / 1e3, grid, name[:70]))


if __name__ == '__main__':
    main()
