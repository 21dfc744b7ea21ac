#!/usr/bin/env python
"""Idle time between consecutive kernels in a rocprofv3 --kernel-trace csv.  usage: gap_analysis.py kernel_trace.csv"""
import csv
import sys

rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))))
# keep the last ~40% of the trace (steady-state graph replays)
rows = rows[int(len(rows) * 0.6):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
big = sorted(((g, rows[i][2][:50], rows[i + 1][2][:50]) for i, g in enumerate(gaps)), reverse=True)[:8]
print('kernels %d  span %.3f ms  busy %.3f ms (%.1f%%)  idle %.3f ms  mean gap %.2f us  median %.2f us' % (
    len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, sum(gaps) / len(gaps) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
for g, a, b in big:
    print('  gap %8.1f us after %-50s before %s' % (g / 1e3, a, b))
small = [r for r in rows if r[1] - r[0] < 10000]
print('kernels shorter than 10 us: %d (%.1f%% of launches), their time %.3f ms' % (len(small), 100.0 * len(small) / len(rows), sum(e - s for s, e, _ in small) / 1e6))
