#!/usr/bin/env python
"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats (csv) run.  usage: prof_summary.py DIR ITERS [TOP]"""
import csv
import glob
import sys

d, iters = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = (glob.glob(d + '/*/*kernel_stats.csv') + glob.glob(d + '/*kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
ig = sum(float(r['TotalDurationNs']) for r in rows if any(k in r['Name'] for k in ('igemm_kernel', 'bgemm_kernel', 'bgemm9_kernel', 'igemm_h_kernel', 'igemm_hd_kernel', 'igemm_hft_kernel', 'igemm_h_filter_kernel', 'igemm_pair_kernel', 'igemm_hd8_kernel')))
print('kernel time: %.3f ms/iter ; GEMM kernels (igemm_kernel, bgemm_kernel, bgemm9_kernel, igemm_h_kernel, igemm_hd_kernel, igemm_hft_kernel, igemm_h_filter_kernel, igemm_pair_kernel): %.3f ms/iter (%.1f%%)' % (tot / iters / 1e6, ig / iters / 1e6, 100 * ig / tot))
for r in rows[:top]:
    print('%-84s %5s calls %9.1f us/iter %6.2f%% avg %8.1f us' % (r['Name'][:84], r['Calls'], float(r['TotalDurationNs']) / iters / 1e3,
                                                                float(r['Percentage']), float(r['AverageNs']) / 1e3))
