#!/usr/bin/env python
"""Summarise rocprofv3 --pmc runs of bench.py (one pass per counter group, as MI355X_MICROARCH.md prescribes) for the
implicit-GEMM kernel class.  usage: pmc_summary.py FETCH_DIR WRITE_DIR MFMA_DIR ITERS OUT.json [f32|bf16]"""
import collections
import csv
import glob
import json
import sys


def agg(d):
    f = (glob.glob(d + '/*/*counter_collection.csv') + glob.glob(d + '/*counter_collection.csv'))[0]
    tot = collections.defaultdict(float)
    seen, dur = set(), 0
    for r in csv.DictReader(open(f)):
        if not any(k in r['Kernel_Name'] for k in ('igemm_kernel', 'bgemm_kernel', 'bgemm9_kernel', 'igemm_h_kernel', 'igemm_hd_kernel', 'igemm_hft_kernel', 'igemm_h_filter_kernel', 'igemm_pair_kernel', 'igemm_hd8_kernel')):
            continue
        tot[r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            dur += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    return dict(tot), len(seen), dur


fetch, nf, _ = agg(sys.argv[1])
write, nw, _ = agg(sys.argv[2])
mfma, nm, dur = agg(sys.argv[3])
iters = float(sys.argv[4])
# FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests
# of wide (16 B/lane) reads at 64 B, i.e. reads are under-counted 2x -> doubled here; WRITE_SIZE is uncalibrated (as is).
fetch_b = fetch['FETCH_SIZE'] * 1024 * 2
write_b = write['WRITE_SIZE'] * 1024
cycles = mfma['GRBM_GUI_ACTIVE'] / 8.0                    # summed over the 8 XCDs
out = {
    'kernel': 't2i::igemm_kernel<*> + bgemm_kernel<*> + bgemm9_kernel<*> + igemm_h_kernel<*> + igemm_hd_kernel<*> + igemm_hft_kernel + igemm_h_filter_kernel<*> + igemm_pair_kernel<*>', 'math': (sys.argv[6] if len(sys.argv) > 6 else 'f32'), 'iterations_profiled': iters, 'launches_per_iteration': nm / iters,
    'hbm_side_read_bytes_per_launch': fetch_b / nf, 'hbm_side_write_bytes_per_launch': write_b / nw,
    'traffic_bytes_per_launch': fetch_b / nf + write_b / nw,
    'traffic_bytes_per_iteration': (fetch_b + write_b) / iters,
    'fetch_size_raw_kib': fetch['FETCH_SIZE'], 'write_size_raw_kib': write['WRITE_SIZE'],
    'igemm_ms_per_iteration_profiled': dur / iters / 1e6,
    'effective_clock_ghz': cycles / dur,
    'mfma_busy_cycles': mfma['SQ_VALU_MFMA_BUSY_CYCLES'],
    'mfma_util': mfma['SQ_VALU_MFMA_BUSY_CYCLES'] / (cycles * 1024.0),     # busy cycles / (cycles x 256 CU x 4 SIMD)
    'hbm_gbps_during_igemm': (fetch_b + write_b) / (dur * 1e-9) / 1e9 * (nm / float(nf)),
    'sq_wait_any_frac': mfma['SQ_WAIT_ANY'] / mfma['SQ_WAVE_CYCLES'],
    'sq_wait_inst_any_frac': mfma['SQ_WAIT_INST_ANY'] / mfma['SQ_WAVE_CYCLES'],
    'sq_active_inst_any_frac': mfma['SQ_ACTIVE_INST_ANY'] / mfma['SQ_WAVE_CYCLES'],
}
json.dump(out, open(sys.argv[5], 'w'), indent=1)
print(json.dumps(out, indent=1))
