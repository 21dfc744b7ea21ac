#!/usr/bin/env python
"""Per-kernel PMC table from one or more rocprofv3 --pmc passes (each pass = one counter group over the same command, as
MI355X_MICROARCH.md prescribes).  usage: pmc_kernel.py PATTERN DIR [DIR ...]
Sums every counter over the dispatches whose kernel name contains PATTERN, grouped by the kernel's template arguments, and prints
per-dispatch averages plus the ratios the guide names (SQ_* in quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES in cycles):
  wait_any / wave_cycles        share of wave time parked in s_waitcnt / barriers
  wait_inst_any / wave_cycles   share stalled at issue (MFMA RAW / pipe busy); wait_inst_lds = its LDS-issue part
  active / wave_cycles          share issuing
  mfma_busy / (4 x busy_cycles) matrix-pipe utilisation
  TCC hit rate                  TCC_HIT / (TCC_HIT + TCC_MISS)"""
import collections
import csv
import glob
import re
import sys


def main():
    pat, dirs = sys.argv[1], sys.argv[2:]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(set))
    dur = collections.defaultdict(float)
    passes = collections.defaultdict(lambda: collections.defaultdict(set))     # a counter listed in two passes is averaged, not summed
    for d in dirs:
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                if pat not in r['Kernel_Name']:
                    continue
                name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('t2i::', '')
                tot[name][r['Counter_Name']] += float(r['Counter_Value'])
                cnt[name][r['Counter_Name']].add(r['Dispatch_Id'])
                passes[name][r['Counter_Name']].add(d)
                if (name, r['Dispatch_Id']) not in seen and 'End_Timestamp' in r:
                    seen.add((name, r['Dispatch_Id']))
                    dur[name] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / max(1, len(dirs))
    for name in sorted(tot):
        c = tot[name]
        n = {k: max(1, len(v)) for k, v in cnt[name].items()}
        avg = {k: c[k] / n[k] / max(1, len(passes[name][k])) for k in c}
        nd = max(n.values())
        print('== %s   (%d dispatches per pass, avg %.1f us under the profiler)' % (name, nd, dur[name] / nd / 1e3))
        for k in sorted(avg):
            print('   %-28s %16.1f per dispatch' % (k, avg[k]))
        g = avg.get
        if g('SQ_WAVE_CYCLES'):
            w = g('SQ_WAVE_CYCLES')
            for k, label in (('SQ_WAIT_ANY', 'wait_any'), ('SQ_WAIT_INST_ANY', 'wait_inst_any'), ('SQ_WAIT_INST_LDS', 'wait_inst_lds'),
                             ('SQ_ACTIVE_INST_ANY', 'active_inst_any')):
                if g(k) is not None:
                    print('   %-28s %16.3f of wave cycles' % (label, g(k) / w))
        if g('SQ_VALU_MFMA_BUSY_CYCLES') and g('SQ_BUSY_CYCLES'):
            print('   %-28s %16.3f (mfma busy / (4 x SQ_BUSY_CYCLES))' % ('mfma_util', g('SQ_VALU_MFMA_BUSY_CYCLES') / (4.0 * g('SQ_BUSY_CYCLES'))))
        if g('SQ_VALU_MFMA_BUSY_CYCLES') and g('GRBM_GUI_ACTIVE'):
            print('   %-28s %16.3f (mfma busy / (1024 SIMDs x GRBM_GUI_ACTIVE))' % ('mfma_util_grbm', g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024.0 * g('GRBM_GUI_ACTIVE'))))
        if g('TCC_HIT_sum') is not None and g('TCC_MISS_sum') is not None and g('TCC_HIT_sum') + g('TCC_MISS_sum') > 0:
            print('   %-28s %16.3f' % ('TCC hit rate', g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum'))))
        if g('SQ_INSTS_LDS') and g('SQ_INSTS_VALU_MFMA_MOPS_F32') :
            pass


if __name__ == '__main__':
    main()
