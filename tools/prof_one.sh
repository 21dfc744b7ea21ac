#!/bin/bash
# usage: tools/prof_one.sh LAYER BATCH MODE [reps]   -- rocprofv3 kernel-trace of one conv primitive; prints per-kernel avg
cd /tmp && export TMPDIR=/tmp
out=/tmp/p_$1_$2_$3; rm -rf $out
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python /root/repo/tools/one_conv.py $1 $2 $3 ${4:-20} > /dev/null 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" "$1 $2 $3" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print('==', sys.argv[2])
for r in rows:
    if 't2i::' in r['Name']:
        print('  %-60s calls %4s avg %8.1f us min %8.1f' % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
