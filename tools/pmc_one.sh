#!/bin/bash
# usage: tools/pmc_one.sh LAYER BATCH MODE COUNTER [math]  -- one PMC counter over one conv primitive (sum over igemm dispatches / launches)
cd /tmp && export TMPDIR=/tmp
out=/tmp/pmc_$1_$2_$3_$4; rm -rf $out
T2I_ONE_MATH=${5:-f32} timeout 180 rocprofv3 --pmc $4 --kernel-trace --output-format csv -d $out -o r -- python /root/repo/tools/one_conv.py $1 $2 $3 10 > /dev/null 2>&1
python - "$(find $out -name '*counter_collection.csv' | head -1)" "$1 $2 $3 $4 ${5:-f32} group=${T2I_GROUP_N:-8}" <<'PY'
import csv, sys
tot = 0.0; n = set(); dur = 0
for r in csv.DictReader(open(sys.argv[1])):
    if 'igemm_kernel' in r['Kernel_Name'] or 'bgemm_kernel' in r['Kernel_Name']:
        tot += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in n:
            n.add(r['Dispatch_Id']); dur += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print('%s: %.1f per launch (KiB if *_SIZE), %.1f us/launch' % (sys.argv[2], tot / max(len(n), 1), dur / max(len(n), 1) / 1e3))
PY
