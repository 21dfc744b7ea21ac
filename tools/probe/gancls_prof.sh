#!/bin/bash
# Kernel trace of the gancls iteration (tools/next_rows.py --rows gancls) -> gpurun_out/gcls/timeline_gancls.txt: per-kernel busy time and
# every dispatch of the last iteration.  The number of iterations in the trace = Adam launches / 2.  Run on the GPU box.
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/gcls; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o r -- python $REPO/tools/next_rows.py --rows gancls --budget-s 1.0 > $OUT/kt.log 2>&1
ITERS=$(python - "$OUT/kt" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
print(sum(1 for r in csv.DictReader(open(f)) if 'adam_tf_kernel' in r['Kernel_Name']) // 2)
PY
)
echo "iterations in the trace: $ITERS" > $OUT/timeline_gancls.txt
T2I_TIMELINE_SEQUENCE=$OUT/sequence_gancls.txt python $REPO/tools/timeline.py $OUT/kt $ITERS 70 >> $OUT/timeline_gancls.txt 2>&1
rm -rf $OUT/kt
