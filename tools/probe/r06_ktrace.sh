#!/bin/bash
# quick kernel trace of the fp32 iteration and of config 3 (summary + the last iteration's dispatch sequence)
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
TAG="${1:-kt}"
OUT="$REPO/gpurun_out/r06/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --instrument off --repeats 1 --min-busy-s 0 --no-config3 --no-side-rows"
for m in f32 bf16; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace_$m" -o r -- $BENCH --math $m --steps 5 --warmup 2 > "$OUT/ktrace_$m.log" 2>&1
  python "$REPO/tools/prof_summary.py" "$OUT/ktrace_$m" 9 60 > "$OUT/kernel_stats_summary_$m.txt" 2>&1
  python "$REPO/tools/timeline.py" "$OUT/ktrace_$m" 9 30 > "$OUT/timeline_$m.txt" 2>&1
  rm -rf "$OUT/ktrace_$m"
done
