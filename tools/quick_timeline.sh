#!/bin/bash
# One kernel-trace pass of bench.py (5 steps) and the per-dispatch timeline of its last iteration: the quick look between two
# kernel changes (gpurun -- tools/quick_timeline.sh <tag> [--math bf16]).  Output: gpurun_out/<tag>/{timeline.txt,kernel_stats_summary.txt}
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
TAG="${1:-qt}"; shift || true
OUT="$REPO/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace" -o r -- python $REPO/bench.py --no-cpu-baseline --instrument off --repeats 1 --min-busy-s 0 --no-config3 --no-side-rows --steps 5 --warmup 2 "$@" > "$OUT/ktrace.log" 2>&1
python "$REPO/tools/prof_summary.py" "$OUT/ktrace" 9 60 > "$OUT/kernel_stats_summary.txt" 2>&1
python "$REPO/tools/timeline.py" "$OUT/ktrace" 9 30 > "$OUT/timeline.txt" 2>&1
rm -rf "$OUT/ktrace"
head -75 "$OUT/timeline.txt"
