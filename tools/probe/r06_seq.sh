#!/bin/bash
# dispatch sequence of the last replayed iteration (fp32 and config 3): gpurun_out/r06/seq_{f32,bf16}.txt
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
OUT="$REPO/gpurun_out/r06"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --instrument off --repeats 1 --min-busy-s 0 --no-config3 --no-side-rows"
for m in f32 bf16; do
  rm -rf "$OUT/kt_$m"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$m" -o r -- $BENCH --math $m --steps 5 --warmup 2 > "$OUT/kt_$m.log" 2>&1
  T2I_TIMELINE_SEQUENCE="$OUT/seq_$m.txt" python "$REPO/tools/timeline.py" "$OUT/kt_$m" 9 5 > /dev/null 2>&1
  rm -rf "$OUT/kt_$m"
done
