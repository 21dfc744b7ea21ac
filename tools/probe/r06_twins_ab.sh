#!/bin/bash
# Round 6: bf16 twins of the scoped (fp32-forward, bf16-backward) generator's saved activations written by their producers (T2I_SCOPE_TWINS=1, default)
# vs cast in launches of their own (0 = rounds 4-5): config 3 and the compliant bf16 B = 8 row, same box.  Run on the GPU box.
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"; cd "$REPO"; mkdir -p gpurun_out/r06
for t in 1 0 1 0; do
  T2I_SCOPE_TWINS=$t timeout 900 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06/twins_$t.json
  python - $t <<'PY'
import json, sys
d = json.load(open('gpurun_out/r06/twins_%s.json' % sys.argv[1]))
b8 = d.get('b8_per_gpu', {})
print('scope_twins=%s  f32 %.3f ms  config3 %.3f ms  b8 bf16 %s' % (sys.argv[1], d['ms_per_step'], d.get('config3_bf16', {}).get('ms_per_step', 0), b8.get('bf16', {}).get('ms_per_iteration')))
PY
done
