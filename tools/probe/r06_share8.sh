#!/bin/bash
# N independent single-process soaks (tools/probe/r06_soak.py, same seeds) time-slicing ONE GPU: do they all finish, and with the same bits?
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"; cd "$REPO"; mkdir -p gpurun_out/r06
N=${1:-8}; IT=${2:-40}; B=${3:-64}
pids=()
for i in $(seq 1 $N); do
  SOAK_B=$B SOAK_ITERS=$IT timeout 600 python tools/probe/r06_soak.py > gpurun_out/r06/share_$i.out 2> gpurun_out/r06/share_$i.err &
  pids+=($!)
done
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done
echo "all exited cleanly: $([ $rc = 0 ] && echo yes || echo NO)"
grep -l "ILLEGAL\|fault\|Error" gpurun_out/r06/share_*.err | head
python - $N <<'PY'
import json, sys
n = int(sys.argv[1]); rows = []
for i in range(1, n + 1):
    try:
        rows.append(json.loads(open('gpurun_out/r06/share_%d.out' % i).read().strip().splitlines()[-1]))
    except Exception as e:
        rows.append(None); print('process', i, 'left no result:', e)
ok = [r for r in rows if r]
print('%d of %d finished; finite: %s; bit-identical to process 1: %s' % (len(ok), n, [r['finite'] for r in ok], [r['rows'] == ok[0]['rows'] and r['wnorm'] == ok[0]['wnorm'] for r in ok]))
for r in ok:
    if r['rows'] != ok[0]['rows']:
        first = next(i for i, (a, b) in enumerate(zip(r['rows'], ok[0]['rows'])) if a != b)
        print('  first differing iteration', first + 1, r['rows'][first], ok[0]['rows'][first])
PY
