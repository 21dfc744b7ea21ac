#!/bin/bash
# Weak-scaling table on ONE node: bench.py at N = 1, 2, 4, 8 GPUs (B = 64 per GPU), each run with the data-parallel preflight
# (N ranks on identical data must reproduce the single replica; bench.py aborts the run otherwise) and the replica sync check.
#   usage: tools/scale_8gpu.sh [steps] [warmup]        env: NS="1 2 4 8", T2I_DP_GRAPHS=0 (eager bucket overlap instead of graph segments)
# Prints one table: N | fp32 img/s | ms/step | efficiency vs N x (N = 1) | config-3 (bf16) img/s | ms/step | efficiency | preflight.
# Never run by the build itself (no multi-GPU box there): this is what a maintainer runs on first contact with an 8-GPU node.
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
STEPS="${1:-20}"; WARM="${2:-5}"; NS="${NS:-1 2 4 8}"
OUT="${OUT:-$REPO/gpurun_out/scale}"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 T2I_CHECK_SYNC=1
cd "$REPO"
for n in $NS; do
  port=$((29500 + n))
  if [ "$n" = 1 ]; then
    timeout 900 python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline > "$OUT/n$n.out" 2> "$OUT/n$n.err"
  else
    timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" \
        bench.py --gpus "$n" --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline > "$OUT/n$n.out" 2> "$OUT/n$n.err"
  fi
  echo "N=$n rc=$?" >&2
  grep '"metric"' "$OUT/n$n.out" | tail -1 > "$OUT/n$n.json"
done
python - "$OUT" $NS <<'PY'
import json, sys
out, ns = sys.argv[1], [int(x) for x in sys.argv[2:]]
rows, base = [], {}
for n in ns:
    try:
        d = json.loads(open('%s/n%d.json' % (out, n)).read())
    except Exception as e:
        rows.append((n, None, 'no JSON line (%s): see %s/n%d.err' % (type(e).__name__, out, n)))
        continue
    rows.append((n, d, ''))
    if n == 1:
        base = {'f32': d['value'], 'bf16': (d.get('config3_bf16') or {}).get('value')}
print('%3s | %10s %8s %6s | %10s %8s %6s | %s' % ('N', 'fp32 img/s', 'ms/step', 'eff', 'bf16 img/s', 'ms/step', 'eff', 'preflight (fp32 / bf16 buckets)'))
for n, d, note in rows:
    if d is None:
        print('%3d | %s' % (n, note)); continue
    c3 = d.get('config3_bf16') or {}
    e1 = d['value'] / (n * base['f32']) if base.get('f32') else float('nan')
    e3 = c3.get('value', float('nan')) / (n * base['bf16']) if base.get('bf16') else float('nan')
    pf = lambda p: 'n/a' if not p else ('%s%s' % ('ok' if p.get('ok') else 'OUTSIDE ITS BOUNDS', ' (exact)' if p.get('exact') else ' (gradients within %.1e / %.1e of the single replica\'s)' % (
        (p.get('max_gradient_diff_rel') or {}).get('critic', -1), (p.get('max_gradient_diff_rel') or {}).get('generator', -1))))
    print('%3d | %10.1f %8.3f %6.3f | %10.1f %8.3f %6.3f | %s / %s' % (n, d['value'], d['ms_per_step'], e1, c3.get('value', float('nan')),
          c3.get('ms_per_step', float('nan')), e3, pf(d.get('dp_preflight')), pf(c3.get('dp_preflight'))))
PY
