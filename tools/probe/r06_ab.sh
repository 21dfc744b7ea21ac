# A/B/A/B on ONE box: $1 = env var name, rest = bench args.  Prints ms/step for fp32 and config 3.
VAR=$1; shift
for rep in 1 2; do for v in 0 1; do
  env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-rows --instrument off "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d.get('config3_bf16', {})
print('$VAR=$v  f32 %.3f ms  config3 %s ms  all_bf16 %s' % (d['ms_per_step'], c.get('ms_per_step'), (c.get('all_bf16_side_row') or {}).get('ms_per_step')))"
done; done
