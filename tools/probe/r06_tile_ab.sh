for t in 11 0 21 12 22 11; do
  T2I_BGEMM_TILE=$t python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-rows --instrument off --no-config3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('T2I_BGEMM_TILE=$t  f32 %.3f ms' % d['ms_per_step'])"
done
for v in 1024 1536 2048; do
  T2I_BGEMM_TILE=0 T2I_BGEMM_BIG_ITEMS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-rows --instrument off --no-config3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('T2I_BGEMM_TILE=0 BIG_ITEMS=$v  f32 %.3f ms' % d['ms_per_step'])"
done
