run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-rows --instrument off --no-config3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  f32 %.3f ms' % d['ms_per_step'])"; }
run T2I_BGEMM_TILE=11
for v in 1536 2048 2560 3072 4096 6144; do run T2I_BGEMM_TILE=0 T2I_BGEMM_BIG_ITEMS=$v; done
run T2I_BGEMM_TILE=11
