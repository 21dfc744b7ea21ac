run() { env "$@" python bench.py --math bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-side-rows --instrument off 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  config3 %.3f ms' % d['ms_per_step'])"; }
run T2I_NOOP=1
run T2I_DMA_SPLIT_US=60
run T2I_DMA_SPLIT_US=100
run T2I_DMA_SPLIT_US=200
run T2I_FORCE_SPLITK=1
run T2I_DMA_OVH=80
run T2I_DMA_OVH=160
run T2I_HFT_OVH=40
run T2I_HFT_OVH=120
run T2I_NOOP=1
