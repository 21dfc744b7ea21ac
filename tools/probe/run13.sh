cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gp_golden" 2>&1 | tail -2
b() { python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 1.5 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['ms_per_step'])"; }
for ovh in 60 90 120 160; do for sp in 29 60 100; do echo -n "ovh=$ovh split_us=$sp: "; T2I_DMA_OVH=$ovh T2I_DMA_SPLIT_US=$sp b; done; done
