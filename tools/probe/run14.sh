cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_storage_gpu.py -q -m gpu -x -k "thin or deconv or boundary or golden" 2>&1 | tail -3
for b2 in 0 32 64; do for m in f32 bf16; do for B in 64 192; do echo "THIN_B2=$b2 $m B=$B"; T2I_THIN_B2=$b2 python tools/bench_conv.py --math $m --batch $B --reps 20 --filter G9dc 2>&1 | grep "^G9dc"; done; done; done
for b2 in 0 32; do echo "bench THIN_B2=$b2"; for m in f32 bf16; do T2I_THIN_B2=$b2 python bench.py --math $m --no-cpu-baseline --no-config3 --instrument off --min-busy-s 1.5 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['dtype'], d['value'], d['ms_per_step'])"; done; done
