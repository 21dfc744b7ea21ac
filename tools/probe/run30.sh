cd $GRAFT_REPO_ROOT
python tools/probe/loop_ablation.py default y4
python -m pytest tests/test_storage_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/run30_tests.log 2>&1; grep -n "passed\|failed" gpurun_out/run30_tests.log | tail -3
for w in 4 8 4 8; do echo -n "bench bf16 waves=$w: "; T2I_BF16_WAVES=$w python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 2 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"; done
