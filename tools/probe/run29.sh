cd $GRAFT_REPO_ROOT
echo "waves 4:"; python tools/probe/loop_ablation.py default
echo "waves 8:"; T2I_BF16_WAVES=8 python tools/probe/loop_ablation.py default
T2I_BF16_WAVES=8 python -m pytest tests/test_storage_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "bf16 or storage or pair or epilogue" > gpurun_out/run29_tests.log 2>&1; grep -n "passed\|failed" gpurun_out/run29_tests.log | tail -3
for w in 4 8 4 8; do echo -n "bench bf16 waves=$w: "; T2I_BF16_WAVES=$w python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 2 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"; done
