cd $GRAFT_REPO_ROOT
python -m pytest tests/test_step_gpu.py tests/test_run.py tests/test_storage_gpu.py -x -q -m gpu > gpurun_out/run53_tests.log 2>&1; grep -n "passed\|failed\|Fatal" gpurun_out/run53_tests.log | tail -3; grep -n "Error\|assert" gpurun_out/run53_tests.log | head -5
for v in 0 1 0 1; do for m in f32 bf16; do echo -n "NOISE_IN_GRAPH=$v $m: "; T2I_NOISE_IN_GRAPH=$v python bench.py --math $m --no-cpu-baseline --no-config3 --instrument off --min-busy-s 2 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"; done; done
