cd $GRAFT_REPO_ROOT
python -m pytest tests/test_step_gpu.py tests/test_step_b64_gpu.py tests/test_storage_gpu.py tests/test_run.py -x -q -m gpu > gpurun_out/run51_tests.log 2>&1; grep -n "passed\|failed\|Fatal" gpurun_out/run51_tests.log | tail -3
for m in f32 bf16; do python bench.py --math $m --no-cpu-baseline --no-config3 --instrument off --min-busy-s 2 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['dtype'], d['value'], d['ms_per_step'])"; done
