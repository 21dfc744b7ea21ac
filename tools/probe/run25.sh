cd $GRAFT_REPO_ROOT
python -m pytest tests/test_step_gpu.py tests/test_storage_gpu.py tests/test_run.py tests/test_step_b64_gpu.py -x -q -m gpu > gpurun_out/run25_tests.log 2>&1
grep -n "passed\|failed\|Error" gpurun_out/run25_tests.log | tail -5
