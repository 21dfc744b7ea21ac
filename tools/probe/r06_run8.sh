python __graft_entry__.py smoke 2>&1 | tail -1
python -m pytest tests/test_step_gpu.py tests/test_run.py tests/test_gancls.py tests/test_stackgan.py tests/test_pggan.py tests/test_dp_segments_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^FAILED|^E " | head
python -m pytest tests/test_step_b64_gpu.py -m gpu -q -x -k "B8" 2>&1 | grep -E "passed|failed|^FAILED|^E " | head
bash tools/probe/r06_ab.sh T2I_STORE_FIRST
