cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu_r04b.log 2>&1; grep -n "passed\|failed" gpurun_out/full_gpu_r04b.log | tail -3
python __graft_entry__.py smoke 2>&1 | tail -1
