cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_storage_gpu.py -q -m gpu -x 2>&1 | tail -5
