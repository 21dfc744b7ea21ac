python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "trunc_normal" 2>&1 | tail -5
python -m pytest tests/test_step_gpu.py tests/test_run.py tests/test_stackgan.py tests/test_pggan.py tests/test_gancls.py -m gpu -q -x 2>&1 | tail -4
bash tools/probe/r06_ab.sh T2I_NOOP_AB
