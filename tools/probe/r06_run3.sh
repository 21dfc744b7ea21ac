mkdir -p gpurun_out/r06
python -m pytest tests/test_step_gpu.py tests/test_dp_exactness_gpu.py tests/test_dp_segments_gpu.py tests/test_dp_diffdata_gpu.py tests/test_storage_gpu.py -m gpu -q > gpurun_out/r06/t_step.log 2>&1; tail -12 gpurun_out/r06/t_step.log
python -m pytest "tests/test_step_b64_gpu.py" -m gpu -x -q -k "B8" > gpurun_out/r06/t_b64.log 2>&1; tail -3 gpurun_out/r06/t_b64.log
