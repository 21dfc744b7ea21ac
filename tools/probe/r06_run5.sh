mkdir -p gpurun_out/r06
python -m pytest tests/test_step_gpu.py tests/test_dp_diffdata_gpu.py -m gpu -q -x > gpurun_out/r06/t_step.log 2>&1; tail -4 gpurun_out/r06/t_step.log
python -m pytest "tests/test_step_b64_gpu.py" -m gpu -x -q -k "B8 or B16" > gpurun_out/r06/t_b64.log 2>&1; tail -3 gpurun_out/r06/t_b64.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-rows > gpurun_out/r06/bench_stackC.json 2> gpurun_out/r06/bench_stackC.err
bash tools/probe/r06_ktrace.sh kt2
