mkdir -p gpurun_out/r06
python -m pytest tests/test_step_gpu.py -m gpu -x -q > gpurun_out/r06/t_step.log 2>&1; tail -15 gpurun_out/r06/t_step.log
python -m pytest "tests/test_step_b64_gpu.py" -m gpu -x -q -k "B8 or B16" > gpurun_out/r06/t_b64.log 2>&1; tail -15 gpurun_out/r06/t_b64.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-rows > gpurun_out/r06/bench_stackA.json 2> gpurun_out/r06/bench_stackA.err; tail -c 600 gpurun_out/r06/bench_stackA.err
T2I_STACK_XHAT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-rows > gpurun_out/r06/bench_stack0.json 2> gpurun_out/r06/bench_stack0.err
