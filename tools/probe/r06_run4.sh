mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r06/gputest_full.log 2>&1; tail -32 gpurun_out/r06/gputest_full.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_full.json 2> gpurun_out/r06/bench_full.err
