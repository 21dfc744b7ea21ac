mkdir -p gpurun_out/r06
python __graft_entry__.py smoke > gpurun_out/r06/smoke.log 2>&1; tail -2 gpurun_out/r06/smoke.log
python -m pytest tests/test_step_gpu.py tests/test_run.py tests/test_storage_gpu.py -m gpu -q -x > gpurun_out/r06/t_step.log 2>&1; tail -6 gpurun_out/r06/t_step.log
for v in 0 1; do T2I_PAIR_G=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-rows --instrument off 2>gpurun_out/r06/pair$v.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d.get('config3_bf16', {})
print('PAIR_G=$v  f32 %.3f ms  config3 %s ms  all_bf16 %s' % (d['ms_per_step'], c.get('ms_per_step'), (c.get('all_bf16_side_row') or {}).get('ms_per_step')))"; tail -3 gpurun_out/r06/pair$v.err; done
