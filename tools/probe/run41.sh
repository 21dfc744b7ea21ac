cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu_r04c.log 2>&1; grep -n "passed\|failed" gpurun_out/full_gpu_r04c.log | tail -3
python __graft_entry__.py smoke 2>&1 | tail -1
( time python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_final.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config3_bf16']['value'], d['config3_bf16']['ms_per_step'], d['cpu_baseline']['value'])
PY
