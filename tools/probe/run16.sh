cd $GRAFT_REPO_ROOT
echo "##### full gpu suite"
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8
echo "##### smoke"
python __graft_entry__.py smoke 2>&1 | tail -2
echo "##### bench"
( time python bench.py > gpurun_out/bench_r04_b.json 2> gpurun_out/bench_r04_b.err ) 2>&1 | tail -3
tail -3 gpurun_out/bench_r04_b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04_b.json').read().strip().splitlines()[-1])
print('f32', d['value'], d['ms_per_step'], d['roofline']['frac'])
c=d['config3_bf16']; print('c3', c['value'], c['ms_per_step'], c['roofline']['frac'])
print('b8', d['b8_per_gpu']['f32'].get('images_per_sec'), d['b8_per_gpu']['bf16'].get('images_per_sec'))
for r in d['next_rows']['rows']: print(r.get('row'), r.get('dtype'), r.get('images_per_sec'), r.get('frac_vs_driver_ms'), r.get('error'))
PY
