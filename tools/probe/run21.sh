cd $GRAFT_REPO_ROOT
echo "##### full gpu suite"
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6
echo "##### bench"
( time python bench.py > gpurun_out/bench_r04_c.json 2> gpurun_out/bench_r04_c.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04_c.json').read().strip().splitlines()[-1])
r=d['roofline']; print('f32', d['value'], d['ms_per_step'], r['frac'], r['mfma_util'], r['traffic_over_compulsory'], r['traffic_source']['counts_agree'])
c=d['config3_bf16']; r=c['roofline']; print('c3', c['value'], c['ms_per_step'], r['frac'], r['mfma_util'], r['traffic_over_compulsory'], r['traffic_source'])
PY
