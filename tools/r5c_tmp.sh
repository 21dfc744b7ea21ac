set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r5l; rm -rf $OUT; mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > $OUT/tests_k.log 2>&1
tail -4 $OUT/tests_k.log
for f in 0 1 0 1; do
T2I_WINO_FUSE=$f python bench.py --no-cpu-baseline --no-side-rows --no-config3 --steps 20 --instrument off > $OUT/bench_$f.json 2>/dev/null; python -c "
import json;d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1]);print('wino_fuse $f fp32', d['value'],d['ms_per_step'])"
done
timeout 900 python -m pytest tests/test_step_b64_gpu.py tests/test_step_gpu.py -m gpu -q -x > $OUT/tests_s.log 2>&1
tail -4 $OUT/tests_s.log
