set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r5j; rm -rf $OUT; mkdir -p $OUT
cd $REPO
T2I_BGEMM_TILE=41 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "wino or conv or every_tile" > $OUT/tests_41.log 2>&1
tail -5 $OUT/tests_41.log
for t in 11 41; do
T2I_BGEMM_TILE=$t python bench.py --no-cpu-baseline --no-side-rows --no-config3 --steps 20 --instrument off > $OUT/bench_$t.json 2>/dev/null; python -c "
import json;d=json.loads(open('$OUT/bench_$t.json').read().strip().splitlines()[-1]);print('tile $t fp32', d['value'],d['ms_per_step'], d['timing']['ms_per_step_by_region'][:4])"
done
for t in 11 41; do echo "== T2I_BGEMM_TILE=$t"; T2I_BGEMM_TILE=$t timeout 300 python tools/bench_conv.py --batch 192 --filter D 2>&1 | grep -v amdgpu | head -40; done > $OUT/conv_B192.txt
for t in 11 41; do echo "== T2I_BGEMM_TILE=$t"; T2I_BGEMM_TILE=$t timeout 300 python tools/bench_conv.py --batch 64 2>&1 | grep -v amdgpu | head -60; done > $OUT/conv_B64.txt
