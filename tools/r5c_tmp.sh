set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r5i; rm -rf $OUT; mkdir -p $OUT
cd $REPO
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > $OUT/tests_k.log 2>&1
tail -8 $OUT/tests_k.log
python -m pytest tests/test_step_gpu.py tests/test_step_b64_gpu.py tests/test_gancls.py tests/test_storage_gpu.py -m gpu -q -x > $OUT/tests.log 2>&1
tail -8 $OUT/tests.log
python tools/next_rows.py --rows gancls stage1 wgancls_b8 --budget-s 1.5 2>&1 | grep -v amdgpu | tee $OUT/rows_after.txt
python bench.py --no-cpu-baseline --no-side-rows --no-config3 --steps 20 > $OUT/bench_f32.json 2>/dev/null; python -c "
import json;d=json.loads(open('$OUT/bench_f32.json').read().strip().splitlines()[-1]);print('fp32', d['value'],d['ms_per_step'])"
T2I_BN_ONE_ENTRY=0 python bench.py --no-cpu-baseline --no-side-rows --no-config3 --steps 20 > $OUT/bench_f32_old.json 2>/dev/null; python -c "
import json;d=json.loads(open('$OUT/bench_f32_old.json').read().strip().splitlines()[-1]);print('fp32 old BN path', d['value'],d['ms_per_step'])"
