python -m pytest tests/test_step_b64_gpu.py tests/test_storage_gpu.py -m gpu -q -x -k "B8 or storage or bf16" 2>&1 | grep -E "passed|failed|^FAILED|^E " | head
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/kt3 -o r -- python $R/bench.py --no-cpu-baseline --instrument off --repeats 1 --min-busy-s 0 --no-config3 --no-side-rows --math bf16 --steps 5 --warmup 2 > $R/gpurun_out/r06/kt3.log 2>&1
python $R/tools/timeline.py $R/gpurun_out/r06/kt3 9 10 | head -4
python $R/tools/timeline.py $R/gpurun_out/r06/kt3 9 10 | grep -E "cast_bf16|splitk_reduce"
rm -rf $R/gpurun_out/r06/kt3
cd $R
for i in 1 2; do python bench.py --math bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-side-rows --instrument off 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config3 %.3f ms' % d['ms_per_step'])"; done
