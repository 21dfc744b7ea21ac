cd $GRAFT_REPO_ROOT
echo -n "w8 plain : "; python tools/probe/loop_ablation.py default
echo -n "w8 paired: "; T2I_BF16_PAIR_TILES=256 python tools/probe/loop_ablation.py default
echo -n "w4 paired: "; T2I_BF16_WAVES=4 T2I_BF16_PAIR_TILES=256 python tools/probe/loop_ablation.py default
T2I_BF16_PAIR_TILES=256 python -m pytest tests/test_storage_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/run35_tests.log 2>&1; grep -n "passed\|failed" gpurun_out/run35_tests.log | tail -3
T2I_BF16_WAVES=4 T2I_BF16_PAIR_TILES=256 python -m pytest tests/test_storage_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "bf16 or storage or pair" > gpurun_out/run35_tests4.log 2>&1; grep -n "passed\|failed" gpurun_out/run35_tests4.log | tail -3
b() { python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 2 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
echo -n "plain        : "; b
echo -n "paired <=256 : "; T2I_BF16_PAIR_TILES=256 b
echo -n "paired <=512 : "; T2I_BF16_PAIR_TILES=512 b
done
