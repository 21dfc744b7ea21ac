cd $GRAFT_REPO_ROOT
echo "##### kernel tests (all)"
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_storage_gpu.py -x -q -m gpu 2>&1 | tail -5
for d in 3 4; do echo "##### bf16 kernel tests DMA=$d"; T2I_BF16_DMA=$d timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_storage_gpu.py -x -q -m gpu -k "bf16" 2>&1 | tail -3; done
echo "##### fixed cost / K sweep per DMA variant"
for d in 1 2 3 4; do echo "DMA=$d"; T2I_BF16_DMA=$d python tools/probe/gemm_fixed.py 2>&1 | grep -v amdgpu.ids; done
echo "##### fp32 bench VEC_EPI 0/1"
for v in 0 1; do echo "== VEC_EPI=$v fp32 bench"; T2I_VEC_EPI=$v python bench.py --no-cpu-baseline --no-config3 --instrument off --min-busy-s 1.5 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"; done
echo "##### bf16 bench"
python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 1.5 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"
echo "##### bf16 bench side stream"
python bench.py --math bf16 --side-stream 1 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 1.5 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"
