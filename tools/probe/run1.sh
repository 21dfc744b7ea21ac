set -x
cd $GRAFT_REPO_ROOT
export T2I_BF16_DMA=2
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_storage_gpu.py -x -q -m gpu -k "bf16" 2>&1 | tail -5
for dma in 1 2; do
  echo "== DMA=$dma B=64"; T2I_BF16_DMA=$dma python tools/bench_conv.py --math bf16 --batch 64 --reps 10 2>&1 | grep -v amdgpu.ids
done
for dma in 1 2; do
  echo "== DMA=$dma B=192"; T2I_BF16_DMA=$dma python tools/bench_conv.py --math bf16 --batch 192 --reps 10 --filter D 2>&1 | grep -v amdgpu.ids
done
