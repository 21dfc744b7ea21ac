cd $GRAFT_REPO_ROOT
for dma in 1 2; do echo "#### T2I_BF16_DMA=$dma"; T2I_BF16_DMA=$dma python tools/probe/gemm_sweep.py 2>&1 | grep -v amdgpu.ids; done
