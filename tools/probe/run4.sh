cd $GRAFT_REPO_ROOT
echo "##### bf16 kernel tests, vec epilogue"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_storage_gpu.py -x -q -m gpu -k "bf16" 2>&1 | tail -5
echo "##### fp32 winograd tests per bgemm tile"
for t in 22 21 12; do T2I_BGEMM_TILE=$t timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "winograd or wino" 2>&1 | tail -3; done
echo "##### fixed cost"
for v in 0 1; do echo "VEC_EPI=$v"; T2I_VEC_EPI=$v python tools/probe/gemm_fixed.py 2>&1 | grep -v amdgpu.ids; done
echo "##### bench_conv bf16 B=64"
for v in 0 1; do echo "== VEC_EPI=$v"; T2I_VEC_EPI=$v python tools/bench_conv.py --math bf16 --batch 64 --reps 10 2>&1 | grep -v amdgpu.ids; done
echo "== VEC_EPI=1 DMA=2"; T2I_BF16_DMA=2 python tools/bench_conv.py --math bf16 --batch 64 --reps 10 2>&1 | grep -E "TOTAL"
echo "##### bench_conv fp32 bgemm tiles"
for t in 11 0 22 21 12; do for b in 64 192; do echo "== BGEMM_TILE=$t B=$b"; T2I_BGEMM_TILE=$t python tools/bench_conv.py --batch $b --reps 10 --cache $( [ $b = 192 ] && echo --filter D ) 2>&1 | grep -v amdgpu.ids; done; done
