cd $GRAFT_REPO_ROOT
echo "##### ping-pong tests"
T2I_FORCE_TILE=42 T2I_BF16_DMA=8 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_storage_gpu.py -q -m gpu -x -k "bf16" 2>&1 | tail -4
for B in 64 192 512; do echo "== pingpong B=$B"; T2I_FORCE_TILE=42 T2I_BF16_DMA=8 python tools/bench_conv.py --math bf16 --storage bf16 --batch $B --reps 10 2>&1 | grep -E "^D2|^D3|^D4|^D10|^G5c|^G7c|^G8c|TOTAL fwd|TOTAL bwd_data"; done
echo "== default B=192"; python tools/bench_conv.py --math bf16 --storage bf16 --batch 192 --reps 10 2>&1 | grep -E "^D2|^D3|^D4|^D10|^G5c|^G7c|^G8c|TOTAL fwd|TOTAL bwd_data"
echo "##### col_reduce sweep"
for w in 768 1536 3072; do for c in 192 512; do echo "colred_wgs=$w cap=$c"; T2I_COLRED_WGS=$w T2I_COLRED_CAP=$c python tools/bench_aux.py 2>&1 | grep -E "col_reduce|act_bwd_colsum|adam"; done; done
