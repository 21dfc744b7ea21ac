cd $GRAFT_REPO_ROOT
echo "##### smoke via script"
python __graft_entry__.py smoke 2>&1 | tail -15
echo "##### smoke via import"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "##### 8-wave tile: tests with force_tile 42"
T2I_FORCE_TILE=42 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_storage_gpu.py -q -m gpu -x -k "bf16" 2>&1 | tail -4
echo "##### 8-wave tile timing"
for t in 0 22 42; do for B in 64 192 512; do echo "== FORCE_TILE=$t B=$B"; T2I_FORCE_TILE=$t python tools/bench_conv.py --math bf16 --batch $B --reps 10 $( [ $B != 64 ] && echo --filter D ) 2>&1 | grep -E "^D2|^D3|^D4|^D10|^G7c|^G8c|TOTAL fwd|TOTAL bwd_data"; done; done
