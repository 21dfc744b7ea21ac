cd $GRAFT_REPO_ROOT
python __graft_entry__.py smoke 2>&1 | tail -2
for B in 64 512; do echo "== storage bf16 B=$B"; python tools/bench_conv.py --math bf16 --storage bf16 --batch $B --reps 10 2>&1 | grep -v amdgpu.ids; done
echo "== storage bf16 B=512 FORCE_TILE=42"; T2I_FORCE_TILE=42 python tools/bench_conv.py --math bf16 --storage bf16 --batch 512 --reps 10 2>&1 | grep -E "^D2|^D3|^D4|^D10|^G5c|^G7c|^G8c|TOTAL"
