cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_storage_gpu.py -q -m gpu -x -k "thin or deconv or boundary or golden" 2>&1 | tail -3
for parts in 2 1 4; do for m in f32 bf16; do for B in 64 192; do echo "THIN_PARTS=$parts $m B=$B"; T2I_THIN_PARTS=$parts python tools/bench_conv.py --math $m --batch $B --reps 20 --filter G9dc 2>&1 | grep "^G9dc"; done; done; done
