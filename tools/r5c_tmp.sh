set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r5f; mkdir -p $OUT
cd $REPO
python -m pytest tests/test_stackgan.py "tests/test_fullsize_gpu.py" tests/test_dp_segments_gpu.py -m gpu -q -x > $OUT/tests3.log 2>&1
tail -12 $OUT/tests3.log
python tools/next_rows.py --rows gancls stage1 stage2 --budget-s 1.5 2>&1 | grep -v amdgpu | tee $OUT/rows_after.txt
