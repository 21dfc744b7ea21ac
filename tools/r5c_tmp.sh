set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r5g; rm -rf $OUT; mkdir -p $OUT
cd $REPO
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "grouped or batch_norm or sigmoid" > $OUT/tests_k.log 2>&1
tail -8 $OUT/tests_k.log
python -m pytest tests/test_gancls.py tests/test_stackgan.py "tests/test_fullsize_gpu.py" tests/test_dp_segments_gpu.py tests/test_storage_gpu.py -m gpu -q -x > $OUT/tests.log 2>&1
tail -8 $OUT/tests.log
python tools/next_rows.py --rows gancls stage1 stage2 --budget-s 1.5 2>&1 | grep -v amdgpu | tee $OUT/rows_after.txt
