set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r5f; mkdir -p $OUT
cd $REPO
python -m pytest tests/test_gancls.py "tests/test_fullsize_gpu.py" tests/test_dp_segments_gpu.py tests/test_dp_models_gloo.py -m gpu -q -x > $OUT/tests2.log 2>&1
tail -12 $OUT/tests2.log
