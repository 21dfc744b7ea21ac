cd $GRAFT_REPO_ROOT
echo "##### winograd tests with measured values"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -s -k "winograd_3x3_matches or winograd_k4s2_matches" 2>&1 | grep -E "case \(|passed|failed|Error" | head -40
echo "##### b64/b16 + pggan + dp tests"
timeout 2400 python -m pytest tests/test_step_b64_gpu.py tests/test_fullsize_gpu.py tests/test_dp_exactness_gpu.py tests/test_dp_diffdata_gpu.py -q -m gpu -x 2>&1 | tail -30
echo "##### bgemm epilogue ablation (fp32 bench_conv --cache B=64)"
python tools/bench_conv.py --batch 64 --reps 10 --cache 2>&1 | grep TOTAL
T2I_HIP_LIB=$GRAFT_REPO_ROOT/tools/probe/libs/b1/libt2i_hip.so python tools/bench_conv.py --batch 64 --reps 10 --cache 2>&1 | grep TOTAL
echo "##### timelines"
export T2I_TIMELINE_ALL=1
bash tools/quick_timeline.sh r04_f32 > /dev/null 2>&1
bash tools/quick_timeline.sh r04_bf16 --math bf16 > /dev/null 2>&1
head -12 gpurun_out/r04_f32/timeline.txt
