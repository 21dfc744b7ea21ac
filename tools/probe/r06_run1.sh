mkdir -p gpurun_out/r06
python __graft_entry__.py smoke > gpurun_out/r06/smoke.log 2>&1; tail -2 gpurun_out/r06/smoke.log
python -m pytest tests/test_step_b64_gpu.py tests/test_fullsize_gpu.py::test_stackgan_stage2_full_size tests/test_fullsize_gpu.py::test_stackgan_stage2_all_bf16_envelope tests/test_kernels_gpu.py::test_grouped_batch_norm_odd_channels tests/test_step_gpu.py::test_side_stream_and_dp_single_rank_match_plain tests/test_step_gpu.py::test_all_bf16_math_tiny_step_unpinned_envelope tests/test_dp_exactness_gpu.py "tests/test_fullsize_gpu.py::test_pggan_stage_full_width[3-True-16]" -m gpu -q -s --durations=30 > gpurun_out/r06/tests1.log 2>&1; tail -5 gpurun_out/r06/tests1.log
for b in 64 192 256 8 24 32; do python tools/bench_conv.py --cache --filter D --batch $b > gpurun_out/r06/conv_D_f32_b$b.txt 2>&1; done
for b in 64 192 256 8 24 32; do python tools/bench_conv.py --cache --filter D --batch $b --math bf16 --storage bf16 > gpurun_out/r06/conv_D_bf16_b$b.txt 2>&1; done
