cd $GRAFT_REPO_ROOT
echo "##### pair overlap"
python tools/probe/pair_overlap.py 2>&1 | grep -v amdgpu.ids
echo "##### bench default"
( time python bench.py > gpurun_out/bench_r04_a.json 2> gpurun_out/bench_r04_a.err ) 2>&1 | tail -3
tail -c 3000 gpurun_out/bench_r04_a.json | head -c 1500; echo
tail -5 gpurun_out/bench_r04_a.err
echo "##### full gpu suite"
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15
echo "##### bf16 error table"
timeout 1500 python tools/bf16_error_table.py 2>&1 | grep -v amdgpu.ids
