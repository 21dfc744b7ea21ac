set -u
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r03f; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --instrument off --repeats 1 --min-busy-s 0 --no-config3"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace_bf16" -o r -- $BENCH --math bf16 --steps 5 --warmup 2 > "$OUT/ktrace_bf16.log" 2>&1
python "$REPO/tools/prof_summary.py" "$OUT/ktrace_bf16" 9 45 > "$OUT/kernel_stats_summary_bf16.txt" 2>&1
python "$REPO/tools/timeline.py" "$OUT/ktrace_bf16" 9 30 > "$OUT/timeline_bf16.txt" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo "$grp" | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmcb_$tag" -o r -- $BENCH --math bf16 --steps 3 --warmup 2 --no-graphs > "$OUT/pmcb_$tag.log" 2>&1
done
python "$REPO/tools/pmc_summary.py" "$OUT/pmcb_FETCH_SIZE" "$OUT/pmcb_WRITE_SIZE" "$OUT/pmcb_SQ_VALU_MFMA_BUSY_CYCLES" 5 "$OUT/pmc_igemm_bf16.json" bf16 > "$OUT/pmc_summary_bf16.log" 2>&1
rm -rf "$OUT/ktrace_bf16" "$OUT/pmcb_FETCH_SIZE" "$OUT/pmcb_WRITE_SIZE" "$OUT/pmcb_SQ_VALU_MFMA_BUSY_CYCLES"
cat $OUT/pmc_igemm_bf16.json
