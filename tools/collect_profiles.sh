#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box: gpurun -- tools/collect_profiles.sh [round]).
# Kernel trace and each PMC group are separate passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one
# pass; counters are never combined with sys/runtime tracing).  Everything lands in gpurun_out/<round>/ (default r03);
# copy what should be judged into profiles/ with the round prefix.
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
ROUND="${1:-r03}"
OUT="$REPO/gpurun_out/$ROUND"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --instrument off --repeats 1 --min-busy-s 0 --no-config3 --no-side-rows"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace" -o r -- $BENCH --steps 5 --warmup 2 > "$OUT/ktrace.log" 2>&1
cp "$(find "$OUT/ktrace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_bench_s5w2.csv"
python "$REPO/tools/prof_summary.py" "$OUT/ktrace" 9 45 > "$OUT/kernel_stats_summary.txt" 2>&1
python "$REPO/tools/timeline.py" "$OUT/ktrace" 9 30 > "$OUT/timeline_f32.txt" 2>&1     # every dispatch of the last iteration
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo "$grp" | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_$tag" -o r -- $BENCH --steps 3 --warmup 2 --no-graphs > "$OUT/pmc_$tag.log" 2>&1
done
python "$REPO/tools/pmc_summary.py" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES" 5 "$OUT/pmc_igemm.json" f32 > "$OUT/pmc_summary.log" 2>&1
# the same for config 3 (bf16 math): kernel trace + PMC passes
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace_bf16" -o r -- $BENCH --math bf16 --steps 5 --warmup 2 > "$OUT/ktrace_bf16.log" 2>&1
python "$REPO/tools/prof_summary.py" "$OUT/ktrace_bf16" 9 45 > "$OUT/kernel_stats_summary_bf16.txt" 2>&1
python "$REPO/tools/timeline.py" "$OUT/ktrace_bf16" 9 30 > "$OUT/timeline_bf16.txt" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo "$grp" | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmcb_$tag" -o r -- $BENCH --math bf16 --steps 3 --warmup 2 --no-graphs > "$OUT/pmcb_$tag.log" 2>&1
done
python "$REPO/tools/pmc_summary.py" "$OUT/pmcb_FETCH_SIZE" "$OUT/pmcb_WRITE_SIZE" "$OUT/pmcb_SQ_VALU_MFMA_BUSY_CYCLES" 5 "$OUT/pmc_igemm_bf16.json" bf16 > "$OUT/pmc_summary_bf16.log" 2>&1
rm -rf "$OUT/ktrace_bf16" "$OUT/pmcb_FETCH_SIZE" "$OUT/pmcb_WRITE_SIZE" "$OUT/pmcb_SQ_VALU_MFMA_BUSY_CYCLES"
cd "$REPO"
timeout 300 python tools/bench_conv.py --batch 64 > "$OUT/conv_microbench_B64.txt" 2>&1
timeout 300 python tools/bench_conv.py --batch 192 --filter D > "$OUT/conv_microbench_B192.txt" 2>&1
timeout 300 python tools/bench_conv.py --batch 64 --math bf16 > "$OUT/conv_microbench_bf16_B64.txt" 2>&1
timeout 300 python tools/bench_conv.py --batch 512 --math bf16 --reps 10 > "$OUT/conv_microbench_bf16_B512.txt" 2>&1
T2I_SWEEP_MATH=bf16 timeout 600 python tools/sweep_conv.py D2:64 D3:64 D4:64 D7:64 D10:64 G5c:64 G7c:64 G8c:64 G4c:64 G6c:64 D2:192 D3:192 D4:192 D10:192 2>&1 | grep -v amdgpu > "$OUT/bf16_tile_split_sweep.txt"
timeout 300 python tools/bench_aux.py 2>&1 | grep -v amdgpu > "$OUT/hbm_kernels.txt"
{ for b in 8 64; do timeout 200 python text-to-image_amd/models/stackgan/run.py --stage 1 --batch $b 2>&1 | tail -1; done
  for b in 8 32; do timeout 300 python text-to-image_amd/models/stackgan/run.py --stage 2 --batch $b --steps 5 2>&1 | tail -1; done
  timeout 300 python text-to-image_amd/models/stackgan/run.py --stage 2 --batch 32 --steps 5 --math bf16 2>&1 | tail -1
  timeout 600 python text-to-image_amd/models/pggan/train_pggan.py --bench --iters 8 --first 6 --last 12 2>&1 | grep "^pggan"; } > "$OUT/next_rows_throughput.txt"
timeout 600 python bench.py 2>/dev/null | grep '"metric"' > "$OUT/bench_line.json"
# drop the bulky raw traces, keep the per-pass counter csv of the MFMA pass for reference
rm -rf "$OUT/ktrace" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES"
ls -la "$OUT"; cat "$OUT/pmc_summary.log" | tail -30; cat "$OUT/bench_line.json"
