#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box: gpurun -- tools/collect_profiles.sh [round]).
# Kernel trace and each PMC group are separate passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one
# pass; counters are never combined with sys/runtime tracing).  Everything lands in gpurun_out/<round>/ (default r06);
# copy what should be judged into profiles/ with the round prefix.
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
ROUND="${1:-r06}"
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
timeout 300 python tools/bench_conv.py --batch 256 --filter D > "$OUT/conv_microbench_B256.txt" 2>&1     # the stacked critic pass: 4B rows
timeout 300 python tools/bench_conv.py --batch 64 --math bf16 > "$OUT/conv_microbench_bf16_B64.txt" 2>&1
timeout 300 python tools/bench_conv.py --batch 512 --math bf16 --reps 10 > "$OUT/conv_microbench_bf16_B512.txt" 2>&1
{ echo "== bf16 tensors in and out (config 3 as it runs: no cast launches inside the timed calls), B = 64"; timeout 300 python tools/bench_conv.py --batch 64 --math bf16 --storage bf16 --reps 10 2>&1 | grep -v amdgpu
  echo "== B = 512"; timeout 300 python tools/bench_conv.py --batch 512 --math bf16 --storage bf16 --reps 10 2>&1 | grep -v amdgpu; } > "$OUT/conv_microbench_bf16_tensors.txt"
timeout 300 python tools/bench_aux.py 2>&1 | grep -v amdgpu > "$OUT/hbm_kernels.txt"
{ timeout 900 python tools/next_rows.py --math f32 --budget-s 2.0 2>&1 | grep -v amdgpu
  timeout 600 python tools/next_rows.py --math bf16 --budget-s 2.0 --rows wgancls_b8 2>&1 | grep -v amdgpu     # (the compliant config-3 arithmetic; every bf16 row prints its arithmetic and parity test)
  timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu
import sys; sys.path.insert(0, '.')
import t2i_amd
from t2i_amd import kernels as K
K.filter_cache(True)
from tools.next_rows import measure_rows
for r in measure_rows(['stage2'], 'bf16', 2.0, storage='f32'):
    print('%-16s bf16 (fp32 tensors) B=%-3d %8.2f ms/iteration %9.1f img/s | %.2f GFLOP/img | %.3f of the bf16 matrix peak\n                      arithmetic: %s — %s' % (r['row'], r['batch'], r['ms_per_iteration'], r['images_per_sec'], r['algorithmic_gflop_per_image'], r['frac_vs_driver_ms'], r['arithmetic']['mode'], r['arithmetic'].get('parity')) if 'error' not in r else r)
PY
} > "$OUT/next_rows_throughput.txt"
# per-kernel statistics of the next rows (StackGAN Stage-II, PGGAN stage 7): one kernel-trace pass each
cd /tmp
for row in stage2 pggan7; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$row" -o r -- python $REPO/tools/next_rows.py --rows $row --budget-s 1.0 > "$OUT/kt_$row.log" 2>&1
  python - "$OUT/kt_$row" > "$OUT/kernel_stats_summary_$row.txt" 2>&1 <<'PY'
import csv, glob, sys
d = sys.argv[1]
f = (glob.glob(d + '/*/*kernel_stats.csv') + glob.glob(d + '/*kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('whole process (set-up iterations + capture warm-up + timed replays): kernel time %.1f ms; share per kernel' % (tot / 1e6))
for r in rows[:40]:
    print('%-90s %7s calls %6.2f%% avg %9.1f us' % (r['Name'][:90], r['Calls'], float(r['Percentage']), float(r['AverageNs']) / 1e3))
PY
  rm -rf "$OUT/kt_$row"
done
cd "$REPO"
timeout 600 python bench.py 2>/dev/null | grep '"metric"' > "$OUT/bench_line.json"
# drop the bulky raw traces, keep the per-pass counter csv of the MFMA pass for reference
rm -rf "$OUT/ktrace" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES"
ls -la "$OUT"; cat "$OUT/pmc_summary.log" | tail -30; cat "$OUT/bench_line.json"
