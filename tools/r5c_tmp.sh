set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r5e; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "\b\(TCP\|TCC\|SQ\|GRBM\)_[A-Z0-9_a-z]*" | sort -u > $OUT/counter_names.txt
wc -l $OUT/counter_names.txt
# 1. gancls kernel trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_gancls" -o r -- python $REPO/tools/next_rows.py --rows gancls --budget-s 0.5 > "$OUT/kt_gancls.log" 2>&1
python - "$OUT/kt_gancls" > "$OUT/kernel_stats_summary_gancls.txt" 2>&1 <<'PY'
import csv, glob, sys
d = sys.argv[1]
f = (glob.glob(d + '/*/*kernel_stats.csv') + glob.glob(d + '/*kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.3f ms over the whole run' % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:60]:
    print('%-100s calls %6s total %9.1f us %5.2f%% avg %7.1f us' % (r['Name'].split('(')[0][:100], r['Calls'], float(r['TotalDurationNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot, float(r['AverageNs']) / 1e3))
PY
f=$(find $OUT/kt_gancls -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/sequence_gancls.txt <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
# the last iteration: find the period by the adam kernel (2 per iteration)
idx = [i for i, r in enumerate(rows) if 'adam_tf_kernel' in r['Kernel_Name']]
lo, hi = idx[-3] + 1, idx[-1] + 1
it = rows[lo:hi]
t0 = int(it[0]['Start_Timestamp']); prev = None
print('# last iteration: %d dispatches, span %.1f us' % (len(it), (int(it[-1]['End_Timestamp']) - t0) / 1e3))
for r in it:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('t2i::', '')
    grid = int(r.get('Grid_Size_X', 0)) // max(1, int(r.get('Workgroup_Size_X', 256)))
    print('%9.1f %7.1f %5.1f %6d  %s' % ((s - t0) / 1e3, (e - s) / 1e3, 0 if prev is None else max(0, s - prev) / 1e3, grid, name[:90]))
    prev = max(prev or 0, e)
PY
rm -rf $OUT/kt_gancls
head -45 $OUT/kernel_stats_summary_gancls.txt
# 2. PMC passes over the fp32 step (eager, 3 steps) for the batched GEMM
BENCH="python $REPO/bench.py --no-cpu-baseline --instrument off --repeats 1 --min-busy-s 0 --no-config3 --no-side-rows --no-graphs --steps 3 --warmup 2"
i=0
for grp in "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc$i" -o r -- $BENCH > "$OUT/pmc$i.log" 2>&1
  tail -2 "$OUT/pmc$i.log" | cut -c1-200
done
python $REPO/tools/pmc_kernel.py bgemm_kernel $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 > $OUT/pmc_bgemm.txt 2>&1
python $REPO/tools/pmc_kernel.py "igemm_kernel<" $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 > $OUT/pmc_igemm_f32.txt 2>&1
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
cat $OUT/pmc_bgemm.txt
