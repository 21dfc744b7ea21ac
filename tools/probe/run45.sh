cd /tmp && export TMPDIR=/tmp
for L in D3 G7c D2; do
rm -rf /tmp/kt_$L
T2I_BENCH_EAGER=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$L -o r -- python $GRAFT_REPO_ROOT/tools/bench_conv.py --batch 64 --filter $L --cache > /tmp/kt_$L.log 2>&1
echo "== $L"; grep "^$L" /tmp/kt_$L.log
python - /tmp/kt_$L <<'PY'
import csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*/*kernel_stats.csv') + glob.glob(sys.argv[1] + '/*kernel_stats.csv'))[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print('%-70s %6s calls avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
