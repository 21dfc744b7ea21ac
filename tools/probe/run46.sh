cd $GRAFT_REPO_ROOT
for lib in default nt default nt; do
  if [ $lib = nt ]; then export T2I_HIP_LIB=$PWD/tools/probe/libs/nt/libt2i_hip.so; else unset T2I_HIP_LIB; fi
  echo "== $lib"; python tools/bench_conv.py --batch 64 --cache --filter G7c 2>/dev/null | grep "^G7c"; python tools/bench_conv.py --batch 64 --cache --filter D3 2>/dev/null | grep "^D3"
  python bench.py --no-cpu-baseline --no-config3 --instrument off --min-busy-s 2 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('bench f32', d['value'], d['ms_per_step'])"
done
