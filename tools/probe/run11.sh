cd $GRAFT_REPO_ROOT
for c in 256 128 96 160; do echo "#### PAIR_CUS=$c"; T2I_PAIR_CUS=$c python tools/probe/pair_bench.py 2>&1 | grep -v "amdgpu.ids\|t2i plan\] M="; done
for c in 256 128; do echo "PAIR_CUS=$c"; T2I_PAIR_CUS=$c python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 1.5 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"; done
