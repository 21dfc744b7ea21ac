cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do echo -n "H_STATS=$v: "; T2I_H_STATS=$v python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 2 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"; done
