cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_storage_gpu.py -q -m gpu -x 2>&1 | tail -4
for v in 0 1; do echo "PAIR=$v"; T2I_PAIR=$v python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 2 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"; done
for px in 4096 49152; do echo "PAIR_MAX_PX=$px"; T2I_PAIR_MAX_PX=$px python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 2 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"; done
