cd $GRAFT_REPO_ROOT
python -m pytest tests/test_step_gpu.py tests/test_storage_gpu.py -x -q -m gpu 2>&1 | tail -5
b() { python bench.py "$@" --no-cpu-baseline --no-config3 --instrument off --min-busy-s 2 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"; }
for v in 0 1 0 1; do echo -n "bf16 TRUST=$v: "; T2I_TRUST_IMAGES=$v b --math bf16; done
for v in 0 1 0 1; do echo -n "bf16 PAIR_REDUCE=$v: "; T2I_PAIR_REDUCE=$v b --math bf16; done
for v in 0 1 0 1; do echo -n "f32 TRUST=$v: "; T2I_TRUST_IMAGES=$v b; done
