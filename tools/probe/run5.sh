cd $GRAFT_REPO_ROOT
for v in 0 1; do echo "== VEC_EPI=$v bf16 bench"; T2I_VEC_EPI=$v python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 1.0 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])
    else: print(l[:200])
"; done
echo "== VEC_EPI=1 DMA=2"; T2I_BF16_DMA=2 python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 1.0 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"
