cd $GRAFT_REPO_ROOT
b() { python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 2 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
echo -n "DMA=1 w4: "; T2I_BF16_DMA=1 T2I_BF16_WAVES=4 b
echo -n "DMA=1 w8: "; T2I_BF16_DMA=1 T2I_BF16_WAVES=8 b
echo -n "DMA=3   : "; T2I_BF16_DMA=3 b
echo -n "DMA=4   : "; T2I_BF16_DMA=4 b
done
