mkdir -p gpurun_out/r06
python tools/probe/r06_soak.py 2>/dev/null | tail -1 > gpurun_out/r06/soak_new.json
T2I_STACK_XHAT=0 T2I_PAIR_G=0 T2I_STORE_FIRST=0 T2I_BGEMM_TILE=11 python tools/probe/r06_soak.py 2>/dev/null | tail -1 > gpurun_out/r06/soak_old.json
python - <<'PY'
import json
a=json.load(open('gpurun_out/r06/soak_new.json')); b=json.load(open('gpurun_out/r06/soak_old.json'))
print('finite', a['finite'], b['finite'], 'wnorm', a['wnorm'], b['wnorm'])
for i in (0,1,2,3,5,9,19,39,59):
    if i < len(a['rows']):
        print(i+1, ['%.5g' % v for v in a['rows'][i]], ['%.5g' % v for v in b['rows'][i]])
PY
