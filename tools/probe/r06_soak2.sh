mkdir -p gpurun_out/r06
python tools/probe/r06_soak.py 2>/dev/null | tail -1 > gpurun_out/r06/soak_a.json
T2I_STORE_FIRST=0 python tools/probe/r06_soak.py 2>/dev/null | tail -1 > gpurun_out/r06/soak_b.json
T2I_BGEMM_TILE=11 python tools/probe/r06_soak.py 2>/dev/null | tail -1 > gpurun_out/r06/soak_c.json
SOAK_B=64 SOAK_ITERS=30 python tools/probe/r06_soak.py 2>/dev/null | tail -1 > gpurun_out/r06/soak_d.json
SOAK_B=64 SOAK_ITERS=30 T2I_STORE_FIRST=0 T2I_BGEMM_TILE=11 python tools/probe/r06_soak.py 2>/dev/null | tail -1 > gpurun_out/r06/soak_e.json
python - <<'PY'
import json
L={k: json.load(open('gpurun_out/r06/soak_%s.json' % k)) for k in 'abcde'}
print('B=16, 60 iterations: store-first on == off:', L['a']['rows'] == L['b']['rows'], L['a']['wnorm'] == L['b']['wnorm'], '| tile by items == 64x64:', L['a']['rows'] == L['c']['rows'], L['a']['wnorm'] == L['c']['wnorm'])
print('B=64, 30 iterations: both toggles:', L['d']['rows'] == L['e']['rows'], L['d']['wnorm'] == L['e']['wnorm'], 'finite', L['d']['finite'])
PY
