cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
{ timeout 900 python tools/next_rows.py --math f32 --budget-s 2.0 2>&1 | grep -v amdgpu
  timeout 600 python tools/next_rows.py --math bf16 --budget-s 2.0 --rows wgancls_b8 2>&1 | grep -v amdgpu
  timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu
import sys; sys.path.insert(0, '.')
import t2i_amd
from t2i_amd import kernels as K
K.filter_cache(True)
from tools.next_rows import measure_rows
for r in measure_rows(['stage2'], 'bf16', 2.0, storage='f32'):
    print('%-16s bf16 (fp32 tensors) B=%-3d %8.2f ms/iteration %9.1f img/s | %.2f GFLOP/img | %.3f of the bf16 matrix peak\n                      arithmetic: %s — %s' % (r['row'], r['batch'], r['ms_per_iteration'], r['images_per_sec'], r['algorithmic_gflop_per_image'], r['frac_vs_driver_ms'], r['arithmetic']['mode'], r['arithmetic'].get('parity')) if 'error' not in r else r)
PY
} > gpurun_out/r06/next_rows_throughput.txt
cat gpurun_out/r06/next_rows_throughput.txt | cut -c1-200
