cd $GRAFT_REPO_ROOT
echo "##### pair tests"
timeout 900 python -m pytest tests/test_storage_gpu.py -q -m gpu -x -s 2>&1 | grep -E "fused|passed|failed|Error|assert" | head -20
timeout 1500 python -m pytest tests/test_step_gpu.py tests/test_kernels_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error" | head
echo "##### bf16 bench pair 0/1"
for v in 0 1; do echo "PAIR=$v"; T2I_PAIR=$v python bench.py --math bf16 --no-cpu-baseline --no-config3 --instrument off --min-busy-s 1.5 > /tmp/b.out 2> /tmp/b.err; grep "^{" /tmp/b.out | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'])"; grep -v amdgpu.ids /tmp/b.err | tail -8; done
python -c "
import sys; sys.path.insert(0,'.')
import t2i_amd
from t2i_amd._lib import lib
print('stat', lib.t2i_stat(b'pair_fused'))"
