"""The data-parallel contract on DIFFERENT data: two gloo ranks on one GPU, distinct feeds, local batch b.
See tests/workers/dp_diffdata_worker.py for what is compared and why (reference models/wgancls/model.py:63-65, 85, 100)."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('stacked', ['0', '1'], ids=['two_pass', 'stacked'])
def test_two_ranks_on_distinct_data_equal_one_process_at_twice_the_batch(stacked):
    """stacked = '1' (the default critic step since round 6, text-to-image_amd/stacked.py): the forward GEMMs of a 4b-row and an 8b-row
    stacked pass take different split-K plans, so their activations differ in the last bit, a handful of units at their lrelu kink
    take the other branch and the gradient penalty's part of the filter gradients moves by ~1e-3 (tools/probe/r06_stack_check.py:
    the same two forms agree to 7e-7 at b = 4, where the forward is bit-identical; it is the piecewise-linear net, not the kernels —
    tests/test_step_b64_gpu.py pins the branches for that reason).  A violated contract — a wrong 1/N, a per-rank kt step, a missing
    term — is an O(1) error, so the stacked form is held to 1e-2 and the two-pass form (bit-identical forwards at these sizes) to
    the strict 2e-5."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29841', os.path.join('tests', 'workers', 'dp_diffdata_worker.py')]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, T2I_QUIET='1', T2I_STACK_XHAT=stacked), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out = r.stdout.decode() + r.stderr.decode()
    assert r.returncode == 0, out[-4000:]
    m = re.search(r'DPDIFF (\{.*\})', out)
    assert m, out[-3000:]
    rep = json.loads(m.group(1))
    # (1), (3): the exchange delivers exactly the sum of what each rank computes alone on its own data
    assert rep['critic_sum_is_exactly_local0_plus_local1'], rep
    assert rep['kt_means_sum_exact'], rep
    assert rep['generator_sum_is_exactly_local0_plus_local1'], rep
    # (2): critic — two ranks x b  ==  one process x 2b, to fp32 summation-order accuracy
    assert rep['critic_mean_vs_batch_2b_worst_rel_l2'] <= (2e-5 if stacked == '0' else 1e-2), rep
    assert abs(rep['kt_dp'] - rep['kt_batch_2b']) <= 1e-6, rep
    assert abs(rep['kt_dp'] - rep['kt_start']) > 0.0, rep                      # the step did move kt
    for k, v in rep['scalars_batch_2b'].items():
        assert abs(rep['scalars_mean_over_ranks'][k] - v) <= (1e-5 if stacked == '0' else 2e-4) * max(abs(v), 1.0), (k, rep)
    # the generator's batch norm is per replica: the rank-mean is NOT the batch-2b gradient (recorded so nobody assumes it is)
    assert rep['generator_mean_vs_batch_2b_worst_rel_l2__not_a_contract'] > 1e-4, rep
