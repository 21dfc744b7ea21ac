"""Exactness of the data-parallel gradient exchange on a real device (two gloo ranks sharing GPU 0).

With T2I_SAME_DATA=1 every rank sees rank 0's batch and noise, so the rank-averaged gradients equal the local ones bit for
bit and an N-rank run must end with EXACTLY the single-process weights, Adam state and kt.  Unlike a comparison of the
replicas with each other this also catches buckets that all ranks exchange too early (the round's late bug: hooks firing
for leaves whose gradient is undefined)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _signature(cmd, env_extra):
    env = dict(os.environ)
    env.update(T2I_SAME_DATA='1', T2I_FILTER_CACHE='1', T2I_BENCH_FEED_NOISE='1')     # noise in the feed: the same on every rank and in every run
    env.update(env_extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    err = r.stderr.decode()
    assert r.returncode == 0, err[-3000:]
    m = re.search(r'signature after (\d+) iterations: (\[.*\])', err)
    assert m, err[-2000:]
    return int(m.group(1)), m.group(2)


def _two_ranks(port, args, env_extra):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), 'bench.py', '--gpus', '2', '--instrument', 'off', '--no-cpu-baseline', '--repeats', '1', '--min-busy-s', '0', '--no-config3'] + args
    env = dict(T2I_SAME_DEVICE='1', T2I_DIST_BACKEND='gloo', T2I_CHECK_SYNC='1')
    env.update(env_extra)
    return _signature(cmd, env)


def test_two_rank_exchange_reproduces_the_single_process_weights():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    # (T2I_PAIR_G=0: a single GPU stacks the generator's two evaluations of an iteration into one pass; the data-parallel schedules keep
    # two passes — the generator step's forward is what hides the critic's exchange — so their reference does too)
    single = _signature([sys.executable, 'bench.py', '--no-graphs', '--instrument', 'off', '--no-cpu-baseline', '--repeats', '1',
                         '--min-busy-s', '0', '--no-config3', '--warmup', '3', '--steps', '1'], {'T2I_PAIR_G': '0'})
    assert single[0] == 4
    # eager schedule: buckets leave while the backward is still running (first step learns the contribution counts)
    # (T2I_PREFLIGHT=0 on the three fp32 runs: the comparison with `single` below IS that check — the bf16 run and
    # test_bench_gpus_2_launches_itself_and_describes_the_exchange keep bench.py's own preflight)
    eager = _two_ranks(29811, ['--warmup', '3', '--steps', '1'], {'T2I_DP_GRAPHS': '0', 'T2I_PREFLIGHT': '0'})
    # graph segments (the default): 2 eager set-up iterations, capture, then replays — the stacked critic step's arena on the wire while
    # the generator forward runs, the generator's backward cut once
    graphs = _two_ranks(29812, ['--warmup', '1', '--steps', '1'], {'T2I_PREFLIGHT': '0'})
    # the same segment sequence launched eagerly (WGanCls._dg_cut_eager)
    # ... with the fp32 buckets through the reduce-scatter + all-gather form (T2I_DP_F32_EXCHANGE=rs_ag; on gloo: its gather-and-sum
    # stand-in): two ranks, so bit for bit the all-reduce's result
    cut_eager = _two_ranks(29813, ['--warmup', '3', '--steps', '1'], {'T2I_DP_GRAPHS': '0', 'T2I_DP_CUT_EAGER': '1', 'T2I_DP_F32_EXCHANGE': 'rs_ag', 'T2I_PREFLIGHT': '0'})
    assert eager == single, (eager, single)
    assert graphs == single, (graphs, single)
    assert cut_eager == single, (cut_eager, single)
    # config 3: bf16 gradient buckets summed in fp32.  bench.py's preflight demands that the two-rank run on identical data equals,
    # bit for bit, a single replica whose gradient arena is rounded to bf16 once (dp.LocalRounding) and aborts otherwise
    _two_ranks(29814, ['--math', 'bf16', '--warmup', '1', '--steps', '1'], {})
    # (that run went through bench.py's own data-parallel preflight, which aborts the run on a mismatch)


def test_bench_gpus_2_launches_itself_and_describes_the_exchange():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE (the driver's SCALE invocation form): bench.py re-runs itself as two
    ranks through torch.distributed.run (two gloo ranks on GPU 0 here: T2I_SAME_DEVICE / T2I_DIST_BACKEND), and rank 0's ONE JSON line
    says how many ranks the communicator counted and what the gradient exchange moved and cost."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(T2I_SAME_DEVICE='1', T2I_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--repeats', '1', '--min-busy-s', '0',
                        '--no-cpu-baseline', '--no-config3', '--instrument', 'off'], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2 and out['steps'] == 2
    assert out['config']['global_batch'] == 128 and out['scaling'] == 'weak'
    ex = out['gradient_exchange']
    assert ex['ranks'] == 2 and ex['dtype_on_wire'] == 'f32' and ex['backend'] == 'gloo'
    # both arenas cross the wire once per iteration: 28 995 329 + 22 643 287 fp32 parameters (+ per-variable padding to 16 bytes)
    assert 4 * (28995329 + 22643287) <= ex['payload_bytes_per_step'] <= 4 * (28995329 + 22643287) + 4096
    assert ex['ms_in_collectives_per_step'] > 0 and ex['ms_compute_stream_stalled_per_step'] >= 0
    assert out['dp_preflight']['ok'] and out['dp_preflight']['exact']


def _bench_gpus_3_preflight_holds_an_inexact_sum_to_rounding():
    """(110 s on one GPU — three processes, eight iterations each behind their imports — so it is part of the suite under T2I_FULL_SWEEP=1 only,
    like the extra PGGAN cases: the default `-m gpu` run keeps the two-rank tests above.)  Three ranks (gloo on GPU 0): the sum of three identical gradients is not representable in general, so the preflight cannot demand bits
    — its free-running form of rounds 3-5 declared exactly this exchange broken (95 % of the weights off after 4 iterations, the model amplifying
    the last bit) and stopped the run.  The lockstep form must pass: gradients of the critic's arena within 1e-5 of the single replica's, the
    generator's within 5e-3 (measured 8e-8 / 6e-5), nothing fatal, and the line is printed."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(T2I_SAME_DEVICE='1', T2I_DIST_BACKEND='gloo', OMP_NUM_THREADS='2')
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '3', '--steps', '1', '--warmup', '1', '--repeats', '1', '--min-busy-s', '0',
                        '--no-cpu-baseline', '--no-config3', '--instrument', 'off'], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')][-1])
    pf = out['dp_preflight']
    assert out['n_gpus'] == 3 and out['config']['global_batch'] == 192 and pf['ranks'] == 3 and pf['form'] == 'lockstep'
    assert pf['ok'] and not pf['fatal'] and not pf['exact_required'] and pf['ranks_outside_bounds'] == 0 and out['dp_preflight_ok'] is True
    assert pf['max_gradient_diff_rel']['critic'] <= 1e-5 and pf['max_gradient_diff_rel']['generator'] <= 5e-3


if os.environ.get('T2I_FULL_SWEEP') == '1':
    test_bench_gpus_3_preflight_holds_an_inexact_sum_to_rounding = _bench_gpus_3_preflight_holds_an_inexact_sum_to_rounding
