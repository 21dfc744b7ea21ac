#!/bin/bash
# N-rank data parallelism on identical data must reproduce the single-process weights exactly (T2I_SAME_DATA=1, bench.py).
# Runs on a 1-GPU box: both ranks on device 0, gloo in place of RCCL.  Prints one signature line per schedule.
export T2I_BENCH_FEED_NOISE=1   # conditioning noise travels in the feed: identical on every rank and in every run
run2() {  # env, bench args
  port=$((29700 + RANDOM % 200))
  env T2I_SAME_DEVICE=1 T2I_DIST_BACKEND=gloo T2I_SAME_DATA=1 T2I_CHECK_SYNC=1 $1 timeout 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --instrument off --no-cpu-baseline --repeats 1 --min-busy-s 0 --no-config3 $2 >/tmp/o 2>/tmp/e
  echo "2 ranks [$1 | $2] rc=$? $(grep -E 'signature|diverged' /tmp/e | head -1 | cut -c1-220)"
}
env T2I_SAME_DATA=1 timeout 300 python bench.py --no-graphs --instrument off --no-cpu-baseline --repeats 1 --min-busy-s 0 --no-config3 --warmup 3 --steps 2 2>&1 >/dev/null | grep signature | sed 's/^/1 process eager:   /'
env T2I_SAME_DATA=1 timeout 300 python bench.py --instrument off --no-cpu-baseline --repeats 1 --min-busy-s 0 --no-config3 --warmup 1 --steps 2 2>&1 >/dev/null | grep signature | sed 's/^/1 process graphs:  /'
run2 "T2I_DP_GRAPHS=0" "--warmup 3 --steps 2"
run2 "T2I_DP_GRAPHS=0 T2I_DP_NO_OVERLAP=1" "--warmup 3 --steps 2"
run2 "" "--warmup 1 --steps 2"
run2 "T2I_DP_GRAPHS=0 T2I_DP_CUT_EAGER=1" "--warmup 3 --steps 2"
# the same with critic-only iterations in between (N_CRITIC = 2: d_step's own graph segments next to the merged D+G ones)
export T2I_N_CRITIC=2
env T2I_SAME_DATA=1 timeout 300 python bench.py --no-graphs --instrument off --no-cpu-baseline --repeats 1 --min-busy-s 0 --no-config3 --warmup 4 --steps 2 2>&1 >/dev/null | grep signature | sed 's/^/N_CRITIC=2, 1 process eager:   /'
run2 "T2I_DP_GRAPHS=0" "--warmup 4 --steps 2"
run2 "" "--warmup 2 --steps 2"
