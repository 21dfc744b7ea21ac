#!/bin/bash
# usage: pf.sh "<env assignments>" "<extra bench args>"
port=$((29620 + RANDOM % 200))
env T2I_SAME_DEVICE=1 T2I_DIST_BACKEND=gloo T2I_CHECK_SYNC=1 $1 timeout 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --steps 2 --warmup 1 $2 >/tmp/o 2>/tmp/e
echo "[$1 | $2] rc=$? $(grep -E 'sync check passed|replicas diverged' /tmp/e | head -1 | cut -c1-1500)"
