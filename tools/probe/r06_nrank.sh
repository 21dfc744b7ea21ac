#!/bin/bash
# N gloo ranks on ONE GPU (T2I_SAME_DEVICE=1): the data-parallel preflight's report for several world sizes / schedules.  usage: r06_nrank.sh "<n> [ENV=..]" ...
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"; cd "$REPO"; mkdir -p gpurun_out/r06
for spec in "$@"; do
  set -- $spec; n=$1; shift
  env T2I_SAME_DEVICE=1 T2I_DIST_BACKEND=gloo OMP_NUM_THREADS=2 "$@" timeout 600 python bench.py --gpus $n --steps 1 --warmup 1 --repeats 1 --min-busy-s 0 --no-cpu-baseline --no-config3 --instrument off > gpurun_out/r06/nrank.out 2> gpurun_out/r06/nrank.err
  echo "== $spec: rc=$?"
  grep -o "preflight[^{]*{[^}]*}\|PREFLIGHT FAILED on rank 0[^}]*}" gpurun_out/r06/nrank.err | head -1 | cut -c1-700
  grep -E "Error|error:" gpurun_out/r06/nrank.err | grep -v "ChildFailedError\|elastic" | head -3 | cut -c1-300
done
