#!/bin/bash
# Builds libt2i_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).  Usage: build.sh [outdir]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${1:-$HERE/../lib}"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
pids=()
for f in t2i_igemm t2i_igemm_h t2i_bgemm t2i_aux t2i_thin t2i_winograd t2i_capi; do
  if [ ! -f "$OUT/$f.o" ] || [ "$HERE/$f.hip" -nt "$OUT/$f.o" ] || [ "$HERE/t2i_internal.h" -nt "$OUT/$f.o" ] \
     || [ "$HERE/../../include/t2i_hip.h" -nt "$OUT/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$OUT/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libt2i_hip.so" "$OUT/t2i_igemm.o" "$OUT/t2i_igemm_h.o" "$OUT/t2i_bgemm.o" "$OUT/t2i_aux.o" "$OUT/t2i_thin.o" "$OUT/t2i_winograd.o" "$OUT/t2i_capi.o"
echo "built $OUT/libt2i_hip.so"
