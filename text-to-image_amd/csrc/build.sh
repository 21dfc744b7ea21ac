#!/bin/bash
# Builds libt2i_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).  Usage: [T2I_BUILD_FORCE=1] build.sh [outdir]
# Objects newer than their sources (and the two headers) are reused unless T2I_BUILD_FORCE=1; the last line says how many were compiled.
# T2I_SANITIZE=address,undefined builds the HOST side with those sanitizers (device code unchanged: -fno-gpu-sanitize) into
# <outdir>/san/libt2i_hip_san.so, for tools/sanitize_host.sh — the product library is never built this way.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${1:-$HERE/../lib}"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
LIB="libt2i_hip.so"
LDFLAGS=""
if [ -n "${T2I_SANITIZE:-}" ]; then
  OUT="$OUT/san"
  mkdir -p "$OUT"
  FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -w -fsanitize=$T2I_SANITIZE -fno-gpu-sanitize -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared-libsan"
  LIB="libt2i_hip_san.so"
  LDFLAGS="-fsanitize=$T2I_SANITIZE -shared-libsan"
fi
pids=()
compiled=0
total=0
for f in t2i_igemm t2i_igemm_h t2i_bgemm t2i_aux t2i_thin t2i_winograd t2i_capi; do
  total=$((total + 1))
  if [ "${T2I_BUILD_FORCE:-0}" = "1" ] || [ ! -f "$OUT/$f.o" ] || [ "$HERE/$f.hip" -nt "$OUT/$f.o" ] || [ "$HERE/t2i_internal.h" -nt "$OUT/$f.o" ] \
     || [ "$HERE/../../include/t2i_hip.h" -nt "$OUT/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$OUT/$f.o" &
    pids+=($!)
    compiled=$((compiled + 1))
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done     # (set -e: a failed compile fails the build here)
$HIPCC --offload-arch=gfx950 -shared -fPIC $LDFLAGS -o "$OUT/$LIB" "$OUT/t2i_igemm.o" "$OUT/t2i_igemm_h.o" "$OUT/t2i_bgemm.o" "$OUT/t2i_aux.o" "$OUT/t2i_thin.o" "$OUT/t2i_winograd.o" "$OUT/t2i_capi.o"
echo "built $OUT/$LIB: compiled $compiled of $total objects (the others were up to date), linked 1 shared library"
