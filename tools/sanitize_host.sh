#!/bin/bash
# Sanitizer run of the host side of the C ABI: builds lib/san/libt2i_hip_san.so with ASan + UBSan on the host code
# (csrc/build.sh, T2I_SANITIZE), builds tests/workers/san_host.c against it and runs it.  No GPU needed: only entry points that
# return before any launch are exercised.  ~2 minutes (seven hipcc compiles).  Exit code != 0 on any sanitizer report.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SAN="${T2I_SANITIZE:-address,undefined}"
T2I_SANITIZE="$SAN" bash "$ROOT/text-to-image_amd/csrc/build.sh" | tail -1
OUT="$ROOT/text-to-image_amd/lib/san"
CLANG=/opt/rocm/lib/llvm/bin/clang
$CLANG -O1 -g -fsanitize="$SAN" -fno-sanitize-recover=undefined -shared-libsan -I"$ROOT/include" "$ROOT/tests/workers/san_host.c" \
  -L"$OUT" -lt2i_hip_san -Wl,-rpath,"$OUT" -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,"$(dirname "$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)")" -o "$OUT/san_host"
ASAN_OPTIONS="${ASAN_OPTIONS:-detect_leaks=0:abort_on_error=0:protect_shadow_gap=0}" UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1" "$OUT/san_host"
