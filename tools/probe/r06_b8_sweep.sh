# wgancls at B = 8 (stacked critic pass: 32 rows; paired generator: 16 rows) under the Winograd thresholds
run() { echo "== $*"; env "$@" python tools/next_rows.py --rows wgancls_b8 --budget-s 1.5 2>&1 | grep wgancls_b8 | cut -c1-60; env "$@" python tools/next_rows.py --rows wgancls_b8 --math bf16 --budget-s 1.5 2>&1 | grep wgancls_b8 | cut -c1-60; }
run T2I_NOOP=1
run T2I_WINOGRAD_K4S2_MINWORK=80000000
run T2I_WINOGRAD_K4S2_MINWORK=320000000
run T2I_WINOGRAD_MINWORK=25000000
run T2I_WINOGRAD_MINWORK=100000000
run T2I_WINOGRAD_K4S2_MINITEMS=200
run T2I_NOOP=1
