echo "== default"; T2I_DEBUG_PLAN=1 python tools/bench_conv.py --cache --batch 64 --filter G8c 2>&1 | grep -E "plan\]|^G8c" | sort | uniq -c | sort -rn | head -8
for sk in 4 8 16 32; do echo "== force splitk $sk"; T2I_FORCE_SPLITK=$sk python tools/bench_conv.py --cache --batch 64 --filter G8c 2>&1 | grep -E "^G8c"; done
for t in 11 21 12; do echo "== force tile $t"; T2I_FORCE_TILE=$t python tools/bench_conv.py --cache --batch 64 --filter G8c 2>&1 | grep -E "^G8c"; done
echo "== winograd filter-grad forced (minc)"; T2I_DEBUG_PLAN=1 python tools/bench_conv.py --cache --batch 128 --filter G8c 2>&1 | grep -E "plan\]|^G8c" | sort | uniq -c | sort -rn | head -6
