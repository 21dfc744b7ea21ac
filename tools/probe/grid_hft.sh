for boost in 100 115 130 150; do for ovh in 40 80 120; do
  a=$(T2I_HFT_BOOST=$boost T2I_HFT_OVH=$ovh python tools/bench_conv.py --math bf16 --batch 64 2>&1 | grep "TOTAL bwd_filter" | awk '{print $5}')
  b=$(T2I_HFT_BOOST=$boost T2I_HFT_OVH=$ovh python tools/bench_conv.py --math bf16 --batch 192 --filter D 2>&1 | grep "TOTAL bwd_filter" | awk '{print $5}')
  echo "boost $boost ovh $ovh : B64 $a us  B192(D) $b us"
done; done
