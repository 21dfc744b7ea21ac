mkdir -p gpurun_out/r03
for L in D3 D4 D10 G5c G7c D7; do for m in fwd bwdD bwdF; do bash tools/prof_one.sh $L 64 $m 20; done; done > gpurun_out/r03/wino_prof.txt 2>&1
for L in D3 D4 D10; do for m in fwd bwdD bwdF; do bash tools/prof_one.sh $L 192 $m 20; done; done >> gpurun_out/r03/wino_prof.txt 2>&1
tail -5 gpurun_out/r03/wino_prof.txt
