cd $GRAFT_REPO_ROOT
for d in 1 2 3 4; do echo -n "DMA=$d "; T2I_BF16_DMA=$d python tools/probe/loop_ablation.py default; done
echo -n "DMA=1 waves8 "; T2I_BF16_WAVES=8 python tools/probe/loop_ablation.py default
