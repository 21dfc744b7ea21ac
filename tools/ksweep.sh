#!/bin/bash
# usage: tools/ksweep.sh  -- per-K-tile cost of the batched (Winograd) GEMM: igemm dispatch time against Cin at fixed M, N
for shape in 8,8,256,512,3,1,SAME 8,8,512,512,3,1,SAME 8,8,1024,512,3,1,SAME 8,8,2048,512,3,1,SAME 16,16,256,256,3,1,SAME 16,16,512,256,3,1,SAME 16,16,1024,256,3,1,SAME; do
  T2I_ONE_SHAPE=$shape bash /root/repo/tools/prof_one.sh "X$shape" 64 fwd 20 | grep -E "==|gemm"
done
