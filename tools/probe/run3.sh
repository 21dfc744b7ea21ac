cd $GRAFT_REPO_ROOT
python tools/probe/gemm_fixed.py 2>&1 | grep -v amdgpu.ids
for v in 1 2 3; do T2I_HIP_LIB=$GRAFT_REPO_ROOT/tools/probe/libs/x$v/libt2i_hip.so python tools/probe/gemm_fixed.py 2>&1 | grep -v amdgpu.ids; done
