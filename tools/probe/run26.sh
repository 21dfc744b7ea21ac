cd $GRAFT_REPO_ROOT
python tools/probe/native_ops.py f32 2>&1 | tail -40
echo ---- bf16
python tools/probe/native_ops.py bf16 2>&1 | tail -40
