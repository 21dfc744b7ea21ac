cd $GRAFT_REPO_ROOT
export T2I_TIMELINE_ALL=1
bash tools/quick_timeline.sh r04_bf16 --math bf16 > /dev/null 2>&1
bash tools/quick_timeline.sh r04_f32 > /dev/null 2>&1
python tools/next_rows.py --math f32 --budget-s 1.0 2>&1 | grep -v amdgpu.ids
python tools/next_rows.py --math bf16 --budget-s 1.0 --rows stage2 wgancls_b8 gancls 2>&1 | grep -v amdgpu.ids
