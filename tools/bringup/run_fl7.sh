cd $GRAFT_REPO_ROOT
echo "== flags, markers"; ZRA_PP_TRACE=1 ZRA_PP_MIN=1 timeout 30 python tools/bringup/gpu_speed.py 0.01 3 65536 1 2>&1 | grep -v amdgpu.ids | tail -4
