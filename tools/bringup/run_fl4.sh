cd $GRAFT_REPO_ROOT
echo "== E link ring mode"; ZRA_MF_LK=1 ZRA_LK_MODE=1 ZRA_LK_GROUP=64 timeout 60 python tools/bringup/gpu_speed.py 0.01 3 65536 2 2>&1 | tail -3
echo "== D flags, trace"; ZRA_PP_TRACE=1 ZRA_PP_MIN=1 timeout 60 python tools/bringup/gpu_speed.py 0.01 3 65536 1 2>&1 | tail -4
