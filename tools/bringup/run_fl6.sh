cd $GRAFT_REPO_ROOT
echo "== F flags, same stream"; ZRA_PP_SAMESTREAM=1 ZRA_PP_MIN=1 timeout 40 python tools/bringup/gpu_speed.py 0.01 3 65536 2 2>&1 | tail -3
echo "== G flags, same stream, 1 GiB"; ZRA_PP_SAMESTREAM=1 timeout 60 python tools/bringup/gpu_speed.py 1 3 65536 2 2>&1 | tail -3
