cd $GRAFT_REPO_ROOT
echo "== A link mode 0, default LDS"; ZRA_MF_LK=1 timeout 60 python tools/bringup/gpu_speed.py 0.01 3 65536 2 2>&1 | tail -3
echo "== B link mode 0, LDS 155712"; ZRA_PP_LDS=155712 ZRA_MF_LK=1 timeout 60 python tools/bringup/gpu_speed.py 0.01 3 65536 2 2>&1 | tail -3
echo "== C flags, LDS 155712, trace"; ZRA_PP_LDS=155712 ZRA_PP_TRACE=1 ZRA_PP_MIN=1 timeout 60 python tools/bringup/gpu_speed.py 0.01 3 65536 2 2>&1 | tail -4
echo "== D flags, default LDS, trace"; ZRA_PP_TRACE=1 ZRA_PP_MIN=1 timeout 60 python tools/bringup/gpu_speed.py 0.01 3 65536 2 2>&1 | tail -4
