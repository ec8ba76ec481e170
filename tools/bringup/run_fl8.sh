cd $GRAFT_REPO_ROOT
echo "== flags small"; ZRA_PP_MIN=1 timeout 40 python tools/bringup/gpu_speed.py 0.01 3 65536 2 2>&1 | grep -v amdgpu.ids | tail -3
echo "== ring small"; ZRA_MF_LK=1 ZRA_LK_MODE=1 ZRA_LK_GROUP=64 timeout 40 python tools/bringup/gpu_speed.py 0.01 3 65536 2 2>&1 | grep -v amdgpu.ids | tail -3
for cfg in "ZRA_MF_FLAGS=0" "ZRA_MF_FLAGS=1" "ZRA_MF_FLAGS=1 ZRA_PP_CUS=24" "ZRA_MF_FLAGS=1 ZRA_PP_CUS=48"; do echo "== $cfg"; env $cfg timeout 100 python tools/bringup/gpu_speed.py 4 2>&1 | grep compress; done
