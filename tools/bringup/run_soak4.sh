cd $GRAFT_REPO_ROOT
# round 4: the differential compress / decode soak on fresh seeds, default path and the two opt-in dfast formulations
( timeout 500 python tools/bringup/gpu_soak.py 5000 5200; ZRA_MF_LK=1 timeout 500 python tools/bringup/gpu_soak.py 5200 5350; ZRA_MF_LK=1 ZRA_LK_MODE=1 ZRA_LK_GROUP=16 timeout 400 python tools/bringup/gpu_soak.py 5350 5450; ZRA_MF_FLAGS=1 ZRA_PP_MIN=1 timeout 500 python tools/bringup/gpu_soak.py 5450 5600 ) 2>&1 | grep -v amdgpu.ids | grep "FAIL\|soak done\|Error" 
