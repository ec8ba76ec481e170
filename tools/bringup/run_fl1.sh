cd $GRAFT_REPO_ROOT
ZRA_PP_MIN=1 timeout 400 python tools/bringup/gpu_lk_check.py 0 > gpurun_out/fl_check1.log 2>&1; tail -3 gpurun_out/fl_check1.log
for cfg in "ZRA_MF_FLAGS=0" "ZRA_MF_FLAGS=1" "ZRA_MF_FLAGS=1 ZRA_PP_CUS=24" "ZRA_MF_FLAGS=1 ZRA_PP_CUS=48"; do echo "== $cfg"; env $cfg timeout 200 python tools/bringup/gpu_speed.py 4 2>&1 | grep compress; done > gpurun_out/fl_speed1.log 2>&1
cat gpurun_out/fl_speed1.log
