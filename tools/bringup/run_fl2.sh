cd $GRAFT_REPO_ROOT
ZRA_PP_TRACE=1 ZRA_PP_MIN=1 timeout 120 python tools/bringup/gpu_speed.py 0.01 3 65536 2 > gpurun_out/fl_dbg.log 2>&1
tail -12 gpurun_out/fl_dbg.log
