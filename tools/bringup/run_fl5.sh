cd $GRAFT_REPO_ROOT
AMD_LOG_LEVEL=3 ZRA_PP_TRACE=1 ZRA_PP_MIN=1 timeout 30 python tools/bringup/gpu_speed.py 0.01 3 65536 1 > /tmp/apilog.txt 2>&1
grep -v "hipStreamQuery\|Check HW event" /tmp/apilog.txt | tail -400 | cut -c1-300 > gpurun_out/fl5_api.log
wc -l /tmp/apilog.txt gpurun_out/fl5_api.log
