#!/bin/bash
# rocprofv3 kernel table of C4's one-GPU share: 8 GiB log-like, level 9, 256 KiB frames (tools/bringup/gpu_speed.py, two passes)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
LOGLIKE=1 timeout 600 python3 tools/bringup/gpu_speed.py 8 9 262144 2 > gpurun_out/c4_speed.txt 2>&1 < /dev/null
cd /tmp; rm -rf /tmp/kst_c4
LOGLIKE=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst_c4 -o k -- python3 $GRAFT_REPO_ROOT/tools/bringup/gpu_speed.py 8 9 262144 2 > $GRAFT_REPO_ROOT/gpurun_out/c4_speed_under_rocprof.txt 2>&1 < /dev/null
f=$(find /tmp/kst_c4 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_c4.csv
grep -v amdgpu.ids $GRAFT_REPO_ROOT/gpurun_out/c4_speed.txt | cut -c1-200; head -6 $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_c4.csv | cut -c1-140
