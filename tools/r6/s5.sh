#!/bin/bash
# round 6, session 5: kernel trace of one 16 GiB compress (the split stage's three kernels), host trace of single-query random access
export TMPDIR=/tmp; mkdir -p gpurun_out; root=$(pwd)
cd /tmp; rm -rf /tmp/kt5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -o k -- python3 $root/tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd.so 16 2 > $root/gpurun_out/r06_kt_compress.log 2>&1 < /dev/null
f=$(find /tmp/kt5 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $root/gpurun_out/r06_kernel_stats_compress16g.csv && head -12 $root/gpurun_out/r06_kernel_stats_compress16g.csv | cut -c1-160
cd $root
ZRA_RA_TRACE=1 timeout 600 python3 tools/bringup/gpu_ra_latency.py 1 > gpurun_out/r06_ra_latency_trace.txt 2>&1 < /dev/null
grep -v "^  ra\|^    small" gpurun_out/r06_ra_latency_trace.txt | tail -5
grep "^  ra\|^    small" gpurun_out/r06_ra_latency_trace.txt | sed -n 20,60p
cd /tmp; rm -rf /tmp/kt6
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt6 -o k -- python3 $root/tools/bringup/gpu_ra_latency.py 1 > /dev/null 2>&1 < /dev/null
f=$(find /tmp/kt6 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $root/gpurun_out/r06_kernel_stats_ra_latency.csv && head -12 $root/gpurun_out/r06_kernel_stats_ra_latency.csv | cut -c1-160
