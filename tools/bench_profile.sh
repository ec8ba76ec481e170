#!/bin/bash
# run on the GPU box from the repo root: the headline bench line, then the same command under rocprofv3 --kernel-trace --stats;
# results in gpurun_out/ as bench_$1.json, bench_$1_under_rocprof.json, kernel_stats_$1.csv
tag=${1:-x}
root=$(pwd); export TMPDIR=/tmp
mkdir -p $root/gpurun_out
timeout 900 python3 $root/bench.py > $root/gpurun_out/bench_$tag.json 2> $root/gpurun_out/bench_$tag.err < /dev/null
tail -c 600 $root/gpurun_out/bench_$tag.json
cd /tmp; rm -rf /tmp/kst_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst_$tag -o k -- python3 $root/bench.py --no-cpu-baseline > $root/gpurun_out/bench_${tag}_under_rocprof.json 2>/dev/null < /dev/null
f=$(find /tmp/kst_$tag -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $root/gpurun_out/kernel_stats_$tag.csv && head -8 $root/gpurun_out/kernel_stats_$tag.csv | cut -c1-150
