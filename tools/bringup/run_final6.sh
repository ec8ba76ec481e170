#!/bin/bash
# PMC passes of the bench -> profiles/traffic.json, the bench line, rocprofv3 kernel stats of the same command (no test suite)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-r04_final4}
bash tools/pmc_bench.sh r04_pmc_bench16g.txt 2>&1 | tail -14
python3 tools/pmc_to_traffic.py gpurun_out/r04_pmc_bench16g.txt bench16g_r04b > gpurun_out/traffic_entry_$tag.json 2>&1
cp profiles/traffic.json gpurun_out/traffic.json
timeout 900 python3 bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err < /dev/null
tail -c 300 gpurun_out/bench_$tag.json; echo
cd /tmp; rm -rf /tmp/kst_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst_$tag -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_${tag}_under_rocprof.json 2>/dev/null < /dev/null
f=$(find /tmp/kst_$tag -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$tag.csv && head -4 $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$tag.csv | cut -c1-150
