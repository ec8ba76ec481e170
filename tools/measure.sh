#!/bin/bash
# The measurements of record for a tree, on one box (run on the GPU box from the repo root): tools/measure.sh TAG [suite]
#   1. with "suite": the whole GPU test suite                                   -> gpurun_out/gputest_TAG.txt
#   2. the bench line                                                           -> gpurun_out/bench_TAG.json
#   3. the same command under rocprofv3 --kernel-trace --stats                  -> gpurun_out/kernel_stats_TAG.csv, bench_TAG_under_rocprof.json
#   4. the two PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench's timed region  -> gpurun_out/TAG_pmc_bench16g.txt
# Copy what is to be judged into profiles/ and add the traffic entry with tools/pmc_to_traffic.py (profiles/README.md).
tag=${1:-x}; root=$(pwd); export TMPDIR=/tmp; mkdir -p $root/gpurun_out
if [ "$2" = suite ]; then
  ( timeout 2400 python3 -m pytest tests -q -m gpu -p no:cacheprovider < /dev/null 2>&1 | tail -8 ) > $root/gpurun_out/gputest_$tag.txt; cat $root/gpurun_out/gputest_$tag.txt
fi
bash tools/bench_profile.sh $tag
bash tools/pmc_bench.sh ${tag}_pmc_bench16g.txt
