#!/bin/bash
# round 5: the measurements of record for a tree — bench line, the same command under rocprofv3 --kernel-trace --stats, the two PMC passes
tag=${1:-r05_a}
root=$(pwd); export TMPDIR=/tmp
bash tools/bench_profile.sh $tag
bash tools/pmc_bench.sh ${tag}_pmc_bench16g.txt
