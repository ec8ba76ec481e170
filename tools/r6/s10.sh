#!/bin/bash
# round 6, session 10: the measurements of record on the final tree (one box): GPU suite, bench line, the same under rocprofv3, PMC passes,
# the two-rank dry run
tag=${1:-r06_f}
bash tools/measure.sh $tag suite
bash tools/bench_2rank_dry.sh > gpurun_out/${tag}_2rank_dry.log 2>&1; cp gpurun_out/bench_2rank_dry.json gpurun_out/${tag}_bench_2rank_dry.json
tail -c 300 gpurun_out/${tag}_bench_2rank_dry.json
