#!/bin/bash
# round 5: the measurements of record for the final tree + the two-rank launch line of bench.py on the one GPU
tag=${1:-r05_c}
bash tools/measure.sh $tag suite
bash tools/bench_2rank_dry.sh 2>&1 | tail -6 | cut -c1-700
