#!/bin/bash
# round 6, session 6: two ranks on the one GPU with the gather beside the serving (tests + the bench's 2-rank dry run), then the entropy
# stage's issue priority with the split stage (one box, alternating)
export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_gpu_multi.py -x -q -m gpu -p no:cacheprovider < /dev/null 2>&1 | tail -5 ) > gpurun_out/r06_s6_tests.txt; cat gpurun_out/r06_s6_tests.txt
bash tools/bench_2rank_dry.sh > gpurun_out/r06_2rank_dry.log 2>&1; tail -5 gpurun_out/r06_2rank_dry.log; cp gpurun_out/bench_2rank_dry.json gpurun_out/r06_bench_2rank_dry.json
ZRA_BENCH_GATHER_SYNC=1 bash tools/bench_2rank_dry.sh > gpurun_out/r06_2rank_dry_sync.log 2>&1; cp gpurun_out/bench_2rank_dry.json gpurun_out/r06_bench_2rank_dry_sync.json
python3 - <<'PY'
import json
for f in ("r06_bench_2rank_dry", "r06_bench_2rank_dry_sync"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("gather_overlap"), d["config"]["transport"][:100])
    except Exception as e: print(f, "unreadable", e)
PY
bash tools/ab.sh -v A -v A:ZRA_ENT_PRIO=3 -v A:ZRA_ENT_PRIO=0 -v A:ZRA_ENC_SUB=4096 -r 2 -o r06_ab_prio.txt
