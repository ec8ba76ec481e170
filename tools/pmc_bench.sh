#!/bin/bash
# run on the GPU box from the repo root: HBM-side byte counters of the headline bench run itself (16 GiB, one step), one --pmc pass per
# counter, nothing else traced; per-kernel sums and launch counts go to gpurun_out/$1
out=${1:-r04_pmc_bench16g.txt}
root=$(pwd); export TMPDIR=/tmp; cd /tmp
: > $root/gpurun_out/$out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcb_$c -o p -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --timed-only > /tmp/pmcb_$c.json 2>/dev/null < /dev/null
  python3 $root/tools/pmc_summarize.py /tmp/pmcb_$c >> $root/gpurun_out/$out
done
cat $root/gpurun_out/$out
