#!/bin/bash
# run on the GPU box from the repo root: A/B of ENVIRONMENT knobs on ONE box: for every word of AB_ENVS (e.g. "ZRA_MF_WAVES=16
# ZRA_MF_WAVES=20"; "-" = defaults; several variables in one word joined by ";") run the bench line twice without the CPU baseline
root=$(pwd); mkdir -p $root/gpurun_out; : > $root/gpurun_out/bench_ab_env.txt
for e in $AB_ENVS; do
  for i in 1 2; do
    if [ "$e" = "-" ]; then timeout 600 python3 bench.py --no-cpu-baseline 2>/dev/null < /dev/null > /tmp/ab_line.json
    else env $(echo "$e" | tr ";" " ") timeout 600 python3 bench.py --no-cpu-baseline 2>/dev/null < /dev/null > /tmp/ab_line.json; fi
    python3 -c "
import json
d=json.loads(open('/tmp/ab_line.json').read().strip().splitlines()[-1])
print('$e', 'value %.3f compress %.3f ra_us %.3f mf_ms %.1f ent_ms %.1f dec_ms %.1f' % (d['value'], d['compress_gibs'], d['ra_us_per_query'], d['roofline']['launch_ms'], d['roofline']['other_kernels_launch_ms']['zra_entropy_kernel'], d['roofline']['other_kernels_launch_ms']['zra_dec_parse+huf+chain+exec (one decode pass)']))" >> $root/gpurun_out/bench_ab_env.txt
  done
done
cat $root/gpurun_out/bench_ab_env.txt
