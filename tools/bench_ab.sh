#!/bin/bash
# run on the GPU box from the repo root: A/B of compile-time variants on ONE box (box-to-box spread is +-2 %): for every word of
# AB_FLAGS (e.g. "-DZRA_ENT_WAVES=7 -DZRA_ENT_WAVES=5") rebuild and run the bench line twice without the CPU baseline
root=$(pwd); mkdir -p $root/gpurun_out; : > $root/gpurun_out/bench_ab.txt
for f in $AB_FLAGS; do
  ZRA_EXTRA_CFLAGS="${f//;/ }" timeout 300 python zra_amd/build.py --force > /dev/null 2>&1 < /dev/null
  for i in 1 2; do
    timeout 600 python3 bench.py --no-cpu-baseline 2>/dev/null < /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f', 'value %.3f compress %.3f ra_us %.3f mf_ms %.1f ent_ms %.1f' % (d['value'], d['compress_gibs'], d['ra_us_per_query'], d['roofline']['launch_ms'], d['roofline']['other_kernels_launch_ms']['zra_entropy_kernel']))" >> $root/gpurun_out/bench_ab.txt
  done
done
cat $root/gpurun_out/bench_ab.txt
