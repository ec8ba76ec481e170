#!/bin/bash
# run on the GPU box from the repo root: A/B of compile-time variants of the DECODER on one box: per word of AB_FLAGS rebuild, then
# (several flags in one word joined by ';')
# tools/bringup/gpu_dec_bench.py 8 under rocprofv3 --kernel-trace --stats (per-kernel averages of the same run)
root=$(pwd); mkdir -p $root/gpurun_out; export TMPDIR=/tmp; : > $root/gpurun_out/dec_ab.txt
for f in $AB_FLAGS; do
  ZRA_EXTRA_CFLAGS="${f//;/ }" timeout 300 python zra_amd/build.py --force > /dev/null 2>&1 < /dev/null
  echo "== $f" >> $root/gpurun_out/dec_ab.txt
  rm -rf /tmp/decab; cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/decab -o k -- python3 $root/tools/bringup/gpu_dec_bench.py 8 2>/dev/null < /dev/null | grep "decode \|RA " >> $root/gpurun_out/dec_ab.txt
  cd $root
  k=$(find /tmp/decab -name '*kernel_stats.csv' | head -1)
  [ -n "$k" ] && grep "zra_dec_" "$k" | cut -d, -f1,2,4 >> $root/gpurun_out/dec_ab.txt
done
cat $root/gpurun_out/dec_ab.txt
