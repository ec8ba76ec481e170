#!/bin/bash
# round 5 against round 4 on ONE box, alternating processes: zra_amd/libzra_amd_r4.so is the library built from the round-4 tree (d26905f)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
: > $out/r5_ab_r4.txt
for r in 1 2 3; do
  for L in libzra_amd_r4.so libzra_amd.so; do
    env $EXTRA timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/$L 16 3 2>/dev/null | tail -2 >> $out/r5_ab_r4.txt
  done
done
cat $out/r5_ab_r4.txt
