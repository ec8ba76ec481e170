#!/bin/bash
# round 5: where does round 4's fast / slow launch state come from? r4 library, r5 in r4's match-finder configuration (18 waves, no flags,
# entropy stage behind the finder), r5 default — alternating processes on ONE box
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
: > $out/r5_ab_r4b.txt
for r in 1 2 3 4; do
  timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd_r4.so 16 2 2>/dev/null | tail -1 >> $out/r5_ab_r4b.txt
  echo -n "r5 18w noflags: " >> $out/r5_ab_r4b.txt; ZRA_MF_FLAGS=0 ZRA_MF_WAVES=18 timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd.so 16 2 2>/dev/null | tail -1 >> $out/r5_ab_r4b.txt
  echo -n "r5 18w flags:   " >> $out/r5_ab_r4b.txt; ZRA_MF_WAVES=18 timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd.so 16 2 2>/dev/null | tail -1 >> $out/r5_ab_r4b.txt
  echo -n "r5 default:     " >> $out/r5_ab_r4b.txt; timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd.so 16 2 2>/dev/null | tail -1 >> $out/r5_ab_r4b.txt
done
cat $out/r5_ab_r4b.txt
