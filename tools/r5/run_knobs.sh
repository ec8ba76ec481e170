#!/bin/bash
# round 5: knobs of the default pipeline on ONE box, alternating: entropy issue priority, filter geometry, sub-batch size
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
: > $out/r5_knobs.txt
for r in 1 2; do
for v in "X=0" "ZRA_ENT_PRIO=0" "ZRA_ENT_PRIO=1" "ZRA_MF_FILTER=1,2,8" "ZRA_MF_FILTER=2,3,8" "ZRA_ENC_SUB=4096" "ZRA_ENC_SUB=16384" "ZRA_ENT_WGS=4" "ZRA_ENT_WGS=16"; do
  echo -n "$v: " >> $out/r5_knobs.txt; env $v timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd.so 16 2 2>/dev/null | tail -1 >> $out/r5_knobs.txt
done
done
cat $out/r5_knobs.txt
