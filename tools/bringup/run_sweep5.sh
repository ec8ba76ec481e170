#!/bin/bash
# waves per CU and duplicate-slot geometry of the dfast kernel after the round-detection rewrite (3 GiB compress each, second pass printed)
root=$(pwd); out=$root/gpurun_out/sweep5.txt; mkdir -p $root/gpurun_out; : > $out
for r in 1 2; do
for spec in "ZRA_MF_WAVES=18" "ZRA_MF_WAVES=20" "ZRA_MF_WAVES=22" "ZRA_MF_WAVES=16" "ZRA_MF_FILTER=1,2,7" "ZRA_MF_FILTER=1,2,9" "ZRA_MF_FILTER=1,2,9;ZRA_MF_WAVES=16" "ZRA_MF_FILTER=2,3,8;ZRA_MF_WAVES=22"; do
  echo -n "$spec: " >> $out
  timeout 300 python3 tools/bringup/gpu_mf_sweep.py "$spec" 2>&1 < /dev/null | tail -1 >> $out
done
done
cat $out
