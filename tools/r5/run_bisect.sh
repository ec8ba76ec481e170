#!/bin/bash
# round 5: one soak seed under knob variants (which part of the new dfast path breaks it?) — usage: run_bisect.sh SEED [generator]
seed=${1:-90047}; gen=$2
root=$(pwd); out=$root/gpurun_out/r5_bisect.txt; : > $out
for v in "ZRA_MF_LS=0" "ZRA_MF_LS=0 ZRA_MF_SPAN=0" "ZRA_MF_LS=0 ZRA_MF_FLAGS=0" "ZRA_MF_LS=0 ZRA_PIPE=0" "ZRA_MF_LS=0 ZRA_PIPE=2" "ZRA_MF_LS=0 ZRA_MF_SPAN=512" "ZRA_MF_LS=0 ZRA_MF_SPAN=0 ZRA_MF_FLAGS=0" "X=0" $EXTRA; do
  echo "== $v" >> $out
  env $v timeout 200 python3 tools/bringup/gpu_soak.py $seed $((seed + 1)) $gen 2>&1 | grep -v amdgpu.ids | grep "FAIL\|soak done" | head -3 >> $out
done
cat $out
