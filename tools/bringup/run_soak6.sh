#!/bin/bash
# round 4, last: differential compress soak on fresh seeds over the four-link hash-chain finder, half of it over a poisoned table scratch
root=$(pwd); mkdir -p $root/gpurun_out
( timeout 400 python3 tools/bringup/gpu_soak.py 7000 7200 < /dev/null
  ZRA_ENC_POISON=1 timeout 400 python3 tools/bringup/gpu_soak.py 7200 7400 < /dev/null
  ZRA_ENC_POISON=1 timeout 300 python3 tools/bringup/gpu_soak.py 7400 7500 v2 < /dev/null ) 2>&1 | grep -v amdgpu.ids | grep "FAIL\|soak done\|Error" > $root/gpurun_out/soak6.txt
cat $root/gpurun_out/soak6.txt
