#!/bin/bash
# round 4, late: differential compress / decode soak on fresh seeds over the code new since the last soak — two-link hash-chain search,
# duplicate detection in rounds, the LDS-source dfast kernel (default for these small inputs), the table kernel forced onto small inputs,
# far-offset inputs
root=$(pwd); mkdir -p $root/gpurun_out
( timeout 500 python3 tools/bringup/gpu_soak.py 6000 6300 < /dev/null
  ZRA_MF_LS=0 timeout 400 python3 tools/bringup/gpu_soak.py 6300 6500 < /dev/null
  timeout 400 python3 tools/bringup/gpu_soak.py 6500 6650 v2 < /dev/null
  ZRA_MF_LS_MAX=1000000 timeout 300 python3 tools/bringup/gpu_soak.py 6650 6750 < /dev/null ) 2>&1 | grep -v amdgpu.ids | grep "FAIL\|soak done\|Error" > $root/gpurun_out/soak5.txt
cat $root/gpurun_out/soak5.txt
