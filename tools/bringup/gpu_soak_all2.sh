#!/bin/bash
# bring-up: the long soaks on the final build, seed ranges shifted by SOAK_BASE
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
b=${SOAK_BASE:-0}
{
  timeout 300 python tools/bringup/gpu_soak_headers.py $((1000 + b)) $((4000 + b)) < /dev/null 2>&1 | tail -3
  timeout 300 python tools/bringup/gpu_soak_ra_damage.py $((1000 + b)) $((2500 + b)) < /dev/null 2>&1 | tail -3
  timeout 300 python tools/bringup/gpu_soak_corrupt.py $((20000 + b)) $((22000 + b)) < /dev/null 2>&1 | tail -3
  timeout 420 python tools/bringup/gpu_soak.py $((30000 + b)) $((30400 + b)) < /dev/null 2>&1 | tail -2
  timeout 420 python tools/bringup/gpu_soak.py $((40000 + b)) $((40400 + b)) v2 < /dev/null 2>&1 | tail -2
} > gpurun_out/soak_final2.txt 2>&1
grep -v amdgpu.ids gpurun_out/soak_final2.txt
