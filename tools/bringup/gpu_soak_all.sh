#!/bin/bash
# bring-up: the three soaks one after the other on the final build (differential compress/decode with both generators,
# damaged archives against libzstd, determinism of the persistent pipeline); SOAK_BASE shifts the seed ranges
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
b=${SOAK_BASE:-0}
{
  timeout 400 python tools/bringup/gpu_soak.py $((7000 + b)) $((7400 + b)) v2 < /dev/null 2>&1 | tail -2
  timeout 400 python tools/bringup/gpu_soak.py $((9000 + b)) $((9400 + b)) < /dev/null 2>&1 | tail -2
  timeout 300 python tools/bringup/gpu_soak_corrupt.py $((3000 + b)) $((3400 + b)) < /dev/null 2>&1 | tail -3
  timeout 200 python tools/bringup/gpu_soak_tiny.py $((77 + b)) 1500 < /dev/null 2>&1 | tail -2
  timeout 200 python tools/bringup/gpu_soak_determinism.py 2 12 < /dev/null 2>&1 | tail -2
} > gpurun_out/soak_final.txt 2>&1
grep -v amdgpu.ids gpurun_out/soak_final.txt
