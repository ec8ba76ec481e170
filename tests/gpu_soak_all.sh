#!/bin/bash
# bring-up: the three soaks one after the other on the final build (differential compress/decode with the far-offset generator,
# damaged archives against libzstd, determinism of the persistent pipeline)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
  timeout 420 python tests/gpu_soak.py 7000 7400 v2 < /dev/null 2>&1 | tail -2
  timeout 420 python tests/gpu_soak_corrupt.py 3000 3400 < /dev/null 2>&1 | tail -3
  timeout 300 python tests/gpu_soak_determinism.py 2 12 < /dev/null 2>&1 | tail -2
} > gpurun_out/soak_final.txt 2>&1
cat gpurun_out/soak_final.txt
