#!/bin/bash
# bring-up: hash-chain levels — parity subset, then throughput, with the product build and with other compile-time variants
# (HC_FLAGS="-DX=1 -DY=2": one rebuild + measurement per word)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "differential_compress or far or match_finder or larger_than" < /dev/null 2>&1 | tail -3
  GIB=2 timeout 300 python tools/bringup/gpu_levels.py 9,262144 7,65536 5,65536 6,65536 < /dev/null
  for k in $HC_FLAGS; do
    ZRA_EXTRA_CFLAGS=$k timeout 300 python zra_amd/build.py --force > /dev/null 2>&1 < /dev/null
    echo "variant $k"
    GIB=2 timeout 300 python tools/bringup/gpu_levels.py 9,262144 6,65536 < /dev/null
  done
} > gpurun_out/hc_try.txt 2>&1
grep -v amdgpu.ids gpurun_out/hc_try.txt | tail -30
