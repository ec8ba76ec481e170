#!/bin/bash
# bring-up: profile build of the library on the GPU box (scratch copy), then the hash-chain phase report for both corpora
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE timeout 300 python zra_amd/build.py --force > gpurun_out/hc_prof_build.log 2>&1 < /dev/null
{
  timeout 300 python tools/bringup/gpu_hc_profile.py 9 262144 2 < /dev/null
  LOGLIKE=1 timeout 300 python tools/bringup/gpu_hc_profile.py 9 262144 2 < /dev/null
  timeout 300 python tools/bringup/gpu_hc_profile.py 6 65536 2 < /dev/null
} > gpurun_out/hc_prof.txt 2>&1
tail -20 gpurun_out/hc_prof.txt
