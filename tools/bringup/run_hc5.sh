#!/bin/bash
root=$(pwd); mkdir -p $root/gpurun_out
ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE timeout 300 python3 zra_amd/build.py --force > gpurun_out/hc_prof_build.log 2>&1 < /dev/null
{
  LOGLIKE=1 timeout 300 python3 tools/bringup/gpu_hc_profile.py 9 262144 2 < /dev/null
  LOGLIKE=1 timeout 300 python3 tools/bringup/gpu_hc_profile.py 5 65536 2 < /dev/null
  timeout 300 python3 tools/bringup/gpu_hc_profile.py 9 262144 2 < /dev/null
} 2>&1 | grep -v amdgpu.ids > gpurun_out/hc5_prof.txt
cat gpurun_out/hc5_prof.txt
