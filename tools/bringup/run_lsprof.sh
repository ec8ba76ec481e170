#!/bin/bash
root=$(pwd); mkdir -p $root/gpurun_out; out=$root/gpurun_out/lsprof.txt; : > $out
ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE timeout 300 python3 zra_amd/build.py --force > gpurun_out/lsprof_build.log 2>&1 < /dev/null
echo "== 16 frames, source in LDS" >> $out
timeout 300 python3 tools/bringup/gpu_mf_profile.py 0.0009765625 < /dev/null 2>&1 | grep -v "^entropy\|^  [0-9] [a-zA-Z]*.*emit\|amdgpu.ids" | head -16 >> $out
echo "== 16 frames, source in memory (ZRA_MF_LS=0)" >> $out
ZRA_MF_LS=0 timeout 300 python3 tools/bringup/gpu_mf_profile.py 0.0009765625 < /dev/null 2>&1 | grep -v amdgpu.ids | head -16 >> $out
echo "== 1 GiB, throughput mode" >> $out
timeout 300 python3 tools/bringup/gpu_mf_profile.py 1 < /dev/null 2>&1 | grep -v amdgpu.ids | head -16 >> $out
cat $out
