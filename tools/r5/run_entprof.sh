#!/bin/bash
# round 5: where the entropy stage's time goes (profile build: per-phase s_memtime of thread 0, barriers in), beside the match finder and alone
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE timeout 600 python3 zra_amd/build.py --force > $out/r5_prof_build.log 2>&1 < /dev/null
echo "== entropy stage under the match finder (default), 2 GiB" > $out/r5_entprof.txt
timeout 300 python3 tools/bringup/gpu_mf_profile.py 2 2>&1 | grep -v amdgpu.ids >> $out/r5_entprof.txt
echo "== entropy stage alone (ZRA_ENT_DEFER=1), 2 GiB" >> $out/r5_entprof.txt
ZRA_ENT_DEFER=1 timeout 300 python3 tools/bringup/gpu_mf_profile.py 2 2>&1 | grep -v amdgpu.ids >> $out/r5_entprof.txt
echo "== with the in-wave flags (ZRA_MF_FLAGS=2), 2 GiB" >> $out/r5_entprof.txt
ZRA_MF_FLAGS=2 timeout 300 python3 tools/bringup/gpu_mf_profile.py 2 2>&1 | grep -v amdgpu.ids >> $out/r5_entprof.txt
cat $out/r5_entprof.txt
