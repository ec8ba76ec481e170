#!/bin/bash
# round 5: the entropy stage alone (ZRA_PIPE=0), phases at 1 and at 5 workgroups per CU: which phase pays for the occupancy?
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE timeout 600 python3 zra_amd/build.py --force > $out/r5_prof_build.log 2>&1 < /dev/null
: > $out/r5_entprof5.txt
for w in 1 2 5; do
echo "== ZRA_PIPE=0 ZRA_ENT_WGS=$w ZRA_MF_WAVES=22, 4 GiB" >> $out/r5_entprof5.txt
ZRA_PIPE=0 ZRA_ENT_WGS=$w ZRA_MF_WAVES=22 timeout 300 python3 tools/bringup/gpu_mf_profile.py 4 2>&1 | grep -v amdgpu.ids | grep -A12 "^stats\|^entropy" | grep -v "^  [0-9]* [a-z].*[0-9] %$" >> $out/r5_entprof5.txt
ZRA_PIPE=0 ZRA_ENT_WGS=$w ZRA_MF_WAVES=22 timeout 300 python3 tools/bringup/gpu_mf_profile.py 4 2>&1 | grep -v amdgpu.ids | grep -A10 "^entropy" >> $out/r5_entprof5.txt
done
cat $out/r5_entprof5.txt
