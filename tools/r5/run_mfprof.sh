#!/bin/bash
# round 5: where a frame's time goes in the dfast parse UNDER LOAD (profile build: s_memtime at phase boundaries, each behind s_waitcnt 0):
# 4 GiB, the default pipeline (18 waves per CU beside the entropy stage), the sequential one at 22 waves, and one wave per CU
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE timeout 600 python3 zra_amd/build.py --force > $out/r5_prof_build.log 2>&1 < /dev/null
: > $out/r5_mfprof.txt
for v in "X=0" "ZRA_PIPE=0" "ZRA_PIPE=0+ZRA_MF_WAVES=1" $EXTRA_VARIANTS; do
echo "== $v, 4 GiB" >> $out/r5_mfprof.txt
env $(echo $v | tr '+' ' ') timeout 300 python3 tools/bringup/gpu_mf_profile.py 4 2>&1 | grep -v amdgpu.ids | grep -B2 -A40 "^frames" | grep -v "^entropy\|^  [0-9]* [a-zA-Z(].*emit\|Huffman\|FSE\|tile\|literal\|seq code\|tail " >> $out/r5_mfprof.txt
done
cat $out/r5_mfprof.txt
