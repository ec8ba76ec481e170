#!/bin/bash
# C4's one-GPU share (8 GiB log-like, level 9, 256 KiB frames): second match-finder stream (default) against ZRA_MF_ONE_STREAM=1; then parity
cd $GRAFT_REPO_ROOT; out=gpurun_out/c4b.txt; : > $out
for r in 1 2; do
  for e in default one; do
    echo -n "$e: " >> $out
    if [ $e = default ]; then LOGLIKE=1 timeout 600 python3 tools/bringup/gpu_speed.py 8 9 262144 3 2>&1 < /dev/null | grep compress | tail -2 | cut -c1-60 | tr '\n' ' ' >> $out
    else LOGLIKE=1 ZRA_MF_ONE_STREAM=1 timeout 600 python3 tools/bringup/gpu_speed.py 8 9 262144 3 2>&1 < /dev/null | grep compress | tail -2 | cut -c1-60 | tr '\n' ' ' >> $out; fi
    echo >> $out
  done
done
echo -n "level 5 @ 64 KiB 8 GiB default: " >> $out; timeout 600 python3 tools/bringup/gpu_speed.py 8 5 65536 3 2>&1 < /dev/null | grep compress | tail -1 | cut -c1-60 >> $out
echo -n "level 5 @ 64 KiB 8 GiB one stream: " >> $out; ZRA_MF_ONE_STREAM=1 timeout 600 python3 tools/bringup/gpu_speed.py 8 5 65536 3 2>&1 < /dev/null | grep compress | tail -1 | cut -c1-60 >> $out
( timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "c4 or larger_than or (bit_exact and (1- or 9- or 13- or 19-)) or c5" < /dev/null 2>&1 | tail -3 ) >> $out
cat $out
