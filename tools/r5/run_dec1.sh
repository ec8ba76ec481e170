#!/bin/bash
# round 5: the decoder's stages by frame size (8 GiB, level 3) and the sequence-chain stage with its two kernels apart
root=$(pwd); out=$root/gpurun_out/r5_dec1.txt; mkdir -p $root/gpurun_out; : > $out; export TMPDIR=/tmp
for v in "X=0 65536" "ZRA_DEC_CHAIN_LDS=0 65536" "ZRA_DEC_CHAIN_LDS=2 65536" "X=0 262144" "X=0 2097152" $EXTRA; do
  set -- $v; echo "== $1 frame $2" >> $out
  env $1 timeout 300 python3 tools/bringup/gpu_dec_bench.py 8 $2 d 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
done
cat $out
