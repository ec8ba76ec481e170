#!/bin/bash
# Is the dfast match finder bound by instruction issue or by latency? The same kernel with idle instructions added per sequence:
# B = +100 scalar, D = +200 scalar, C = +50 vector (libzra_amd_{B,C,D}.so built with -DZRA_MF_PAD_SALU / -DZRA_MF_PAD_VALU), A = none.
root=$(pwd); out=$root/gpurun_out/pad.txt; mkdir -p $root/gpurun_out; : > $out
for r in 1 2; do
  for lib in A B D C; do
    L=$root/zra_amd/libzra_amd.so; [ $lib != A ] && L=$root/zra_amd/libzra_amd_$lib.so
    echo -n "$lib: " >> $out
    timeout 600 python3 tools/bringup/gpu_mf_sweep.py "ZRA_AMD_BRINGUP=1;ZRA_AMD_LIB=$L" 2>&1 < /dev/null | tail -1 >> $out
  done
done
# lone frames: the latency mode with and without the padding
for lib in A B D C; do
  L=$root/zra_amd/libzra_amd.so; [ $lib != A ] && L=$root/zra_amd/libzra_amd_$lib.so
  echo "$lib lone:" >> $out
  ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L timeout 300 python3 tools/bringup/gpu_small_compress2.py 2>&1 < /dev/null | grep " 16 frames\| 512 frames" >> $out
done
cat $out
