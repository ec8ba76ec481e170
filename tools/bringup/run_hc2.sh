#!/bin/bash
# A/B on one box: hash-chain finder with two links per chain slot (A = in-tree build) against the build copied to libzra_amd_B.so
root=$(pwd); out=$root/gpurun_out/hc2.txt; mkdir -p $root/gpurun_out; : > $out
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact or short_last or match_finder or differential_compress or larger_than" < /dev/null 2>&1 | tail -5 ) >> $out
for r in 1 2; do
  for cfg in "2 5 65536" "2 7 65536" "2 9 262144" "1 6 1048576"; do
    for lib in A B; do
      L=$root/zra_amd/libzra_amd.so; [ $lib = B ] && L=$root/zra_amd/libzra_amd_B.so
      echo -n "$lib [$cfg]: " >> $out
      ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L timeout 300 python3 tools/bringup/gpu_speed.py $cfg 3 < /dev/null 2>&1 | tail -1 >> $out
    done
  done
done
cat $out
