#!/bin/bash
# generic A/B on one box: A = in-tree build, B = zra_amd/libzra_amd_B.so; parity subset first, then 3 GiB throughput x3 alternating, then small calls
root=$(pwd); out=$root/gpurun_out/ab2.txt; mkdir -p $root/gpurun_out; : > $out
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${AB_TESTS:-bit_exact or short_last or match_finder or differential_compress or sub_batch}" < /dev/null 2>&1 | tail -3 ) >> $out
for r in 1 2 3; do
  for lib in A B; do
    L=$root/zra_amd/libzra_amd.so; [ $lib != A ] && L=$root/zra_amd/libzra_amd_$lib.so
    echo -n "$lib: " >> $out
    timeout 600 python3 tools/bringup/gpu_mf_sweep.py "ZRA_AMD_BRINGUP=1;ZRA_AMD_LIB=$L" 2>&1 < /dev/null | tail -1 >> $out
  done
done
for lib in A B; do
  L=$root/zra_amd/libzra_amd.so; [ $lib != A ] && L=$root/zra_amd/libzra_amd_$lib.so
  echo "$lib small calls:" >> $out
  ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L timeout 300 python3 tools/bringup/gpu_small_compress2.py 2>&1 < /dev/null | grep " 1 frames\| 152 frames\| 512 frames\| 1024 frames\| 4096 frames" >> $out
done
cat $out
for r in 1 2; do
  for cfg in "2 5 65536" "2 7 65536" "2 9 262144"; do
    for lib in A B; do
      L=$root/zra_amd/libzra_amd.so; [ $lib = B ] && L=$root/zra_amd/libzra_amd_B.so
      echo -n "$lib [$cfg]: " >> $out
      ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L timeout 300 python3 tools/bringup/gpu_speed.py $cfg 3 < /dev/null 2>&1 | tail -1 | cut -c1-110 >> $out
    done
  done
done
cat $out
