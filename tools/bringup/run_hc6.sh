#!/bin/bash
# A/B on one box, hash-chain search: candidates' 16 bytes as one load (A) against two (B); text-like and log-like corpora
root=$(pwd); out=$root/gpurun_out/hc6.txt; mkdir -p $root/gpurun_out; : > $out
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact and (5- or 6- or 7- or 8- or 9- or 10-) or differential_compress or short_last or larger_than or c4 or poisoned" < /dev/null 2>&1 | tail -3 ) >> $out
for r in 1 2; do
  for cfg in "2 5 65536" "2 7 65536" "2 9 262144"; do
    for lib in A B; do
      L=$root/zra_amd/libzra_amd.so; [ $lib = B ] && L=$root/zra_amd/libzra_amd_B.so
      echo -n "$lib [$cfg]: " >> $out
      ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L timeout 300 python3 tools/bringup/gpu_speed.py $cfg 3 < /dev/null 2>&1 | tail -1 | cut -c1-100 >> $out
    done
  done
  for lib in A B; do
    L=$root/zra_amd/libzra_amd.so; [ $lib = B ] && L=$root/zra_amd/libzra_amd_B.so
    echo -n "$lib loglike [2 9 262144]: " >> $out
    LOGLIKE=1 ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L timeout 300 python3 tools/bringup/gpu_hc_profile.py 9 262144 2 < /dev/null 2>&1 | grep "GiB/s" | cut -c1-100 >> $out
  done
done
cat $out
