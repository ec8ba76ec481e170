#!/bin/bash
# A/B on one box: hash-chain lazy parse with the 16-byte repcode cache + the post-match fetch riding on the backward extension (A) against B
root=$(pwd); out=$root/gpurun_out/hc4.txt; mkdir -p $root/gpurun_out; : > $out
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact and (5- or 6- or 7- or 9-) or differential_compress" < /dev/null 2>&1 | tail -3 ) >> $out
for r in 1 2; do
  for cfg in "2 5 65536" "2 7 65536" "2 9 262144"; do
    for lib in A B; do
      L=$root/zra_amd/libzra_amd.so; [ $lib = B ] && L=$root/zra_amd/libzra_amd_B.so
      echo -n "$lib [$cfg]: " >> $out
      ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L timeout 300 python3 tools/bringup/gpu_speed.py $cfg 3 < /dev/null 2>&1 | tail -1 | cut -c1-120 >> $out
    done
  done
done
cat $out
