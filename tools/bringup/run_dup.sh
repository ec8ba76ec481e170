#!/bin/bash
# A = window duplicate detection in parallel rounds (+ LDS-source latency mode), B = HEAD before
root=$(pwd); out=$root/gpurun_out/dup.txt; mkdir -p $root/gpurun_out; : > $out
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact or short_last or match_finder or differential_compress or edge or sub_batch or small_inputs or opt_in or c3 or baseline_size" < /dev/null 2>&1 | tail -5 ) >> $out
for r in 1 2 3; do
  for lib in A B; do
    L=$root/zra_amd/libzra_amd.so; [ $lib != A ] && L=$root/zra_amd/libzra_amd_$lib.so
    echo -n "$lib: " >> $out
    timeout 600 python3 tools/bringup/gpu_mf_sweep.py "ZRA_AMD_BRINGUP=1;ZRA_AMD_LIB=$L" 2>&1 < /dev/null | tail -1 >> $out
  done
done
for lib in A B; do
  L=$root/zra_amd/libzra_amd.so; [ $lib != A ] && L=$root/zra_amd/libzra_amd_$lib.so
  echo "$lib lone:" >> $out
  ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L timeout 300 python3 tools/bringup/gpu_small_compress2.py 2>&1 < /dev/null | grep "frames" >> $out
done
echo "A lone, LS off:" >> $out
ZRA_MF_LS=0 timeout 300 python3 tools/bringup/gpu_small_compress2.py 2>&1 < /dev/null | grep " 1 frames\| 16 frames\| 152 frames\| 512 frames" >> $out
cat $out
