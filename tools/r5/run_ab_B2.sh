#!/bin/bash
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
sel="compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384 or 0-16384 or 9-65536 or 1-65536 or 13- or 4-131072) or sub_batch_boundaries or short_last_frame or randomised_differential_compress"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_abB_parity.txt 2>&1
tail -3 $out/r5_abB_parity.txt
: > $out/r5_abB.txt
for r in 1 2 3; do
  timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd_B.so 16 2 2>/dev/null | tail -1 >> $out/r5_abB.txt
  timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd.so 16 2 2>/dev/null | tail -1 >> $out/r5_abB.txt
  echo -n "A 19w: " >> $out/r5_abB.txt; ZRA_MF_WAVES=19 timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd.so 16 2 2>/dev/null | tail -1 >> $out/r5_abB.txt
  echo -n "A 20w: " >> $out/r5_abB.txt; ZRA_MF_WAVES=20 timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd.so 16 2 2>/dev/null | tail -1 >> $out/r5_abB.txt
done
cat $out/r5_abB.txt
