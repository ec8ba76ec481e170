#!/bin/bash
# round 5 against round 4 on ONE box, alternating processes: r4 library | r5 sequential (default) | r5 with round 4's overlap (ZRA_PIPE=1) at 18 / 20 waves
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
sel="compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384) or sub_batch_boundaries or short_last_frame or randomised_differential_compress"
ZRA_PIPE=1 ZRA_MF_LS=0 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_pipe1_parity.txt 2>&1
tail -3 $out/r5_pipe1_parity.txt
: > $out/r5_ab_r4d.txt
for r in 1 2 3; do
  timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd_r4.so 16 2 2>/dev/null | tail -1 >> $out/r5_ab_r4d.txt
  echo -n "r5 default:        " >> $out/r5_ab_r4d.txt; timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd.so 16 2 2>/dev/null | tail -1 >> $out/r5_ab_r4d.txt
  echo -n "r5 PIPE=1 18w:     " >> $out/r5_ab_r4d.txt; ZRA_PIPE=1 timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd.so 16 2 2>/dev/null | tail -1 >> $out/r5_ab_r4d.txt
  echo -n "r5 PIPE=1 20w:     " >> $out/r5_ab_r4d.txt; ZRA_PIPE=1 ZRA_MF_WAVES=20 timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd.so 16 2 2>/dev/null | tail -1 >> $out/r5_ab_r4d.txt
  echo -n "r5 PIPE=1 18w nofl:" >> $out/r5_ab_r4d.txt; ZRA_PIPE=1 ZRA_MF_FLAGS=0 timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd.so 16 2 2>/dev/null | tail -1 >> $out/r5_ab_r4d.txt
done
cat $out/r5_ab_r4d.txt
