#!/bin/bash
root=$(pwd); mkdir -p $root/gpurun_out; out=$root/gpurun_out/ls.txt; : > $out
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact or short_last or match_finder or differential_compress or edge or sub_batch or small_inputs" < /dev/null 2>&1 | tail -5 ) >> $out
ZRA_MF_LS=0 timeout 300 python3 tools/bringup/gpu_small_compress2.py < /dev/null >> $out 2>&1
ZRA_MF_LS_MAX=100000 timeout 300 python3 tools/bringup/gpu_small_compress2.py < /dev/null >> $out 2>&1
cat $out
