#!/bin/bash
root=$(pwd); out=$root/gpurun_out/v2lone.txt; mkdir -p $root/gpurun_out; : > $out
echo "default:" >> $out; timeout 300 python3 tools/bringup/gpu_small_compress2.py 2>&1 < /dev/null | grep "frames" >> $out
echo "LS off:" >> $out; ZRA_MF_LS=0 timeout 300 python3 tools/bringup/gpu_small_compress2.py 2>&1 < /dev/null | grep "frames" >> $out
echo "mask kernel (ZRA_MF_V2=1):" >> $out; ZRA_MF_V2=1 timeout 300 python3 tools/bringup/gpu_small_compress2.py 2>&1 < /dev/null | grep "frames" >> $out
cat $out
