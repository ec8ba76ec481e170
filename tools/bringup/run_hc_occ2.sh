#!/bin/bash
root=$(pwd); out=$root/gpurun_out/hc_occ2.txt; : > $out
run() { echo -n "ZRA_MF_LDS=$1 [$2]: " >> $out; ZRA_MF_LDS=$1 timeout 300 python3 tools/bringup/gpu_speed.py $2 3 < /dev/null 2>&1 | tail -1 | cut -c1-110 >> $out; }
for lds in 0 5120 6144 7168 8192 9216; do run $lds "2 9 262144"; done
for lds in 0 7168; do run $lds "2 6 262144"; run $lds "2 9 65536"; run $lds "2 8 1048576"; run $lds "2 10 262144"; done
for lds in 0 4096; do run $lds "2 7 65536"; done
cat $out
