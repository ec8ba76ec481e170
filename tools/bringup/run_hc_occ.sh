#!/bin/bash
# hash-chain finder against resident waves per CU (ZRA_MF_LDS pads each wave's LDS: 3 KiB static + the padding)
root=$(pwd); out=$root/gpurun_out/hc_occ.txt; : > $out
for lds in 0 2048 4096 7168 12288 17408; do
  for cfg in "2 5 65536" "2 9 262144"; do
    echo -n "ZRA_MF_LDS=$lds [$cfg]: " >> $out
    ZRA_MF_LDS=$lds timeout 300 python3 tools/bringup/gpu_speed.py $cfg 3 < /dev/null 2>&1 | tail -1 | cut -c1-110 >> $out
  done
done
cat $out
