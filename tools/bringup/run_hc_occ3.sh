#!/bin/bash
root=$(pwd); out=$root/gpurun_out/hc_occ3.txt; : > $out
run() { echo -n "$1 [$2]: " >> $out; if [ "$1" = "default" ]; then timeout 300 python3 tools/bringup/gpu_speed.py $2 3 < /dev/null 2>&1 | tail -1 | cut -c1-110 >> $out; else ZRA_MF_LDS=$1 timeout 300 python3 tools/bringup/gpu_speed.py $2 3 < /dev/null 2>&1 | tail -1 | cut -c1-110 >> $out; fi; }
for c in "2 9 262144" "2 10 262144" "2 9 524288"; do run default "$c"; run 0 "$c"; run default "$c"; run 0 "$c"; done
( timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "c4 or (bit_exact and (9- or 10-))" < /dev/null 2>&1 | tail -3 ) >> $out
cat $out
