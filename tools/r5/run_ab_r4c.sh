#!/bin/bash
# round 5: the clock levels and the power during round 4's launches, fast against slow processes (one box)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
ls /sys/class/drm/ > $out/r5_ab_r4c.txt 2>&1; ls /sys/class/drm/card*/device/ 2>/dev/null | grep -i "pp_dpm\|hwmon" | head >> $out/r5_ab_r4c.txt
for r in 1 2 3 4 5 6 7 8; do
  SAMPLE=1 timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/libzra_amd_r4.so 16 3 2>&1 | tail -3 >> $out/r5_ab_r4c.txt
done
cat $out/r5_ab_r4c.txt
