#!/bin/bash
# what the entropy stage under the match finder costs it, and whether the process-to-process spread of the finder's launch time lives in
# (round 5: the knob ZRA_ENT_DEFER of round 4 became pipeline mode ZRA_PIPE=0 — match finder, then the entropy stage, nothing overlapped)
# that overlap: 8 processes alternating default / ZRA_PIPE=0 (entropy only behind the whole finder launch), 3 GiB each, then 2 x 2 at 16 GiB
root=$(pwd); out=$root/gpurun_out/defer.txt; mkdir -p $root/gpurun_out; : > $out
for r in 1 2 3 4 5 6; do
  for spec in "ZRA_X=0" "ZRA_PIPE=0"; do
    echo -n "$spec: " >> $out
    timeout 300 python3 tools/bringup/gpu_mf_sweep.py "$spec" 2>&1 < /dev/null | tail -1 >> $out
  done
done
for r in 1 2; do
  for spec in "-" "ZRA_PIPE=0"; do
    echo -n "16 GiB $spec: " >> $out
    if [ "$spec" = "-" ]; then timeout 300 python3 tools/bringup/gpu_speed.py 16 3 65536 2 2>&1 < /dev/null | tail -1 | cut -c1-200 >> $out
    else ZRA_PIPE=0 timeout 300 python3 tools/bringup/gpu_speed.py 16 3 65536 2 2>&1 < /dev/null | tail -1 | cut -c1-200 >> $out; fi
  done
done
cat $out
