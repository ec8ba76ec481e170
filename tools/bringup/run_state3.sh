#!/bin/bash
# the process-to-process spread of the dfast match finder's launch time: does it follow WHERE the process's allocations land (virtual
# addresses of the table scratch / input / output)? 8 processes, 16 GiB each, hipMalloc results from the runtime's own log
root=$(pwd); out=$root/gpurun_out/state3.txt; mkdir -p $root/gpurun_out; : > $out
for r in 1 2 3 4 5 6 7 8; do
  AMD_LOG_LEVEL=3 AMD_LOG_MASK=1 timeout 300 python3 tools/bringup/gpu_speed.py 16 3 65536 2 > /tmp/st_$r.out 2> /tmp/st_$r.err < /dev/null
  echo "process $r: $(tail -1 /tmp/st_$r.out | cut -c1-140)" >> $out
  grep -i "hipMalloc\b\|hipMalloc " /tmp/st_$r.err | grep -i "return\|0x" | awk '{print "   ", $0}' | cut -c1-200 | head -40 >> $out
done
cat $out | head -150
