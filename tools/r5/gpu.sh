#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3: nothing charged). usage: tools/r5/gpu.sh <timeout_s> <logfile> <command...>
t=$1; log=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > $log 2>&1; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
