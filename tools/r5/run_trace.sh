#!/bin/bash
# round 5: stream timeline of one 16 GiB call (match-finder launch, entropy launches per sub-batch), default and with the in-wave flags
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
for v in "" "ZRA_MF_FLAGS=2"; do
  echo "== $v" >> $out/r5_trace2.txt
  env $v ZRA_ENC_TRACE=1 timeout 300 python3 tools/r5/gpu_tele.py 16 2 2>&1 | grep -v amdgpu.ids | cut -c1-150 | awk 'NR<=4 || /mf / || (NR%6==0) || /gib/' >> $out/r5_trace2.txt
done
cat $out/r5_trace2.txt
