#!/bin/bash
# round 5, first GPU session: (1) which address-translation counters rocprofv3 knows on this chip, (2) the process-to-process spread of the
# match finder's launch time with the launch telemetry (effective shader clock, waves per CU / XCD, frames per XCD), (3) the stream timeline
# of one call (where the tail behind the launch goes)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L > $out/r5_counters.txt 2>&1 < /dev/null); grep -i -c . $out/r5_counters.txt
grep -i "utcl\|tlb\|translation" $out/r5_counters.txt | cut -c1-160 | head -40
: > $out/r5_state.txt
for r in 1 2 3 4 5 6 7 8; do
  timeout 300 python3 tools/r5/gpu_tele.py 16 2 >> $out/r5_state.txt 2> /tmp/tele_$r.err < /dev/null || tail -3 /tmp/tele_$r.err
done
cut -c1-900 $out/r5_state.txt
ZRA_ENC_TRACE=1 timeout 300 python3 tools/r5/gpu_tele.py 16 2 > $out/r5_trace.out 2> $out/r5_trace.err < /dev/null
grep -c . $out/r5_trace.err; tail -45 $out/r5_trace.err
