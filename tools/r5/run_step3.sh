#!/bin/bash
# round 5, late: the flag sweep's fix — regression test, then fresh seeds of both generators on the table kernel; chain waves at 256 KiB frames
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "flag_sweep" -p no:cacheprovider 2>&1 | tail -4
SOAK_SEEDS=0.6 SOAK_TIMEOUT=700 bash tools/soak.sh -b 70000 -o r5_soak_f.txt -e ZRA_MF_LS=0 compress compress2
: > $out/r5_dec2.txt
for v in "X=0" "ZRA_DEC_CHAIN_WAVES=1" "ZRA_DEC_CHAIN_WAVES=4"; do
  echo "== $v frame 262144" >> $out/r5_dec2.txt
  env $v timeout 300 python3 tools/bringup/gpu_dec_bench.py 8 262144 d 2>&1 | grep "^decode" | tail -1 >> $out/r5_dec2.txt
done
cat $out/r5_dec2.txt
