#!/bin/bash
# round 5: the entropy stage's FSE chains on 64 lanes per stream — compress-side parity, then the headline configuration with / without the in-wave flags
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "compress or sequences or sub_batch or short_last or larger_than_the_window or streaming" -p no:cacheprovider > $out/r5_chain_parity.txt 2>&1
tail -5 $out/r5_chain_parity.txt
: > $out/r5_chain_ab.txt
for r in 1 2; do
  for v in "" "ZRA_MF_FLAGS=2" "ZRA_MF_FLAGS=2 ZRA_MF_WAVES=20" "ZRA_MF_FLAGS=2 ZRA_MF_WAVES=22"; do
    echo "== $v" >> $out/r5_chain_ab.txt
    env $v timeout 300 python3 tools/r5/gpu_tele.py 16 2 2>/dev/null | cut -c1-120 >> $out/r5_chain_ab.txt
  done
done
cat $out/r5_chain_ab.txt
