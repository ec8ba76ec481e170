#!/bin/bash
# round 5: the in-wave bucket flags (ZRA_MF_FLAGS=2) — parity selection of the compress side, then speed A/B on one box
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
sel="compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384 or 0-16384) or sub_batch_boundaries or short_last_frame or match_finder_sequences and (3-65536 or 3-16384) or randomised_differential_compress"
ZRA_MF_FLAGS=2 ZRA_MF_LS=0 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_flags_parity.txt 2>&1
tail -5 $out/r5_flags_parity.txt
: > $out/r5_flags_ab.txt
for r in 1 2; do
  for v in "" "ZRA_MF_FLAGS=2" "ZRA_MF_FLAGS=2 ZRA_MF_WAVES=20" "ZRA_MF_FLAGS=2 ZRA_MF_WAVES=22"; do
    echo "== $v" >> $out/r5_flags_ab.txt
    env $v timeout 300 python3 tools/r5/gpu_tele.py 16 2 2>/dev/null | cut -c1-120 >> $out/r5_flags_ab.txt
  done
done
cat $out/r5_flags_ab.txt
