#!/bin/bash
# round 5: the two-persistent-kernel pipeline — first contact (small, then 1 GiB, then 16 GiB), parity selection, speed
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
: > $out/r5_pipe1.txt
for g in 0.0625 1 16; do
  echo "== $g GiB" >> $out/r5_pipe1.txt
  ZRA_ENC_TRACE=1 timeout 120 python3 tools/r5/gpu_tele.py $g 2 2>&1 | grep -v amdgpu.ids | cut -c1-260 >> $out/r5_pipe1.txt || echo "FAILED or timed out ($?)" >> $out/r5_pipe1.txt
done
cat $out/r5_pipe1.txt
sel="compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384 or 0-16384) or sub_batch_boundaries or short_last_frame or match_finder_sequences and (3-65536 or 3-16384) or randomised_differential_compress or streaming"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_pipe1_parity.txt 2>&1
tail -5 $out/r5_pipe1_parity.txt
ZRA_MF_LS=0 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_pipe1_parity2.txt 2>&1
tail -5 $out/r5_pipe1_parity2.txt
