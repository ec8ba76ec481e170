#!/bin/bash
# round 5: the in-tree build (A) against zra_amd/libzra_amd_B.so (B = the build before the change under test), alternating processes on ONE box;
# compress-side parity selection of A first
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
sel="compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384 or 0-16384 or 4-131072 or 3-4096) or sub_batch_boundaries or short_last_frame or randomised_differential_compress or match_finder_sequences"
ZRA_MF_LS=0 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_abB_parity.txt 2>&1
tail -3 $out/r5_abB_parity.txt
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_abB_parity2.txt 2>&1
tail -3 $out/r5_abB_parity2.txt
: > $out/r5_abB.txt
for r in 1 2 3; do
  for L in libzra_amd_B.so libzra_amd.so; do
    env $EXTRA timeout 300 python3 tools/r5/gpu_ab_lib.py $root/zra_amd/$L 16 2 2>/dev/null | tail -1 >> $out/r5_abB.txt
  done
done
cat $out/r5_abB.txt
