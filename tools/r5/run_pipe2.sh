#!/bin/bash
# round 5: the two-persistent-kernel pipeline — small calls, parity selection; then the entropy stage's phases under the match finder (profile build)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
: > $out/r5_pipe2.txt
for g in 0.0009765625 0.0625 1; do
  echo "== $g GiB" >> $out/r5_pipe2.txt
  timeout 120 python3 tools/r5/gpu_tele.py $g 2 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $out/r5_pipe2.txt || echo "FAILED or timed out ($?)" >> $out/r5_pipe2.txt
done
cat $out/r5_pipe2.txt
sel="compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384 or 0-16384) or sub_batch_boundaries or short_last_frame or match_finder_sequences and (3-65536 or 3-16384) or randomised_differential_compress or streaming"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_pipe2_parity.txt 2>&1
tail -5 $out/r5_pipe2_parity.txt
ZRA_MF_LS=0 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_pipe2_parity2.txt 2>&1
tail -5 $out/r5_pipe2_parity2.txt
ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE timeout 600 python3 zra_amd/build.py --force > $out/r5_prof_build.log 2>&1 < /dev/null
echo "== profile build, 2 GiB, default" > $out/r5_entprof2.txt
timeout 300 python3 tools/bringup/gpu_mf_profile.py 2 2>&1 | grep -v amdgpu.ids >> $out/r5_entprof2.txt
tail -14 $out/r5_entprof2.txt
