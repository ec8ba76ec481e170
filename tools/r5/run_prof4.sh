#!/bin/bash
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
sel="compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384 or 0-16384 or 9-65536 or 1-65536 or 13-) or sub_batch_boundaries or short_last_frame or randomised_differential_compress"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_prof4_parity.txt 2>&1
tail -3 $out/r5_prof4_parity.txt
ZRA_ENC_TRACE=1 timeout 200 python3 tools/r5/gpu_tele.py 16 2 2>&1 | grep -v amdgpu.ids | cut -c1-100 | awk '/^gat/ {n++; if (n%4==1) print; next} {print}' > $out/r5_prof4_trace.txt
tail -30 $out/r5_prof4_trace.txt
ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE timeout 600 python3 zra_amd/build.py --force > $out/r5_prof_build.log 2>&1 < /dev/null
echo "== profile build, 2 GiB, default" > $out/r5_entprof4.txt
timeout 300 python3 tools/bringup/gpu_mf_profile.py 2 2>&1 | grep -v amdgpu.ids >> $out/r5_entprof4.txt
tail -11 $out/r5_entprof4.txt
