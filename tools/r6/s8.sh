#!/bin/bash
# round 6, session 8: the decoder's block-parallel pass — parity (forced onto every call: damaged archives, headers, random access), stage
# times at 256 KiB and 2 MiB frames with and without it; then the compress sub-batch size again
export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 2400 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "opt_in and (block-parallel or rounds-only)" < /dev/null 2>&1 | grep -E "passed|failed|Error|error" | tail -8 ) > gpurun_out/r06_s8_tests.txt; cat gpurun_out/r06_s8_tests.txt
: > gpurun_out/r06_dec_fmb.txt
for v in "ZRA_DEC_FMB=1 262144" "ZRA_DEC_FMB=0 262144" "ZRA_DEC_FMB=1 2097152" "ZRA_DEC_FMB=0 2097152" "ZRA_DEC_FMB=1 1048576" "ZRA_DEC_FMB=1 65536"; do
  set -- $v; echo "== $1 frame $2" >> gpurun_out/r06_dec_fmb.txt
  env $1 timeout 300 python3 tools/bringup/gpu_dec_bench.py 8 $2 d 2>&1 | grep -v amdgpu.ids | tail -2 >> gpurun_out/r06_dec_fmb.txt
done
cat gpurun_out/r06_dec_fmb.txt
( timeout 900 python3 -m pytest tests/test_gpu_multi.py -x -q -m gpu -p no:cacheprovider < /dev/null 2>&1 | grep -E "passed|failed" | tail -3 )
bash tools/ab.sh -v A -v A:ZRA_ENC_SUB=4096 -v A:ZRA_ENC_SUB=2048 -v A:ZRA_ENC_SUB=1024 -r 3 -o r06_ab_sub_b.txt
