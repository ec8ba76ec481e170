#!/bin/bash
# round 6, session 13: table stores with the sc1 policy (build B: -DZRA_MF_SC1ST) — parity on B, then A / B alternating on one box
export TMPDIR=/tmp; mkdir -p gpurun_out; root=$(pwd)
( ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$root/zra_amd/libzra_amd_B.so timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "sub_batch or (compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384)) or short_last_frame or epoch or flag_sweep" < /dev/null 2>&1 | grep -E "passed|failed" | tail -3 ) > gpurun_out/r06_s13_tests.txt; cat gpurun_out/r06_s13_tests.txt
bash tools/ab.sh -v A -v B -v C -r 4 -o r06_ab_sc1.txt
