#!/bin/bash
root=$(pwd); mkdir -p $root/gpurun_out
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact or short_last or match_finder or differential_compress or larger_than or edge or c4" < /dev/null 2>&1 | tail -5 ) > gpurun_out/hc3_tests.txt
{
  for cfg in "9 262144 2" "5 65536 2" "7 65536 2"; do timeout 300 python3 tools/bringup/gpu_hc_profile.py $cfg < /dev/null; done
} > /dev/null 2>&1
ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE timeout 300 python3 zra_amd/build.py --force > gpurun_out/hc_prof_build.log 2>&1 < /dev/null
{
  for cfg in "9 262144 2" "5 65536 2" "7 65536 2"; do timeout 300 python3 tools/bringup/gpu_hc_profile.py $cfg < /dev/null; done
} > gpurun_out/hc3_prof.txt 2>&1
cat gpurun_out/hc3_tests.txt gpurun_out/hc3_prof.txt
