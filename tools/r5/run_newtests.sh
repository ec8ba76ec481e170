#!/bin/bash
# round 5: the tests added this round (RCCL pieces, error exit with two contexts, all C3 frames, pipeline modes), then one headline check
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout 1700 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider -k "rccl_messages or error_exit_between or opt_in_kernels and (sequence or resident) or config_c3" > $out/r5_newtests.txt 2>&1
tail -5 $out/r5_newtests.txt
ZRA_ENC_TRACE=1 timeout 200 python3 tools/r5/gpu_tele.py 16 2 2>&1 | grep -v amdgpu.ids | cut -c1-600
