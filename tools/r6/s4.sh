#!/bin/bash
# round 6, session 4: the split entropy stage (front / chains with lane = (frame, stream) / back) — parity selection on every call size,
# then one-box A/B: round 5's library, this tree with the split off, this tree (split on for calls of 256 frames and more)
export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 2000 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "sub_batch or (compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384 or 3-4096)) or (opt_in and entropy) or short_last_frame or epoch or device_api_at_baseline" < /dev/null 2>&1 | tail -5 ) > gpurun_out/r06_s4_tests.txt
cat gpurun_out/r06_s4_tests.txt
bash tools/ab.sh -v r5 -v A:ZRA_ENT_SPLIT=0 -v A -r 3 -o r06_ab_split_a.txt
