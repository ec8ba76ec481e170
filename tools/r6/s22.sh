#!/bin/bash
# round 6, session 22: the content checksum kernel behind the finder's residency — pipeline parity selection, the headline call four times,
# then the bench line, the same under rocprofv3 and the PMC passes on the final tree
export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "sub_batch or short_last_frame or error_exit or (opt_in and (geometry or sequence or resident or one-wave)) or (compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-4096)) or streaming or two_engines" < /dev/null 2>&1 | grep -E "passed|failed" | tail -3 ) > gpurun_out/r06_s22_tests.txt; cat gpurun_out/r06_s22_tests.txt
bash tools/ab.sh -v A -r 4 -o r06_ab_ck.txt
bash tools/measure.sh r06_j
