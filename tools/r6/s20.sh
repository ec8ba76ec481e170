#!/bin/bash
# round 6, session 20: the hash-chain parse's repeat-offset tests by the whole wave (64 positions per round trip instead of 5) — parity of the
# levels 5-10 cases, then build H (the tree before) against the tree on one box: levels 5 / 7 / 9 at 64 KiB, levels 6 / 9 at 256 KiB, C4's data
export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 2000 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "(compress_buffer_bit_exact and (5- or 6- or 7- or 9- or 10-)) or short_last_frame or match_finder_sequences or randomised_differential_compress or poisoned or error_exit" < /dev/null 2>&1 | grep -E "passed|failed" | tail -3 ) > gpurun_out/r06_s20_tests.txt; cat gpurun_out/r06_s20_tests.txt
bash tools/ab.sh -v H -v A -c "2 5 65536" -c "2 7 65536" -c "2 9 65536" -c "2 6 262144" -c "2 9 262144" -r 2 -o r06_ab_hc.txt
bash tools/ab.sh -v H:LOGLIKE=1 -v A:LOGLIKE=1 -c "2 9 262144" -c "2 5 262144" -r 2 -o r06_ab_hc_loglike.txt
