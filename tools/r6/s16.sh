#!/bin/bash
# round 6, session 16: the entropy workgroup in 14 LDS pieces (histograms share storage with the tables) and 19 match-finder waves per CU —
# parity selection, then one box alternating: 18 waves (the old geometry on the new layout), 19 waves, round 5's library
export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 2000 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "sub_batch or compress_buffer_bit_exact or short_last_frame or epoch or (opt_in and (entropy or dfast)) or corruption_statuses or streaming" < /dev/null 2>&1 | grep -E "passed|failed" | tail -3 ) > gpurun_out/r06_s16_tests.txt; cat gpurun_out/r06_s16_tests.txt
bash tools/ab.sh -v A:ZRA_MF_WAVES=18+ZRA_CHAIN_G=6 -v A -v A:ZRA_MF_WAVES=19+ZRA_CHAIN_G=3 -v r5 -r 4 -o r06_ab_w19.txt
