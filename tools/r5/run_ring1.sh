#!/bin/bash
# round 5: the LDS-table chain kernel with its bitstream ring — parity with that kernel ALONE on the decode selections, then the stage times
# of the tree (A) against the build copied to libzra_amd_B.so, both chain kernels side by side (default) and the LDS-table kernel alone
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
sel="randomised_differential_decode or randomised_corruption_statuses or randomised_header_damage or random_access_on_damaged or ra_vs_bruteforce or golden_frames or libzstd_frames or multi_block"
ZRA_DEC_SMALL_MAX=0 ZRA_DEC_CHAIN_LDS_MIN=1 ZRA_DEC_CHAIN_LDS=2 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_ring_parity.txt 2>&1
tail -4 $out/r5_ring_parity.txt
bash tools/r5/run_ring3.sh
