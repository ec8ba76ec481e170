#!/bin/bash
# round 5: the tests added this round, on their own
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout 1700 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider -k "rccl_messages or error_exit_between or flag_sweep or two_half_batches or runs_past_its_input or opt_in_kernels and (sequence or resident or lds-table)" > $out/r5_newtests.txt 2>&1
tail -5 $out/r5_newtests.txt
