#!/bin/bash
# round 6, session 1 (GPU box, repo root): epoch cells — parity selection, one-box A/B against round 5's library and against clear-per-frame,
# the write-side counter with and without them; fresh counters of the hash-chain finder (level 9 @ 256 KiB)
root=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "epoch or sub_batch or (compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384)) or (opt_in and (epochs or cleared)) or short_last_frame or flag_sweep" < /dev/null 2>&1 | tail -5 ) > gpurun_out/r06_s1_tests.txt
cat gpurun_out/r06_s1_tests.txt
bash tools/ab.sh -v r5 -v A -v A:ZRA_MF_EPOCH=0 -r 3 -o r06_ab_epoch.txt
for e in 4 0; do
  ZRA_MF_EPOCH=$e bash tools/pmc_pass.sh r06_pmc_w_e$e "WRITE_SIZE" 2 > gpurun_out/r06_pmc_w_e$e.txt 2>&1
  ZRA_MF_EPOCH=$e bash tools/pmc_pass.sh r06_pmc_f_e$e "FETCH_SIZE" 2 > gpurun_out/r06_pmc_f_e$e.txt 2>&1
done
grep -h "dfast" gpurun_out/r06_pmc_[wf]_e*.txt
bash tools/pmc_hc.sh; cp gpurun_out/pmc_hc.txt gpurun_out/r06_pmc_hc.txt
