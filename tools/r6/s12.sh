#!/bin/bash
# round 6, session 12: the tests added after the measurements of record (tiny compressed blocks; the smaller rounds-only selection), then every
# soak set on the final tree, fresh seeds
export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 1800 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "tiny_compressed_blocks or (opt_in and rounds-only) or epoch" < /dev/null 2>&1 | grep -E "passed|failed" | tail -3 ) > gpurun_out/r06_s12_tests.txt; cat gpurun_out/r06_s12_tests.txt
SOAK_SEEDS=0.6 SOAK_TIMEOUT=420 bash tools/soak.sh -b 125000 -o r06_soak_final.txt all > /dev/null
cat gpurun_out/r06_soak_final.txt
