#!/bin/bash
# round 5, late: two-byte cells in the LDS-table chain kernel — parity with that kernel alone + stage times (A/B), then the round's soak on fresh seeds
bash tools/r5/run_ring1.sh
if grep -q "failed\|Error\|Aborted" gpurun_out/r5_ring_parity.txt; then echo "PARITY FAILED - no soak"; exit 1; fi
bash tools/r5/run_soak_r5.sh 60000
