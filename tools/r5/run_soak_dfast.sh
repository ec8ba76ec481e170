#!/bin/bash
# round 5: levels 3-4 over arbitrary frame sizes, on the table kernel (flags, span, the three pipeline modes) and on the LDS-source kernel
b=${1:-0}
export SOAK_SEEDS=${SOAK_SEEDS:-0.5} SOAK_TIMEOUT=600
bash tools/soak.sh -b $b -o r5_soak_g.txt -e ZRA_MF_LS=0 dfast
bash tools/soak.sh -b $((b + 1000)) -o r5_soak_h.txt -e ZRA_MF_LS=0+ZRA_PIPE=2 dfast
bash tools/soak.sh -b $((b + 2000)) -o r5_soak_i.txt -e ZRA_MF_LS=0+ZRA_PIPE=0 dfast
bash tools/soak.sh -b $((b + 3000)) -o r5_soak_j.txt dfast
