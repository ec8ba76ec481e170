#!/bin/bash
# round 5: fresh seeds over the code new this round — the flag / span / verify-on-use dfast kernel forced onto small inputs (ZRA_MF_LS=0) in the
# three pipeline modes, and the decoder with the LDS-table chain kernel (two-byte cells, bitstream ring) ALONE on every job
b=${1:-50000}
export SOAK_SEEDS=${SOAK_SEEDS:-0.4}
bash tools/soak.sh -b $b -o r5_soak_a.txt -e ZRA_MF_LS=0 compress compress2
bash tools/soak.sh -b $((b + 1000)) -o r5_soak_b.txt -e ZRA_MF_LS=0+ZRA_PIPE=2 compress
bash tools/soak.sh -b $((b + 2000)) -o r5_soak_c.txt -e ZRA_MF_LS=0+ZRA_PIPE=0+ZRA_MF_SPAN=0 compress2
SOAK_SEEDS=0.2 bash tools/soak.sh -b $((b + 3000)) -o r5_soak_d.txt -e ZRA_DEC_SMALL_MAX=0+ZRA_DEC_CHAIN_LDS_MIN=1+ZRA_DEC_CHAIN_LDS=2 corrupt ra_damage headers
bash tools/soak.sh -b $((b + 4000)) -o r5_soak_e.txt -e ZRA_DEC_SMALL_MAX=0+ZRA_DEC_CHAIN_LDS_MIN=1+ZRA_DEC_CHAIN_LDS=2 compress2
