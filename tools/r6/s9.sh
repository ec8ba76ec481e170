#!/bin/bash
# round 6, session 9: soaks on fresh seeds (base 110000) with the round's new code forced onto the soak's small inputs:
#   a  the split entropy stage + the table finder (epoch cells, wave-cooperative table builds) on every call
#   b  one resident match-finder wave per CU: every wave through many epochs
#   c  the decoder's block-parallel pass on every call: damaged archives, headers, random access on damaged frames, differential decode
#   d  everything at its defaults
export SOAK_SEEDS=${SOAK_SEEDS:-0.5} SOAK_TIMEOUT=420
bash tools/soak.sh -b 110000 -e ZRA_ENT_SPLIT=2+ZRA_MF_LS=0 -o r06_soak_a.txt compress compress2 dfast > /dev/null
bash tools/soak.sh -b 110400 -e ZRA_MF_LS=0+ZRA_MF_WAVES=1 -o r06_soak_b.txt compress dfast > /dev/null
bash tools/soak.sh -b 110000 -e ZRA_DEC_SMALL_MAX=0+ZRA_DEC_FMB_MIN=1 -o r06_soak_c.txt compress2 corrupt headers ra_damage > /dev/null
bash tools/soak.sh -b 111000 -o r06_soak_d.txt all > /dev/null
cat gpurun_out/r06_soak_a.txt gpurun_out/r06_soak_b.txt gpurun_out/r06_soak_c.txt gpurun_out/r06_soak_d.txt
