#!/bin/bash
# round 6, session 11: the two seeds the soak found with the block-parallel pass forced onto every call (150028: compress2 round trip,
# 111269: random access on a damaged archive), with and without the pass; then the decode sets again on new seeds
export TMPDIR=/tmp; mkdir -p gpurun_out; out=gpurun_out/r06_soak_bisect.txt; : > $out
for e in "ZRA_DEC_SMALL_MAX=0 ZRA_DEC_FMB_MIN=1" "ZRA_DEC_SMALL_MAX=0 ZRA_DEC_FMB=0" "X=1"; do
  echo "== [$e] compress2 seed 150028" >> $out
  env $e timeout 300 python3 tools/bringup/gpu_soak.py 150028 150029 v2 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
  echo "== [$e] ra_damage seed 111269" >> $out
  env $e timeout 300 python3 tools/bringup/gpu_soak_ra_damage.py 111269 111270 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
done
cat $out
SOAK_SEEDS=0.5 SOAK_TIMEOUT=420 bash tools/soak.sh -b 113000 -e ZRA_DEC_SMALL_MAX=0+ZRA_DEC_FMB_MIN=1 -o r06_soak_e.txt compress2 corrupt headers ra_damage > /dev/null
cat gpurun_out/r06_soak_e.txt
