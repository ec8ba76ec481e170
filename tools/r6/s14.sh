#!/bin/bash
# round 6, session 14: the seeds the final soak listed (corrupt: status 20 against libzstd's 22; RA on damaged frames: bytes), on this tree, on
# round 5's library and with the page-locked small host path off — old or new?
export TMPDIR=/tmp; mkdir -p gpurun_out; root=$(pwd); out=gpurun_out/r06_soak_bisect2.txt; : > $out
for v in "A X=1" "r5 X=1" "A ZRA_HOST_SMALL=0"; do
  set -- $v; L=$root/zra_amd/libzra_amd.so; [ "$1" != A ] && L=$root/zra_amd/libzra_amd_$1.so
  for s in 145238 146031; do
    echo "== [$v] corrupt seed $s" >> $out
    env ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L $2 timeout 200 python3 tools/bringup/gpu_soak_corrupt.py $s $((s+1)) 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
  done
  for s in 126115 126728 126757 126887; do
    echo "== [$v] ra_damage seed $s" >> $out
    env ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L $2 timeout 200 python3 tools/bringup/gpu_soak_ra_damage.py $s $((s+1)) 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
  done
done
cat $out
bash tools/r6/s13.sh
