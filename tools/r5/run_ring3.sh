#!/bin/bash
# round 5: sequence-chain stage, 8 GiB at 64 KiB frames: the tree (A: LDS-table kernel with the bitstream ring, 30 frames per CU) against the
# build copied to libzra_amd_B.so (round 4's: no ring, 31 frames), both kernels side by side (default) and the LDS-table kernel alone
root=$(pwd); out=$root/gpurun_out/r5_ring3.txt; mkdir -p $root/gpurun_out; : > $out; export TMPDIR=/tmp
for r in 1 2; do
for v in "A X=0" "B X=0" "A ZRA_DEC_CHAIN_LDS=2" "B ZRA_DEC_CHAIN_LDS=2"; do
  set -- $v; L=$root/zra_amd/libzra_amd.so; [ $1 = B ] && L=$root/zra_amd/libzra_amd_B.so
  echo -n "$v: " >> $out
  env $2 ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L timeout 300 python3 tools/bringup/gpu_dec_bench.py 8 65536 d 2>&1 | grep "^decode" | tail -1 >> $out
done
done
cat $out
