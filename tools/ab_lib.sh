#!/bin/bash
# run on the GPU box from the repo root: A/B of two BUILDS on ONE box. zra_amd/libzra_amd.so (A) against zra_amd/libzra_amd_B.so (B, built
# with other flags and copied aside), alternating, AB_REPS times (default 2), each with the environment words of AB_ENVS ("-" = none;
# several variables in one word joined by ";"). Prints the dfast match finder's launch time and the compress time of 3 GiB.
root=$(pwd); out=$root/gpurun_out/ab_lib.txt; mkdir -p $root/gpurun_out; : > $out
for r in $(seq 1 ${AB_REPS:-2}); do
  for e in ${AB_ENVS:--}; do
    for lib in A B; do
      L=$root/zra_amd/libzra_amd.so; [ $lib = B ] && L=$root/zra_amd/libzra_amd_B.so
      if [ "$e" = "-" ]; then spec="ZRA_AMD_BRINGUP=1;ZRA_AMD_LIB=$L"; else spec="ZRA_AMD_BRINGUP=1;ZRA_AMD_LIB=$L;$e"; fi
      echo -n "$lib $e: " >> $out
      timeout 600 python3 tools/bringup/gpu_mf_sweep.py "$spec" 2>&1 | tail -1 >> $out
    done
  done
done
cat $out
