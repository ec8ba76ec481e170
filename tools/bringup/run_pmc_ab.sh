#!/bin/bash
# the same PMC passes over two builds on one box: A = in-tree, B = zra_amd/libzra_amd_B.so (ZRA_AMD_LIB); 1 GiB, dfast, serial mode
root=$(pwd); export TMPDIR=/tmp; out=$root/gpurun_out/pmc_ab.txt; : > $out
for lib in A B; do
  L=$root/zra_amd/libzra_amd.so; [ $lib = B ] && L=$root/zra_amd/libzra_amd_B.so
  for set in "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
    rm -rf /tmp/pab; cd /tmp
    ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L ZRA_ENC_SERIAL=1 timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pab -o p -- python3 $root/tools/bringup/gpu_compress_once.py 1 > /tmp/pab.log 2>&1 < /dev/null
    cd $root; echo -n "$lib: " >> $out; python3 tools/pmc_summarize.py /tmp/pab | grep dfast >> $out
  done
done
cat $out
