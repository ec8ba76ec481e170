#!/bin/bash
# round 5: counters of the entropy stage alone (sequential pipeline, 2 GiB): instruction mix, issue / wait cycles, LDS, L1
root=$(pwd); export TMPDIR=/tmp; out=$root/gpurun_out/r5_pmc_ent.txt; : > $out
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pme; cd /tmp
  timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pme -o p -- python3 $root/tools/bringup/gpu_compress_once.py 2 > /tmp/pme.log 2>&1 < /dev/null
  cd $root; python3 tools/pmc_summarize.py /tmp/pme | grep "entropy\|dfast_fl" >> $out
done
cat $out
