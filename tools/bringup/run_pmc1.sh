cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( cd /tmp && timeout 60 rocprofv3 -L 2>/dev/null | grep -o "\b\(TA\|TCP\|TD\|SQ\)_[A-Z0-9_a-z]*" | sort -u | tr '\n' ' ' ) > gpurun_out/pmc_list.txt
wc -c gpurun_out/pmc_list.txt
export ZRA_MF_FLAGS=0
bash tools/pmc_pass.sh pmc_sq1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" 1 2>&1 | grep -i "dfast" | head -3
bash tools/pmc_pass.sh pmc_sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES" 1 2>&1 | grep -i "dfast" | head -3
bash tools/pmc_pass.sh pmc_ta1 "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" 1 2>&1 | grep -i "dfast" | head -3
