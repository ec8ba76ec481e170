#!/bin/bash
# run on the GPU box from the repo root: what the wave-cooperative hash-chain finder waits on — SQ and TCC counters of one level 9 @ 256 KiB
# compress (1 GiB, tools/bringup/gpu_levels.py), one --pmc pass per counter group, nothing else traced; sums per kernel in gpurun_out/pmc_hc.txt
root=$(pwd); export TMPDIR=/tmp; cd /tmp
: > $root/gpurun_out/pmc_hc.txt
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmchc_$i
  GIB=1 timeout 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmchc_$i -o p -- python3 $root/tools/bringup/gpu_levels.py 9,262144 > /tmp/pmchc_$i.log 2>&1 < /dev/null
  python3 $root/tools/pmc_summarize.py /tmp/pmchc_$i | grep zra_mf_hc >> $root/gpurun_out/pmc_hc.txt
done
cat $root/gpurun_out/pmc_hc.txt
