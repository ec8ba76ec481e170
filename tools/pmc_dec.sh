#!/bin/bash
# run on the GPU box from the repo root: what the four decode kernels wait on — SQ and TCC counters of
# tools/bringup/gpu_dec_bench.py 4 (level 3 @ 64 KiB), one --pmc pass per counter group, nothing else traced; sums per kernel in gpurun_out/pmc_dec.txt
root=$(pwd); export TMPDIR=/tmp; cd /tmp
: > $root/gpurun_out/pmc_dec.txt
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcdec_$i
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcdec_$i -o p -- python3 $root/tools/bringup/gpu_dec_bench.py 4 > /tmp/pmcdec_$i.log 2>&1 < /dev/null
  python3 $root/tools/pmc_summarize.py /tmp/pmcdec_$i | grep zra_dec_ >> $root/gpurun_out/pmc_dec.txt
done
cat $root/gpurun_out/pmc_dec.txt
