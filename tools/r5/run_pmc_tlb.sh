#!/bin/bash
# round 5: address translation of the dfast match finder (alone, 4 GiB): UTCL1 requests / misses and stalls, at 18 / 22 / 32 waves per CU
root=$(pwd); export TMPDIR=/tmp; out=$root/gpurun_out/r5_pmc_tlb.txt; : > $out
for w in 18 22 32; do
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum GRBM_GUI_ACTIVE" "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_LRU_INFLIGHT_sum TCP_UTCL1_STALL_MISSFIFO_FULL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum GRBM_UTCL2_BUSY"; do
  rm -rf /tmp/pmt; cd /tmp
  ZRA_MF_WAVES=$w timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pmt -o p -- python3 $root/tools/bringup/gpu_compress_once.py 4 > /tmp/pmt.log 2>&1 < /dev/null
  cd $root; echo -n "waves $w: " >> $out; python3 tools/pmc_summarize.py /tmp/pmt | grep "dfast_fl\|entropy" | tr '\n' ' ' >> $out; echo >> $out
done
done
cat $out
