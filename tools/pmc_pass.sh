#!/bin/bash
# usage: tools/pmc_pass.sh NAME "COUNTER1 COUNTER2 ..." [GiB]   (run on the GPU box, from the repo root; one --pmc pass, nothing else traced)
name=$1; ctrs=$2; gib=${3:-1}
root=$(pwd); export TMPDIR=/tmp; cd /tmp
ZRA_ENC_SERIAL=1 timeout 150 rocprofv3 --pmc $ctrs --output-format csv -d $root/gpurun_out/$name -o p -- python3 $root/tools/bringup/gpu_compress_once.py $gib > $root/gpurun_out/$name.log 2>&1 < /dev/null
cd $root; python3 tools/pmc_summarize.py gpurun_out/$name
