#!/bin/bash
# bring-up: the whole GPU suite, then compress throughput of the one-lane levels
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q < /dev/null > gpurun_out/gputest.txt 2>&1
grep -n "passed\|failed" gpurun_out/gputest.txt | tail -2
{
GIB=0.5 timeout 600 python tools/bringup/gpu_levels.py 11,65536 12,65536 10,16384 13,65536 16,65536 < /dev/null 2>&1
GIB=4 timeout 600 python tools/bringup/gpu_levels.py 3,1048576 1,1048576 5,1048576 < /dev/null 2>&1
} | grep -v amdgpu.ids | tee gpurun_out/levels.txt
