#!/bin/bash
# bring-up: the whole GPU suite, then compress throughput of the one-lane / hash-chain levels at 4 GiB
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q < /dev/null > gpurun_out/gputest.txt 2>&1
tail -3 gpurun_out/gputest.txt
GIB=4 timeout 600 python tests/gpu_levels.py 1,65536 2,65536 -1,65536 -5,65536 1,262144 3,65536 6,65536 9,262144 < /dev/null 2>&1 | grep -v amdgpu.ids | tee gpurun_out/levels.txt
