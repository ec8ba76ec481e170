#!/bin/bash
# round 6, session 7: sub-batch size with the split entropy stage (one box, alternating); host-pointer single-query latency with the
# page-locked small path (bench's ra_latency probe) against the old path (ZRA_HOST_SMALL=0)
export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/ab.sh -v A -v A:ZRA_ENC_SUB=4096 -v A:ZRA_ENC_SUB=2048 -v r5 -r 3 -o r06_ab_sub.txt
for v in 1 0; do
ZRA_HOST_SMALL=$v timeout 600 python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch, zra_amd as Z, bench
dev = torch.device("cuda", 0); eng = Z.Engine(0)
N = 1 << 30; fs = 65536
base = bench.synth_corpus(64 << 20, 1)
d_in = torch.from_numpy(np.resize(base, N)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(N, fs) + 64, dtype=torch.uint8, device=dev)
n = eng.compress(d_in.data_ptr(), N, d_arc.data_ptr(), 3, fs, True)
print("ZRA_HOST_SMALL=%s" % os.environ.get("ZRA_HOST_SMALL"), json.dumps(bench.ra_latency_probe(Z, eng, d_arc, n, d_in, N, 4096, torch)))
PY
done > gpurun_out/r06_host_latency.txt
cat gpurun_out/r06_host_latency.txt
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "error_table or ra_vs_bruteforce or damaged or random_access or streaming or corruption" < /dev/null 2>&1 | tail -3 )
