#!/bin/bash
# bring-up: host-pointer and streaming rates of the drop-in API through the CLI's benchmark mode on a 4 GiB file of the bench corpus
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 - <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, bench
base = bench.synth_corpus(64 << 20, seed=1)
with open("/tmp/zra_host.bin", "wb") as f:
    for i in range(${1:-64}):
        f.write(base.tobytes())
PY
for c in ${2:-1024}; do
  echo "== chunk $c MiB"
  ZRA_HOST_CHUNK_MIB=$c timeout 600 $R/zra_amd/tools/zratool_amd b /tmp/zra_host.bin 3 65536 10 < /dev/null
done
rm -f /tmp/zra_host.bin*
