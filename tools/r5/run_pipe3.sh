#!/bin/bash
# round 5: pipeline with the entropy stage's telemetry, the exact reset-point chains, code histograms folded into the gather; parity, speed, profile
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
sel="compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384 or 0-16384 or 9-65536 or 1-65536 or 13-) or sub_batch_boundaries or short_last_frame or match_finder_sequences and (3-65536 or 3-16384) or randomised_differential_compress or streaming"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_pipe3_parity.txt 2>&1
tail -3 $out/r5_pipe3_parity.txt
: > $out/r5_pipe3.txt
for v in "" "ZRA_MF_WAVES=20" "ZRA_MF_FLAGS=0"; do
  echo "== $v" >> $out/r5_pipe3.txt
  env $v ZRA_ENC_TRACE=1 timeout 200 python3 tools/r5/gpu_tele.py 16 2 2>&1 | grep -v amdgpu.ids | cut -c1-1500 >> $out/r5_pipe3.txt || echo "FAILED or timed out ($?)" >> $out/r5_pipe3.txt
done
grep -v "^{" $out/r5_pipe3.txt; python3 - <<'PY'
import json
for l in open("gpurun_out/r5_pipe3.txt"):
    if l.startswith("{"):
        try:
            d = json.loads(l); t = d["tele"]
            print(d["wall_ms"], d["mf_ms"], d["ent_ms"], "ent:", t.get("entropy"))
        except Exception as e: print("parse", e, l[:200])
PY
ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE timeout 600 python3 zra_amd/build.py --force > $out/r5_prof_build.log 2>&1 < /dev/null
echo "== profile build, 2 GiB, default" > $out/r5_entprof3.txt
timeout 300 python3 tools/bringup/gpu_mf_profile.py 2 2>&1 | grep -v amdgpu.ids >> $out/r5_entprof3.txt
tail -12 $out/r5_entprof3.txt
