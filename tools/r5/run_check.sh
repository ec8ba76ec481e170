#!/bin/bash
# round 5: compress-side parity selection, then the headline configuration (trace + telemetry)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
sel="compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384 or 0-16384 or 9-65536 or 1-65536 or 13-) or sub_batch_boundaries or short_last_frame or randomised_differential_compress or streaming or match_finder_sequences and (3-65536 or 3-16384)"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_check_parity.txt 2>&1
tail -3 $out/r5_check_parity.txt
: > $out/r5_check.txt
for v in "" $EXTRA_VARIANTS; do
echo "== $v" >> $out/r5_check.txt
env $(echo $v | tr ',' ' ') ZRA_ENC_TRACE=1 timeout 200 python3 tools/r5/gpu_tele.py 16 2 2>&1 | grep -v amdgpu.ids | cut -c1-1600 >> $out/r5_check.txt
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r5_check.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        try:
            d = json.loads(l); t = d["tele"]; e = t.get("entropy") or {}
            print("  wall %.1f mf %.1f ent %.1f | ent wgs %s cus %s wait %.3f ms/frame %.3f" % (d["wall_ms"], d["mf_ms"], d["ent_ms"], e.get("workgroups"), e.get("cus"), e.get("waiting_frac", 0), e.get("ms_per_frame", 0)))
        except Exception as ex: print("parse", ex, l[:200])
PY
