#!/bin/bash
# round 5: sequential (ZRA_PIPE=0: entropy stage behind the match finder, 2-5 workgroups per CU) against the resident entropy stage
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
: > $out/r5_ab7.txt
for v in "" "ZRA_PIPE=0 ZRA_ENT_WGS=5 ZRA_MF_WAVES=24" "ZRA_PIPE=0 ZRA_ENT_WGS=5 ZRA_MF_WAVES=22" "ZRA_PIPE=0 ZRA_ENT_WGS=4 ZRA_MF_WAVES=22" "ZRA_PIPE=0 ZRA_ENT_WGS=3 ZRA_MF_WAVES=22" "ZRA_PIPE=0 ZRA_ENT_WGS=6 ZRA_MF_WAVES=22" "ZRA_PIPE=0 ZRA_ENT_WGS=5 ZRA_MF_WAVES=22 ZRA_MF_FLAGS=0" ""; do
echo "== $v" >> $out/r5_ab7.txt
env $v ZRA_ENC_TRACE=1 timeout 200 python3 tools/r5/gpu_tele.py 16 2 2>&1 | grep -v amdgpu.ids | cut -c1-1600 >> $out/r5_ab7.txt
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r5_ab7.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        try:
            d = json.loads(l); t = d["tele"]; e = t.get("entropy") or {}
            print("  wall %.1f mf %.1f ent %.1f | ent wgs %s cus %s wait %.3f ms/frame %.3f" % (d["wall_ms"], d["mf_ms"], d["ent_ms"], e.get("workgroups"), e.get("cus"), e.get("waiting_frac", 0), e.get("ms_per_frame", 0)))
        except Exception as ex: print("parse", ex, l[:200])
PY
