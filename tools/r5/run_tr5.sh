#!/bin/bash
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
: > $out/r5_tr5.txt
for v in "" "ZRA_ENT_WGS=2" "ZRA_MF_WAVES=19"; do
echo "== $v" >> $out/r5_tr5.txt
env $v ZRA_ENC_TRACE=1 timeout 200 python3 tools/r5/gpu_tele.py 16 2 2>&1 | grep -v amdgpu.ids | cut -c1-1600 | awk '/^gat/ {n++; if (n%8==1) print; next} {print}' >> $out/r5_tr5.txt
done
grep -v "^{" $out/r5_tr5.txt | tail -60; python3 - <<'PY'
import json
for l in open("gpurun_out/r5_tr5.txt"):
    if l.startswith("{"):
        try:
            d = json.loads(l); t = d["tele"]
            print(d["wall_ms"], d["mf_ms"], d["ent_ms"], "ent:", t.get("entropy"))
        except Exception as e: print("parse", e, l[:200])
PY
