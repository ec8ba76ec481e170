#!/bin/bash
# round 5: compress-side parity selection, then the headline call (16 GiB, level 3, 64 KiB) once per variant of EXTRA_VARIANTS in a process
# of its own, with the stream timeline (ZRA_ENC_TRACE) and the launch telemetry (waves per CU / XCD, entropy workgroups): the script behind
# most of profiles/r05_experiments.md. EXTRA_VARIANTS = words of '+'-joined environment settings, e.g.
#   EXTRA_VARIANTS="ZRA_PIPE=0 ZRA_PIPE=2 ZRA_MF_WAVES=20+ZRA_ENT_WGS=8" tools/r5/gpu.sh 1500 /tmp/c.log tools/r5/run_check.sh
# REPS (default 1) processes per variant (the process-to-process spread); SKIP_PARITY=1 leaves the tests out.
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
sel="compress_buffer_bit_exact and (3-65536 or 4-65536 or 3-16384 or 0-16384 or 9-65536 or 1-65536 or 13-) or sub_batch_boundaries or short_last_frame or randomised_differential_compress or streaming or match_finder_sequences and (3-65536 or 3-16384)"
if [ -z "$SKIP_PARITY" ]; then
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$sel" -p no:cacheprovider > $out/r5_check_parity.txt 2>&1
tail -3 $out/r5_check_parity.txt
fi
: > $out/r5_check.txt
for r in $(seq 1 ${REPS:-1}); do
for v in "X=0" $EXTRA_VARIANTS; do
echo "== $v" >> $out/r5_check.txt
env $(echo $v | tr '+' ' ') ZRA_ENC_TRACE=1 timeout 200 python3 tools/r5/gpu_tele.py 16 2 2>&1 | grep -v amdgpu.ids | cut -c1-2600 >> $out/r5_check.txt
done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r5_check.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        try:
            d = json.loads(l); t = d["tele"]; e = t.get("entropy") or {}
            print("  wall %.1f mf %.1f ent %.1f | waves/CU %s per SIMD %s | ent wgs %s cus %s wait %.3f ms/frame %.3f" % (d["wall_ms"], d["mf_ms"], d["ent_ms"], t.get("waves_per_cu_hist"), t.get("waves_per_simd_hist"), e.get("workgroups"), e.get("cus"), e.get("waiting_frac", 0), e.get("ms_per_frame", 0)))
        except Exception as ex: print("parse", ex, l[:200])
PY
