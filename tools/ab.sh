#!/bin/bash
# One box, several variants, alternating: the generic A/B of this repo (run on the GPU box from the repo root, e.g.
#   tools/r5/gpu.sh 1500 /tmp/ab.log tools/ab.sh -v A -v B -c "2 5 65536" -c "2 9 262144"
# A box differs from the next one by more than most changes do, and a process from the next one by 5-10 % (profiles/r05_experiments.md §1),
# so a change is judged only against the other build / knob setting on the same box, several processes each.
#   -v NAME            a variant. "A" = the in-tree build; any other name N = zra_amd/libzra_amd_N.so (a build copied aside; r4 = round 4's).
#                      NAME may carry environment words: "A:ZRA_MF_WAVES=20+ZRA_ENT_WGS=8"
#   -c "GiB L FS"      a configuration for tools/bringup/gpu_speed.py (bench corpus; LOGLIKE=1 in the variant's words for C4's data);
#                      without -c: the headline call (16 GiB, level 3, 64 KiB) through tools/r5/gpu_ab_lib.py, which prints wall / mf / entropy
#   -k "selection"     a pytest selection of tests/test_gpu_parity.py run first on the in-tree build (-x)
#   -r N               repetitions (default 2)
#   -o FILE            output under gpurun_out/ (default ab.txt)
root=$(pwd); reps=2; outn=ab.txt; sel=""; variants=(); cfgs=()
while getopts "v:c:k:r:o:" o; do case $o in v) variants+=("$OPTARG");; c) cfgs+=("$OPTARG");; k) sel=$OPTARG;; r) reps=$OPTARG;; o) outn=$OPTARG;; esac; done
[ ${#variants[@]} -eq 0 ] && variants=(A B)
out=$root/gpurun_out/$outn; mkdir -p $root/gpurun_out; : > $out; export TMPDIR=/tmp
if [ -n "$sel" ]; then ( timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "$sel" < /dev/null 2>&1 | tail -3 ) >> $out; fi
one() {   # variant, command...
  local v=$1; shift; local name=${v%%:*}; local words=""; [ "$v" != "$name" ] && words=$(echo "${v#*:}" | tr '+' ' ')
  local L=$root/zra_amd/libzra_amd.so; [ "$name" != A ] && L=$root/zra_amd/libzra_amd_$name.so
  local cmd=(); for a in "$@"; do [ "$a" = "@LIB@" ] && a=$L; cmd+=("$a"); done
  echo -n "$v [$CFG]: " >> $out
  env ZRA_AMD_BRINGUP=1 ZRA_AMD_LIB=$L $words timeout 400 "${cmd[@]}" < /dev/null 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 >> $out
}
for r in $(seq 1 $reps); do
  if [ ${#cfgs[@]} -eq 0 ]; then
    for v in "${variants[@]}"; do CFG="16 3 65536" one "$v" python3 tools/r5/gpu_ab_lib.py @LIB@ 16 2; done
  else
    for c in "${cfgs[@]}"; do for v in "${variants[@]}"; do CFG="$c" one "$v" python3 tools/bringup/gpu_speed.py $c 3; done; done
  fi
done
cat $out
