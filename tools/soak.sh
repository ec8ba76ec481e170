#!/bin/bash
# The soaks of tools/bringup/gpu_soak*.py behind one script (run on the GPU box from the repo root):
#   tools/soak.sh [-b SEED_BASE] [-e "ENV=1+ENV2=2"] [-o FILE] SET...
# SET = compress | compress2 (second generator) | dfast (levels 3-4 over arbitrary frame sizes) | corrupt | tiny | determinism | headers | ra_damage | all (every one of them)
# -e runs the sets with library knobs set (e.g. ZRA_ENC_POISON=1, ZRA_MF_LS=0, ZRA_PIPE=2); -b shifts the seed ranges so that a new round soaks
# seeds no earlier round has seen. One line per set ("soak done ... 0 failures"); FAIL lines are printed in full.
root=$(pwd); b=0; words=""; outn=soak.txt
while getopts "b:e:o:" o; do case $o in b) b=$OPTARG;; e) words=$(echo $OPTARG | tr '+' ' ');; o) outn=$OPTARG;; esac; done
shift $((OPTIND - 1)); sets="$@"; sc() { python3 -c "print(int($1 * ${SOAK_SEEDS:-1.0}))"; }; [ -z "$sets" ] && sets=all
[ "$sets" = all ] && sets="compress compress2 dfast corrupt tiny determinism headers ra_damage"
out=$root/gpurun_out/$outn; mkdir -p $root/gpurun_out; : > $out
run() { echo "== $1 [$words] seeds +$b" >> $out; shift; env $words timeout ${SOAK_TIMEOUT:-500} "$@" < /dev/null > /tmp/soak_one.txt 2>&1; local rc=$?
        grep "FAIL\|MISMATCH\|Error" /tmp/soak_one.txt | head -20 >> $out; grep -v amdgpu.ids /tmp/soak_one.txt | tail -1 >> $out; [ $rc -ne 0 ] && echo "   (exit code $rc: 124 = stopped by the ${SOAK_TIMEOUT:-500} s limit before its last seed)" >> $out; }
for s in $sets; do case $s in
  compress)    run $s python3 tools/bringup/gpu_soak.py $((30000 + b)) $((30000 + b + $(sc 400)));;
  compress2)   run $s python3 tools/bringup/gpu_soak.py $((40000 + b)) $((40000 + b + $(sc 400))) v2;;
  dfast)       run $s python3 tools/bringup/gpu_soak_dfast.py $((0 + b)) $((0 + b + $(sc 600)));;
  corrupt)     run $s python3 tools/bringup/gpu_soak_corrupt.py $((20000 + b)) $((20000 + b + $(sc 2000)));;
  tiny)        run $s python3 tools/bringup/gpu_soak_tiny.py $((77 + b)) 1500;;
  determinism) run $s python3 tools/bringup/gpu_soak_determinism.py 2 12;;
  headers)     run $s python3 tools/bringup/gpu_soak_headers.py $((1000 + b)) $((1000 + b + $(sc 3000)));;
  ra_damage)   run $s python3 tools/bringup/gpu_soak_ra_damage.py $((1000 + b)) $((1000 + b + $(sc 1500)));;
esac; done
cat $out
