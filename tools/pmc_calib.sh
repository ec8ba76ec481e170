#!/bin/bash
# run on the GPU box from the repo root: FETCH_SIZE / WRITE_SIZE / request counters of the calibration patterns, one --pmc pass each
root=$(pwd); export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf /tmp/calib_$n
  timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/calib_$n -o p -- $root/tools/pmc_calib > /tmp/calib.log 2>&1 < /dev/null
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/calib_$n/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    print(k, {a: "%.5g" % b for a, b in v.items()})
PY
done
grep calib /tmp/calib.log
