"""Sums rocprofv3 --pmc counter_collection CSVs per kernel name (bring-up helper; output is copied into profiles/)."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.Counter()
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r.get("Dispatch_Id"))
            if key not in seen: seen.add(key); launches[k] += 1
        for k, v in acc.items():
            if k.startswith("zra_"): print(d.rstrip("/").split("/")[-1], k, "launches=%d" % launches[k], {a: "%.4g" % b for a, b in sorted(v.items())})
