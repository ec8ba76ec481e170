"""bring-up: print the kernels that follow the k-th launch of a marker kernel in a rocprofv3 kernel trace (start offset us, duration us)"""
import csv, glob, sys
root, marker, picks = sys.argv[1], sys.argv[2], [int(x) for x in sys.argv[3].split(",")]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 9
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
if not files:
    sys.exit("no kernel trace under " + root)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
print(len(rows), "launches,", len(idx), "markers")
for p in picks:
    if p >= len(idx):
        continue
    k = idx[p]
    t0 = int(rows[k]["Start_Timestamp"])
    for r in rows[k:k + n]:
        print("%-36s start %9.1f us  dur %9.1f us" % (r["Kernel_Name"][:36], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print()
