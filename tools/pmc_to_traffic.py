"""profiles/traffic.json entry from a tools/pmc_bench.sh output (FETCH_SIZE and WRITE_SIZE passes of the headline bench run).
usage: python tools/pmc_to_traffic.py gpurun_out/<pmc file> <entry key> [frames] ; prints the entry and merges it into profiles/traffic.json.
The kernel-source hash recorded with the entry is the one of the tree the measurement ran on (bench.py compares it with the tree it runs on)."""
import ast, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, root)
import bench
src, key = sys.argv[1], sys.argv[2]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
acc = {}
for line in open(src):
    m = re.match(r"\S+ (zra_\w+) launches=(\d+) (\{.*\})", line.strip())
    if not m:
        continue
    k, n, d = m.group(1), int(m.group(2)), ast.literal_eval(m.group(3))
    e = acc.setdefault(k, {"launches": n})
    for c, v in d.items():
        e["fetch_kib" if c == "FETCH_SIZE" else "write_kib" if c == "WRITE_SIZE" else c] = float(v)
entry = {"source": "tools/pmc_bench.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (one pass each, nothing else traced) of 'bench.py --steps 1 --warmup 0 --no-cpu-baseline'",
         "raw": "profiles/" + os.path.basename(src), "workload_key": "L3_fs65536", "frames": frames, "kernel_source_sha": bench.kernel_source_sha(),
         "ra_mode": "whole frames, XXH64 verified (the timed RA leg since round 6)",
         "calibration": "profiles/r02_pmc_calibration.txt: FETCH_SIZE counts 64 B per narrow random read request, WRITE_SIZE 32 B per partial-line store and 64 B per full line"}
dec = {}
for k, e in acc.items():
    if "fetch_kib" in e and "write_kib" in e:
        e["hbm_bytes_per_launch"] = int((e["fetch_kib"] + e["write_kib"]) * 1024 / max(1, e["launches"]))
    if k.startswith("zra_dec_") or k.startswith("zra_ra_"):
        # per decode launch of the run (the timed 1M x 4 KiB pass, the size classes, the probes): per-launch averages
        dec[k] = {"launches": e["launches"], "fetch_kib": e.get("fetch_kib", 0) / max(1, e["launches"]), "write_kib": e.get("write_kib", 0) / max(1, e["launches"])}
    else:
        entry[k] = e
entry["decode_one_pass_of_16GiB"] = dec
print(json.dumps(entry, indent=1))
tp = os.path.join(root, "profiles", "traffic.json")
t = json.load(open(tp))
t[key] = entry
json.dump(t, open(tp, "w"), indent=1)
