"""bring-up: static instruction counts per source line of zra_mf_kernel (hipcc -gline-tables-only --save-temps)."""
import re, collections, subprocess, sys, os, tempfile
src_path = os.path.abspath(sys.argv[1]); lo, hi = int(sys.argv[2]), int(sys.argv[3]); kern = sys.argv[4] if len(sys.argv) > 4 else "zra_mf_kernel"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp(); 
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-gline-tables-only", "-I" + root + "/include", "-I" + root + "/zra_amd/csrc",
                "--save-temps", "-c", src_path, "-o", "x.o"], cwd=d, stderr=subprocess.DEVNULL, check=True)
asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
src = open(src_path).read().split("\n"); base = os.path.basename(src_path)
files = {}; cur = None; cnt = collections.Counter(); scnt = collections.Counter(); inK = False; tot = 0
for l in open(os.path.join(d, asm)):
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2))
    if l.startswith(kern + ":"): inK = True
    if inK and "s_endpgm" in l: inK = False
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    if inK and re.match(r"\s+[sv]_|\s+(global|flat|ds|buffer)_", l) and cur:
        cnt[cur] += 1; tot += 1
        if re.match(r"\s+s_", l): scnt[cur] += 1
mine = [k for k, v in files.items() if v.endswith(base)]
sub = 0
for (f, ln), c in sorted(cnt.items()):
    if f in mine and lo <= ln <= hi:
        print("%4d %4d(s%4d) %s" % (ln, c, scnt[(f, ln)], src[ln - 1][:120])); sub += c
print("range total", sub, "kernel total", tot)
