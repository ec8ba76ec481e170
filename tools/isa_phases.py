"""CPU: instruction mix between the phase marks of a kernel's assembly (hipcc -S -DZRA_MF_MARK: PROF(k) becomes the comment "; ZMARK k").
usage: isa_phases.py <kernel.s> [first_line last_line]   — static counts in layout order, a first look at where the instructions are."""
import sys, re, collections
lines = open(sys.argv[1]).read().split("\n")
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(lines)
cur = "start"; stats = collections.OrderedDict()
def cat(op):
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "lane"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("v_"): return "valu"
    return None
for i, l in enumerate(lines[lo:hi], lo):
    m = re.search(r"; ZMARK (\d+)", l)
    if m: cur = "after %s (line %d)" % (m.group(1), i + 1); continue
    t = l.strip().split()
    if not t or t[0].startswith(";") or t[0].startswith(".") or t[0].endswith(":"): continue
    c = cat(t[0])
    if c: stats.setdefault(cur, collections.Counter())[c] += 1
for k, v in stats.items():
    print("%-26s total %4d | %s" % (k, sum(v.values()), " ".join("%s %d" % (a, b) for a, b in sorted(v.items()))))
