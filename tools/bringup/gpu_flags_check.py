"""bring-up: the pre-pass's bucket flags of frame 0 against a CPU computation."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch, zra_amd as Z, bench
os.environ["ZRA_PP_DUMP"] = "/tmp/flags0.bin"
eng = Z.Engine(0); dev = torch.device("cuda", 0)
N = 8 << 20; fs = 65536
c = bench.synth_corpus(64 << 20, 1)[:N]
t = torch.from_numpy(c.copy()).to(dev)
out = torch.empty(Z.GetOutputBufferSize(N, fs) + 64, dtype=torch.uint8, device=dev)
eng.compress(t.data_ptr(), N, out.data_ptr(), 3, fs, True)
fl = np.fromfile("/tmp/flags0.bin", dtype=np.uint8)[:fs]
src = c[:fs].astype(np.uint64)
last = fs - 8
v = np.zeros(fs, dtype=np.uint64)
for k in range(8): v[:last + 1] |= src[k:k + last + 1] << np.uint64(8 * k)
bl = ((v * np.uint64(0xCF1BBCDCB7A56463)) >> np.uint64(64 - 16)).astype(np.int64)
bs = (((v << np.uint64(24)) * np.uint64(889523592379)) >> np.uint64(64 - 15)).astype(np.int64)
exp = np.zeros(fs, dtype=np.uint8)
for tbl, b in ((0, bl), (1, bs)):
    first = {}; lastp = {}
    for p in range(1, last + 1):
        x = int(b[p])
        if x not in first: first[x] = p
        lastp[x] = p
    for p in range(1, last + 1):
        x = int(b[p])
        exp[p] |= ((1 if first[x] != p else 0) | ((1 if lastp[x] != p else 0) << 1)) << (2 * tbl)
exp[0] = 0x0F; exp[last + 1:] = 0x0F
bad = np.nonzero(fl != exp)[0]
print("flags frame 0: mismatches", len(bad), "of", fs, "| histogram gpu", np.bincount(fl, minlength=16)[:16].tolist(), "| expected", np.bincount(exp, minlength=16)[:16].tolist())
if len(bad): print("first mismatches", [(int(p), int(fl[p]), int(exp[p])) for p in bad[:10]])
