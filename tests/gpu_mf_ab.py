"""bring-up: first sequence where the default match finder differs from the previous formulation (ZRA_MF_TUNE=7)."""
import sys, os, struct
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, corpus as C
import zra_amd as Z
eng = Z.Engine(0); dev = torch.device("cuda", 0)
data = {"C": C.gen_C(1 << 19), "E": C.gen_E(1 << 20), "D": C.gen_D(1 << 18), "F": C.gen_struct(1 << 18)}
def run(d, lvl, fs, tune):
    if tune: os.environ["ZRA_MF_TUNE"] = str(tune)
    else: os.environ.pop("ZRA_MF_TUNE", None)
    t = torch.from_numpy(np.frombuffer(d, dtype=np.uint8).copy()).to(dev)
    out = torch.empty(Z.GetOutputBufferSize(len(d), fs) + 64, dtype=torch.uint8, device=dev)
    n = eng.compress(t.data_ptr(), len(d), out.data_ptr(), lvl, fs, True)
    nf = (len(d) + fs - 1) // fs
    return bytes(out[:n].cpu().numpy()), [eng.debug_read_seqs(f) for f in range(nf)]
shown = 0
for name, d in data.items():
    for lvl, fs in ((3, 65536), (3, 16384), (4, 131072)):
        a0, s0 = run(d, lvl, fs, 7); a1, s1 = run(d, lvl, fs, 0)
        if a0 == a1: print(name, lvl, fs, "same"); continue
        bad = [f for f in range(len(s0)) if s0[f] != s1[f]]
        print(name, lvl, fs, "DIFF frames", bad[:8], flush=True)
        if shown >= 4: continue
        shown += 1
        f = bad[0]; q0, m0 = s0[f]; q1, m1 = s1[f]
        print("   meta old", m0, "new", m1)
        k = next((i for i in range(min(len(q0), len(q1))) if q0[i] != q1[i]), min(len(q0), len(q1)))
        pos = sum(a + b for a, b, c in q0[:k])
        print("   first differing seq", k, "at frame pos", pos)
        for i in range(max(0, k - 3), min(len(q0), k + 4)): print("     old", i, q0[i], "   new", q1[i] if i < len(q1) else None)
