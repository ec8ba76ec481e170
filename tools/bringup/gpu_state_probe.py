"""bring-up: does the match finder's launch time (the two 'box states') change when the engine's scratch is released and allocated again
inside ONE process? (if it does, the state is a property of where the allocations land, not of the box)"""
import sys, os, time
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1); fs = 65536; n = int(gib * (1 << 30))
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
junk = []
for rnd in range(6):
    ms = []
    for i in range(2):
        eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True); ms.append(eng.kernel_stats()["mf_ms"])
    print("allocation round %d: match finder %s ms" % (rnd, " ".join("%.1f" % m for m in ms)), flush=True)
    eng.release_scratch()
    junk.append(torch.empty((rnd + 1) * 97 << 20, dtype=torch.uint8, device=dev))     # shift where the next allocations land
