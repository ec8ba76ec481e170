"""bring-up: does the match finder's launch time change from one ENGINE (= pair of HIP streams) to the next inside one process?"""
import sys, os, time
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dev = torch.device("cuda", 0)
base = bench.synth_corpus(64 << 20, 1); fs = 65536; n = int(gib * (1 << 30))
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
keep = []
for rnd in range(8):
    eng = Z.Engine(0)
    ms = []
    for i in range(2):
        eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True); ms.append(eng.kernel_stats()["mf_ms"])
    print("engine %d: match finder %s ms" % (rnd, " ".join("%.1f" % m for m in ms)), flush=True)
    eng.release_scratch()
    if rnd % 2: keep.append(eng)       # some engines stay alive (their streams keep their queues), the others are destroyed
    else: del eng
