"""bring-up: phase times of the pre-pass's link step (profile build): thread 0 of workgroup 0, s_memtime (100 MHz) ticks."""
import sys, os, ctypes
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
lib = ctypes.CDLL(Z.LIB_PATH)
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1); fs = 65536; n = 256 << 20
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
buf = (ctypes.c_ulonglong * 32)()
eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
lib.ZraHipDebugReadLkProfile(buf, 0)
v = list(buf)
names = ["0 hash + head read", "1 claim (2 barriers)", "2 in-block grouping (LDS rounds)", "3 ballot loop + masks", "4 links + head + succ (2 barriers)"]
tot = sum(v[26:31])
print("pre-pass link step, workgroup 0, both tables of its frames: %d ticks of 10 ns; ballot-loop iterations %d" % (tot, v[31]))
for i, nm in enumerate(names): print("  %-40s %10d  %5.1f %%" % (nm, v[26 + i], 100.0 * v[26 + i] / max(tot, 1)))
