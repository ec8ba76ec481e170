"""bring-up: phase timers / counters of zra_dec_exec_kernel (library built with ZRA_EXTRA_CFLAGS=-DZRA_DEC_PROFILE)."""
import sys, os, ctypes
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
lib = ctypes.CDLL(Z.LIB_PATH)
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1); fs = 65536; n = int(gib * (1 << 30))
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
d_out = torch.empty(n, dtype=torch.uint8, device=dev)
buf = (ctypes.c_ulonglong * 16)()
eng.decompress(d_arc.data_ptr(), asz, d_out.data_ptr(), n)
lib.ZraHipDebugReadDecProfile(buf, 1)
eng.decompress(d_arc.data_ptr(), asz, d_out.data_ptr(), n)
print(eng.kernel_stats())
lib.ZraHipDebugReadDecProfile(buf, 1)
v = list(buf); nf = max(v[8], 1); ns = max(v[9], 1)
print("frames %d  steps/frame %.1f  seqs/step %.1f  chased %.1f %%  rounds/step %.2f" % (nf, ns / nf, v[10] / ns, 100.0 * v[11] / max(v[10], 1), v[12] / ns))
names = ["0 seq wait + scan", "1 chase", "2 literals + chased copies (+fence)", "3 dependency rounds"]
tot = sum(v[:4])
for i, nm in enumerate(names): print("  %-42s %10.0f ticks/step  %5.1f %%" % (nm, v[i] / ns, 100.0 * v[i] / max(tot, 1)))
print("ticks/frame %.0f" % (tot / nf))
