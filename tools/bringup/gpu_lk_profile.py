"""bring-up: per-phase wall time of the link dfast parse (library built with ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE). usage: gpu_lk_profile.py [GiB]"""
import sys, os, ctypes
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
lib = ctypes.CDLL(Z.LIB_PATH)
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1); fs = 65536; n = int(gib * (1 << 30))
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
buf = (ctypes.c_ulonglong * 32)()
eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
lib.ZraHipDebugReadLkProfile(buf, 1)
eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
print("stats", eng.kernel_stats())
lib.ZraHipDebugReadLkProfile(buf, 1)
v = list(buf); nf = max(v[25], 1)
names = ["0 window loads (entries, source, rep gather)", "1 walk (LDS) + ballots", "2 window exit + flush", "3 deep walk (memory)", "4 which match / probe",
         "5 count loads + next rep gather", "6 ml / back / emit", "7 insertions + immediate repcode", "8 re-walk check", "9 tail"]
print("frames %d  parse ticks/frame %.0f (100 MHz ticks: x24 = shader cycles at 2.4 GHz)" % (nf, v[24] / nf))
tot = sum(v[:10])
for i, nm in enumerate(names): print("  %-50s %10.0f  %5.1f %%" % (nm, v[i] / nf, 100.0 * v[i] / max(tot, 1)))
print("  windows/frame %.0f  deep walks %.0f  slow probes %.0f  seqs %.0f  immediate-rep loads %.0f  re-walks %.0f" % (v[12] / nf, v[13] / nf, v[14] / nf, v[15] / nf, v[16] / nf, v[17] / nf))
