"""bring-up: per-phase wall time of the dfast window match finder (library built with ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE)."""
import sys, os, ctypes
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
lib = ctypes.CDLL(Z.LIB_PATH)
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1); fs = 65536; n = int(gib * (1 << 30))
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
buf = (ctypes.c_ulonglong * 24)()
eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
lib.ZraHipDebugReadMfProfile(buf, 1)
eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
print("stats", eng.kernel_stats())
lib.ZraHipDebugReadMfProfile(buf, 1)
v = list(buf); nf = max(v[22], 1)
names = ["0 src load", "1 dup detect (LDS)", "2 table gather", "3 ballots (probable hits)", "4 rep gather+ballot+visited stores", "5 which match / short probe",
         "6 count loads", "7 ml/back/seq store", "8 complementary inserts", "9 rep loop", "10 tail"]
tot = v[21] / nf
print("frames %d  total/frame %.0f memtime ticks; clear %.0f" % (nf, tot, v[20] / nf))
for i, nm in enumerate(names): print("  %-40s %10.0f  %5.1f %%" % (nm, v[i] / nf, 100.0 * v[i] / max(v[21], 1)))
print("  windows/frame %.0f  cnt13 %.0f  cnt14 %.0f  cnt15 %.0f  cnt16 %.0f  cnt17 %.0f  cnt18 %.0f  (seqs, out-of-window loads, probe loads, rep-loop seqs, duplicate rounds)" % (v[12] / nf, v[13] / nf, v[14] / nf, v[15] / nf, v[16] / nf, v[17] / nf, v[18] / nf))

if hasattr(lib, "ZraHipDebugReadEntProfile"):
    eb = (ctypes.c_ulonglong * 16)(); lib.ZraHipDebugReadEntProfile(eb, 0); e = list(eb); ne = max(e[15], 1)
    en = ["0 literal gather + histogram", "1 literal section: old/new table, stream sizes (2c)", "2 literal emit (stream packing)", "3 seq code histograms (1b)",
          "4 Huffman tree on wave 0 || sequence tables on waves 1-3 (2a + 2b)", "5 tile load", "6 FSE chains (64 lanes per stream)", "7 tile pack + flush", "8 tail"]
    tot = sum(e[:9])
    print("entropy kernel: frames %d (two compress calls)  ticks/frame %.0f" % (ne, tot / ne))
    for i, nm in enumerate(en): print("  %-40s %10.0f  %5.1f %%" % (nm, e[i] / ne, 100.0 * e[i] / max(tot, 1)))
