"""bring-up: phase times of the wave-cooperative hash-chain finder (library built with ZRA_EXTRA_CFLAGS=-DZRA_MF_PROFILE)"""
import sys, os, ctypes, time
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench, corpus as C
lvl, fs = int(sys.argv[1]), int(sys.argv[2]); gib = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
lib = ctypes.CDLL(Z.LIB_PATH)
dev = torch.device("cuda", 0); eng = Z.Engine(0)
n = int(gib * (1 << 30))
base = np.frombuffer(C.gen_loglike(32 << 20, seed=4), dtype=np.uint8) if os.environ.get("LOGLIKE") else bench.synth_corpus(64 << 20, 1)
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
buf = (ctypes.c_ulonglong * 24)()
eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), lvl, fs, True)
lib.ZraHipDebugReadMfProfile(buf, 1)
torch.cuda.synchronize(); t = time.time()
asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), lvl, fs, True)
dt = time.time() - t
lib.ZraHipDebugReadMfProfile(buf, 1)
v = list(buf); nb = max(v[5], 1)
print("L%d fs %d: %.2f GiB/s  ratio %.2f  stats %s" % (lvl, fs, gib / dt, n / asz, eng.kernel_stats()))
tot = v[0] + v[1] + v[2]
print("blocks %d  ticks/block %.0f (100 MHz: %.2f ms)  insert %.1f %%  search %.1f %%  parse %.1f %%" % (nb, tot / nb, tot / nb / 1e5, 100 * v[0] / tot, 100 * v[1] / tot, 100 * v[2] / tot))
print("windows/block %.0f  insert steps/block %.0f  bucket groups/insert step %.1f  seqs/block %.0f  positions/window %.1f" % (v[3] / nb, v[6] / nb, v[7] / max(v[6], 1), v[4] / nb, (n / nb) / max(v[3] / nb, 1)))
print("chain steps/block: all lanes %.0f, lanes the parse asked %.0f (%.1f %%), searches asked/block %.0f, steps per asked search %.1f" % (v[8] / nb, v[9] / nb, 100.0 * v[9] / max(v[8], 1), v[10] / nb, v[9] / max(v[10], 1)))
