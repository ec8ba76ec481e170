"""bring-up: compress speed on the bench corpus. usage: gpu_speed.py [GiB] [level] [frameSize] [iterations]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch, zra_amd as Z, bench
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fs = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
eng = Z.Engine(0); dev = torch.device("cuda", 0)
N = int(gib * (1 << 30))
c64 = bench.synth_corpus(64 << 20, 1)
if os.environ.get("LOGLIKE"):                         # C4's data: the log-like generator of tests/corpus.py
    import corpus as C
    c64 = np.frombuffer(C.gen_loglike(32 << 20, seed=4), dtype=np.uint8)
t = torch.from_numpy(np.resize(c64, N)).to(dev)
out = torch.empty(Z.GetOutputBufferSize(N, fs) + 64, dtype=torch.uint8, device=dev)
import hashlib
for it in range(iters):
    torch.cuda.synchronize(); t0 = time.time()
    n = eng.compress(t.data_ptr(), N, out.data_ptr(), level, fs, True)
    torch.cuda.synchronize(); dt = time.time() - t0
    h = hashlib.sha256(out[:n].cpu().numpy().tobytes()).hexdigest()[:16] if it == 0 else ""
    print("compress %.2f GiB L%d fs %d: %.1f ms = %.2f GiB/s, ratio %.3f, %s %s" % (gib, level, fs, dt * 1e3, gib / dt, N / n, eng.kernel_stats(), h), flush=True)
