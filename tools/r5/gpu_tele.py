"""round 5: compress speed of the headline configuration with the launch telemetry of the persistent match finder.
usage: gpu_tele.py [GiB] [iterations] [level] [frameSize]"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch, zra_amd as Z, bench
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
fs = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
eng = Z.Engine(0); dev = torch.device("cuda", 0)
N = int(gib * (1 << 30))
t = torch.from_numpy(np.resize(bench.synth_corpus(64 << 20, 1), N)).to(dev)
out = torch.empty(Z.GetOutputBufferSize(N, fs) + 64, dtype=torch.uint8, device=dev)
for it in range(iters):
    torch.cuda.synchronize(); t0 = time.time()
    n = eng.compress(t.data_ptr(), N, out.data_ptr(), level, fs, True)
    torch.cuda.synchronize(); dt = time.time() - t0
    ks = eng.kernel_stats(); tl = eng.launch_telemetry()
    print(json.dumps({"gib": gib, "wall_ms": round(dt * 1e3, 1), "gibs": round(gib / dt, 2), "mf_ms": round(ks["mf_ms"], 1), "ent_ms": round(ks["ent_ms"], 1), "ratio": round(N / n, 3), "tele": tl}), flush=True)
