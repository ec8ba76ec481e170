import sys, os, time
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1); fs = 65536; n = int(gib * (1 << 30))
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
for i in range(3):
    torch.cuda.synchronize(); t = time.time(); asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True); dt = time.time() - t
    print("compress %.0f GiB: %.1f ms (%.2f GiB/s) stats %s" % (gib, dt * 1e3, gib / dt, eng.kernel_stats()), flush=True)
