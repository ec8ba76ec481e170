"""bring-up: full decode time of 4 GiB at different frame sizes (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import sys, os, time
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
dev = torch.device("cuda", 0); eng = Z.Engine(0)
fs = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
base = bench.synth_corpus(64 << 20, 1); n = 4 << 30
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
d_out = torch.empty(n, dtype=torch.uint8, device=dev)
for i in range(3):
    torch.cuda.synchronize(); t = time.time(); eng.decompress(d_arc.data_ptr(), asz, d_out.data_ptr(), n); torch.cuda.synchronize(); dt = time.time() - t
print("fs %d: decode %.1f ms (%.1f GiB/s) %s" % (fs, dt * 1e3, 4 / dt, eng.kernel_stats()))
