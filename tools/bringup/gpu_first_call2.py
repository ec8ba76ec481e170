"""bring-up: first-call cost, device-pointer level: engine creation, first / second compress and decode of 256 MiB resident in HBM"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import zra_amd as Z, bench
dev = torch.device("cuda", 0)
base = bench.synth_corpus(64 << 20, 1); n = 256 << 20; fs = 65536
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
d_out = torch.empty(n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
t = time.perf_counter(); eng = Z.Engine(0); print("Engine(0): %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True); torch.cuda.synchronize()
    print("compress 256 MiB call %d: %.1f ms" % (i + 1, (time.perf_counter() - t) * 1e3), flush=True)
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); eng.decompress(d_arc.data_ptr(), asz, d_out.data_ptr(), n); torch.cuda.synchronize()
    print("decompress call %d: %.1f ms" % (i + 1, (time.perf_counter() - t) * 1e3), flush=True)
for lvl in (1, 5, 19):
    m = n if lvl < 19 else 16 << 20
    for i in range(2):
        torch.cuda.synchronize(); t = time.perf_counter(); eng.compress(d_in.data_ptr(), m, d_arc.data_ptr(), lvl, fs, True); torch.cuda.synchronize()
        print("compress level %d (%d MiB) call %d: %.1f ms" % (lvl, m >> 20, i + 1, (time.perf_counter() - t) * 1e3), flush=True)
