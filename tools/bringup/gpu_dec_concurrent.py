"""bring-up: do two decode passes on two engines (own streams, own host threads) overlap? aggregate rate of 2 x G GiB against 1 x 2G GiB"""
import sys, os, time, threading
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root)
import numpy as np, torch, zra_amd as Z, bench
G = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
nthr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0); fs = 65536
base = bench.synth_corpus(64 << 20, 1)
def make(gib, eng):
    n = int(gib * (1 << 30)) // fs * fs
    d_in = torch.from_numpy(np.resize(base, n)).to(dev)
    d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
    asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
    d_out = torch.empty(n, dtype=torch.uint8, device=dev)
    return n, d_arc, asz, d_out
engs = [Z.Engine(0) for _ in range(nthr)]
one = make(G * nthr, engs[0])
parts = [make(G, e) for e in engs]
torch.cuda.synchronize()
def run(e, p):
    n, d_arc, asz, d_out = p
    e.decompress(d_arc.data_ptr(), asz, d_out.data_ptr(), n)
for rep in range(3):
    torch.cuda.synchronize(); t = time.time(); run(engs[0], one); torch.cuda.synchronize(); t1 = time.time() - t
    th = [threading.Thread(target=run, args=(engs[k], parts[k])) for k in range(nthr)]
    torch.cuda.synchronize(); t = time.time()
    for x in th: x.start()
    for x in th: x.join()
    torch.cuda.synchronize(); t2 = time.time() - t
    print("one pass of %.0f GiB: %.1f ms (%.1f GiB/s)   %d concurrent passes of %.0f GiB: %.1f ms (%.1f GiB/s)" % (G * nthr, t1 * 1e3, G * nthr / t1, nthr, G, t2 * 1e3, G * nthr / t2), flush=True)
