import sys, os, time
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1); n = int(float(os.environ.get('GIB', '1')) * (1 << 30))
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
cfgs = ((1, 65536), (3, 16384), (3, 65536), (3, 262144), (5, 65536), (9, 262144))
if len(sys.argv) > 1: cfgs = tuple(tuple(int(x) for x in a.split(',')) for a in sys.argv[1:])
for lvl, fs in cfgs:
    d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
    for i in range(2):
        torch.cuda.synchronize(); t = time.time(); asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), lvl, fs, True); dt = time.time() - t
    d_out = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(); t = time.time(); eng.decompress(d_arc.data_ptr(), asz, d_out.data_ptr(), n); dd = time.time() - t
    print("L%d fs=%dK: compress %.2f GiB/s ratio %.2f  decompress %.2f GiB/s %s" % (lvl, fs >> 10, n / (1 << 30) / dt, n / asz, n / (1 << 30) / dd, torch.equal(d_out, d_in)), flush=True)
    del d_arc, d_out
