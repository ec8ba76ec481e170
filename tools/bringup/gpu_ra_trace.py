import sys, os, time
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1); fs = 65536; n = 16 << 30
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
del d_in
q = 1000000; qb = 4096; rng = np.random.RandomState(42)
offs = rng.randint(0, n - qb - 1, size=q).astype(np.uint64); sizes = np.full(q, qb, dtype=np.uint64); oofs = np.arange(q, dtype=np.uint64) * qb
d_ra = torch.empty(q * qb, dtype=torch.uint8, device=dev)
for i in range(2):
    torch.cuda.synchronize(); t = time.time(); eng.decompress_ra_batch(d_arc.data_ptr(), asz, d_ra.data_ptr(), offs, sizes, oofs); torch.cuda.synchronize()
    print("RA batch %.1f ms" % ((time.time() - t) * 1e3), eng.kernel_stats(), flush=True)
