"""bring-up: the persistent pipeline must be deterministic — repeated compress / decompress / batched RA of the same buffers give the same bytes."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, zra_amd as Z, bench
gib = float(sys.argv[1]); iters = int(sys.argv[2])
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 3); fs = 65536; n = int(gib * (1 << 30))
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
cap = Z.GetOutputBufferSize(n, fs) + 64
ref = torch.empty(cap, dtype=torch.uint8, device=dev); out = torch.empty(cap, dtype=torch.uint8, device=dev)
back = torch.empty(n, dtype=torch.uint8, device=dev)
n0 = eng.compress(d_in.data_ptr(), n, ref.data_ptr(), 3, fs, True)
rng = np.random.RandomState(5); nq = 20000
offs = rng.randint(0, n - 70000, size=nq).astype(np.uint64); sizes = rng.choice([1, 4096, 65536 + 7], size=nq).astype(np.uint64)
oofs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
ra0 = torch.empty(int(sizes.sum()), dtype=torch.uint8, device=dev); ra1 = torch.empty_like(ra0)
eng.decompress_ra_batch(ref.data_ptr(), n0, ra0.data_ptr(), offs, sizes, oofs)
t0 = time.time(); bad = 0
for i in range(iters):
    n1 = eng.compress(d_in.data_ptr(), n, out.data_ptr(), 3, fs, True)
    if n1 != n0 or not torch.equal(out[:n1], ref[:n0]): bad += 1; print("compress differs at iteration", i, flush=True)
    eng.decompress(out.data_ptr(), n1, back.data_ptr(), n)
    if not torch.equal(back, d_in): bad += 1; print("decompress differs at iteration", i, flush=True)
    eng.decompress_ra_batch(out.data_ptr(), n1, ra1.data_ptr(), offs, sizes, oofs)
    if not torch.equal(ra0, ra1): bad += 1; print("RA differs at iteration", i, flush=True)
print("determinism soak %.1f GiB x %d: %d differences, %.0f s" % (gib, iters, bad, time.time() - t0))
