"""bring-up: full-archive decode and batched random-access timing on the device API (not a pytest file). usage: gpu_dec_bench.py [GiB] [frameSize] [decode-only]"""
import sys, os, time
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
dev = torch.device("cuda", 0); eng = Z.Engine(0)
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
base = bench.synth_corpus(64 << 20, 1); n = int(gib * (1 << 30)) // fs * fs
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
d_out = torch.empty(n, dtype=torch.uint8, device=dev)
for i in range(3):
    torch.cuda.synchronize(); t = time.time()
    try: eng.decompress(d_arc.data_ptr(), asz, d_out.data_ptr(), n)
    except Z.ZraError as e:
        if not os.environ.get("NOCHECK"): raise
    torch.cuda.synchronize(); dt = time.time() - t
    st = eng.kernel_stats()
    sg = eng.decode_stage_stats()
    print("decode %.2f GiB: %.1f ms wall (%.1f GiB/s), kernels %.1f ms in %d passes | parse %.1f huf %.1f chain %.1f exec %.1f" % (gib, dt * 1e3, gib / dt, st["dec_ms"], st["dec_launches"], sg["parse_ms"], sg["huf_ms"], sg["chain_ms"], sg["exec_ms"]), flush=True)
assert os.environ.get("NOCHECK") or torch.equal(d_out, d_in)
if len(sys.argv) > 3: sys.exit(0)
rng = np.random.RandomState(42)
for qb, q in ((4096, 1000000), (65536, 100000), (1 << 20, 4000)):
    q = min(q, int(4 * n / qb))
    offs = rng.randint(0, n - qb - 1, size=q).astype(np.uint64); sizes = np.full(q, qb, dtype=np.uint64); oo = np.arange(q, dtype=np.uint64) * qb
    d_ra = torch.empty(q * qb + 64, dtype=torch.uint8, device=dev)
    for i in range(2):
        torch.cuda.synchronize(); t = time.time(); eng.decompress_ra_batch(d_arc.data_ptr(), asz, d_ra.data_ptr(), offs, sizes, oo); torch.cuda.synchronize(); dt = time.time() - t
    st = eng.kernel_stats()
    k = int(rng.randint(0, q))
    assert torch.equal(d_ra[k * qb:(k + 1) * qb], d_in[int(offs[k]):int(offs[k]) + qb])
    sg = eng.decode_stage_stats()
    print("RA %7d B x %7d: %.1f ms (%.3f us/query, %.1f GiB/s returned), decode kernels %.1f ms | parse %.1f huf %.1f chain %.1f exec %.1f" % (qb, q, dt * 1e3, dt / q * 1e6, q * qb / dt / (1 << 30), st["dec_ms"], sg["parse_ms"], sg["huf_ms"], sg["chain_ms"], sg["exec_ms"]), flush=True)
