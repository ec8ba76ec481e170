"""profiling driver: one CompressBuffer + one DecompressBuffer + one RA batch over 1 GiB of the bench corpus (device-resident)."""
import sys, os, time
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1); fs = 65536; n = int(gib * (1 << 30))
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
t = time.time(); asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True); t1 = time.time() - t
d_out = torch.empty(n, dtype=torch.uint8, device=dev)
t = time.time(); eng.decompress(d_arc.data_ptr(), asz, d_out.data_ptr(), n); t2 = time.time() - t
assert torch.equal(d_out, d_in)
print("compress %.1f ms (%.2f GiB/s)  decompress %.1f ms (%.2f GiB/s) ratio %.3f" % (t1 * 1e3, gib / t1, t2 * 1e3, gib / t2, n / asz))
