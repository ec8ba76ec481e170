"""bring-up: whole-archive decode of an archive whose header claims half the real frame size (every frame regenerates twice its slot):
the decoder's sequential tail. Prints the time and checks the bytes."""
import sys, os, time
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, bench
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1); fs = 65536; n = int(gib * (1 << 30))
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
d_arc[30:34] = torch.tensor(list((fs // 2).to_bytes(4, "little")), dtype=torch.uint8, device=dev)      # frameSize field
d_out = torch.zeros(n, dtype=torch.uint8, device=dev)
for i in range(2):
    torch.cuda.synchronize(); t = time.time()
    try:
        eng.decompress(d_arc.data_ptr(), asz, d_out.data_ptr(), n); ok = "ok"
    except Z.ZraError as e:
        ok = "status %s" % ((e.zra, e.zstd),)
    torch.cuda.synchronize(); dt = time.time() - t
    print("decode %.2f GiB with frameSize halved in the header: %.1f ms, %s, bytes equal: %s" % (gib, dt * 1e3, ok, bool(torch.equal(d_out, d_in))), flush=True)
