import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle_lib as O, zra_amd as Z, bench
dev = torch.device("cuda", 0)
eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1)
fs = 65536
for mib in (256, 512, 1024, 1024):
    n = mib << 20
    d_in = torch.from_numpy(np.resize(base, n)).to(dev)
    d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
    asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
    d_out = torch.zeros(n, dtype=torch.uint8, device=dev)
    err = None
    try:
        eng.decompress(d_arc.data_ptr(), asz, d_out.data_ptr(), n)
    except Exception as e:
        err = str(e)
    eq = (d_out.view(-1, fs) == d_in.view(-1, fs)).all(dim=1).cpu().numpy()
    bad = np.nonzero(~eq)[0]
    print(mib, "MiB frames", n // fs, "err", err, "bad frames", len(bad), bad[:20], flush=True)
    if len(bad):
        k = int(bad[0])
        a = d_out[k*fs:(k+1)*fs].cpu().numpy(); b = d_in[k*fs:(k+1)*fs].cpu().numpy()
        diff = np.nonzero(a != b)[0]
        print("  first bad frame", k, "ndiff", len(diff), "first diff idx", diff[:10], "last", diff[-5:])
