"""bring-up: first differing sequence between the HIP match finder and the oracle's, one configuration."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, corpus as C, oracle_lib as O, zra_amd as Z
level, fs = int(sys.argv[1]), int(sys.argv[2])
eng = Z.Engine(0); dev = torch.device("cuda", 0)
for name, g in (("C", C.gen_C), ("E", C.gen_E), ("D", C.gen_D), ("F", C.gen_struct)):
    d = g(1 << 20)[: 6 * fs]
    t = torch.from_numpy(np.frombuffer(d, dtype=np.uint8).copy()).to(dev)
    out = torch.empty(Z.GetOutputBufferSize(len(d), fs) + 64, dtype=torch.uint8, device=dev)
    n = eng.compress(t.data_ptr(), len(d), out.data_ptr(), level, fs, True)
    st, ref_arc = O.zra_compress(d, level, fs, True)
    print(name, "archive equal:", bytes(out[:n].cpu().numpy()) == ref_arc)
    for f in range(len(d) // fs):
        seqs, (nb, last_ll, skip) = eng.debug_read_seqs(f)
        ref = O.sequences(d[f * fs:(f + 1) * fs], level)
        mine = seqs + [(last_ll, 0, 0)]
        if mine != ref:
            k = next((i for i in range(min(len(mine), len(ref))) if mine[i] != ref[i]), min(len(mine), len(ref)))
            print("  frame", f, "nb", nb, "ref", len(ref) - 1, "first diff at", k, "mine", mine[k - 1:k + 2], "ref", ref[k - 1:k + 2])
            break
