"""bring-up: the damaged archive of corruption-soak seed 13141 case 9 (frameSize field overwritten: 4096 -> 1792) through every decode entry point"""
import sys, os
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
import numpy as np, torch, zra_amd as Z, oracle_lib as O
a = open(os.path.join(here, "golden", "corrupt_seed13141_case9.zra"), "rb").read()
U = int.from_bytes(a[18:26], "little")
want, wb = O.zra_decompress(a, U, "zo", defined_only=True)
print("oracle", want, len(wb))
try:
    got = Z.DecompressBuffer(a)
    n = min(len(got), len(wb)); d = next((i for i in range(n) if got[i] != wb[i]), n)
    print("host API: len", len(got), "first diff", d, "equal", got == wb)
    if d < n:
        # which frame-size pattern: print a map of equal/unequal 256-byte cells
        print("".join("." if got[i:i + 256] == wb[i:i + 256] else "X" for i in range(0, n, 256)))
except Z.ZraError as e:
    print("host API raised", e.zra, e.zstd)
eng = Z.Engine(0)
d_arc = torch.from_numpy(np.frombuffer(a, dtype=np.uint8).copy()).cuda()
d_out = torch.zeros(U + 64, dtype=torch.uint8, device="cuda")
try:
    eng.decompress(d_arc.data_ptr(), len(a), d_out.data_ptr(), U)
    g2 = d_out[:U].cpu().numpy().tobytes()
    print("device API: equal", g2 == wb, "stats", eng.kernel_stats())
    if g2 != wb:
        print("".join("." if g2[i:i + 256] == wb[i:i + 256] else "X" for i in range(0, U, 256)))
except Z.ZraError as e:
    print("device API raised", e.zra, e.zstd)
