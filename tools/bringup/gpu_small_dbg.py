"""bring-up: the differential decode cases one by one (prints before each call) to find an input that crashes a kernel"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, oracle_lib as O, corpus as C, zra_amd as Z
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.RandomState(5000 + seed)
for case in range(20):
    fs = int(rng.choice([1024, 4096, 16384, 65536, 131072, 262144, 1 << 20, 50000]))
    n = int(rng.choice([1, 7, 100, fs, fs + 1, 2 * fs + 17, int(rng.randint(1, 4 * fs))]))
    n = min(n, 1500000)
    level = int(rng.choice([-5, -1, 1, 3, 6, 9, 12, 13, 16, 19, 22]))
    d = C.random_lz_input(rng, n)
    st, arc = O.zra_compress(d, level, fs, bool(case & 1), 0, "zl")
    print("case", case, "n", n, "fs", fs, "level", level, "arc", len(arc), flush=True)
    assert Z.DecompressBuffer(arc) == d
    if n > 2:
        for _ in range(3):
            off = int(rng.randint(0, n - 1)); sz = int(rng.randint(1, n - off))
            if off + sz >= n: sz = n - off - 1
            if sz > 0:
                print("  ra", off, sz, flush=True)
                assert Z.DecompressRA(arc, off, sz) == d[off:off + sz]
print("ok")
