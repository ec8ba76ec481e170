"""bring-up: one seed / case of the RA-on-damaged-frames soak, every query with both statuses (env knobs select the decoder paths)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
import numpy as np
import zra_amd as Z
import oracle_lib as O, corpus as C
seed, want = int(sys.argv[1]), int(sys.argv[2])
backend = "zl" if O.have_libzstd() else "zo"
for case, a in C.mutated_archives(20000 + seed, 50, O.zra_compress):
    if not C.seek_table_consistent(a): continue
    U = int.from_bytes(a[18:26], "little"); fs = int.from_bytes(a[30:34], "little")
    rng = np.random.RandomState(seed * 1000 + case)
    for _ in range(4):
        if U < 2: break
        off = int(rng.randint(0, U))
        size = max(1, min(int(rng.choice([1, 100, fs, 2 * fs + 3, max(1, U - off - 1), max(1, U - off)])), 1 << 24))
        if case != want: continue
        wq, qbytes = O.zra_ra(a, off, size, backend)
        try:
            g = Z.DecompressRA(a, off, size); mine = (0, 0, g == qbytes)
            if wq == (0, 0) and g != qbytes:
                other = O.zra_ra(a, off, size, "zo")[1]
                d = [i for i in range(min(len(g), len(qbytes))) if g[i] != qbytes[i]]
                print("   lens", len(g), len(qbytes), "zo==zl", other == qbytes, "first diff", d[:3], "ndiff", len(d), "ours", g[d[0]:d[0]+8].hex() if d else "", "oracle", qbytes[d[0]:d[0]+8].hex() if d else "")
        except Z.ZraError as e:
            mine = (e.zra, e.zstd)
        print("case", case, "U", U, "fs", fs, "len", len(a), "query", (off, size), "oracle", wq, "ours", mine, flush=True)
