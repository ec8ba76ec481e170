"""bring-up: one case of test_randomised_random_access_on_damaged_frames. usage: gpu_ra_case.py seed case"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import zra_amd as zra, oracle_lib as O, corpus as C
seed, want = int(sys.argv[1]), int(sys.argv[2])
backend = "zl" if O.have_libzstd() else "zo"
for case, a in C.mutated_archives(20000 + seed, 50, O.zra_compress):
    if case != want or not C.seek_table_consistent(a):
        continue
    U = int.from_bytes(a[18:26], "little"); fs = int.from_bytes(a[30:34], "little")
    print("case", case, "U", U, "fs", fs, "archive bytes", len(a))
    rng = np.random.RandomState(seed * 1000 + case)
    for _ in range(4):
        off = int(rng.randint(0, U))
        size = max(1, min(int(rng.choice([1, 100, fs, 2 * fs + 3, max(1, U - off - 1), max(1, U - off)])), 1 << 24))
        wq, qbytes = O.zra_ra(a, off, size, backend)
        try:
            g = zra.DecompressRA(a, off, size)
            d = next((i for i in range(min(len(g), len(qbytes))) if g[i] != qbytes[i]), None)
            print("  query", (off, size), "want", wq, "got ok; bytes equal", g == qbytes, "first diff at", d, "of", len(g))
        except zra.ZraError as e:
            print("  query", (off, size), "want", wq, "got", (e.zra, e.zstd))
