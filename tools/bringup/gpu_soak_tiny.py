"""bring-up: differential soak with tiny and odd frame sizes (1..100 bytes), all levels, both checksum settings."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, zra_amd as Z, oracle_lib as O
import test_gpu_parity as T
rng = np.random.RandomState(int(sys.argv[1])); n_cases = int(sys.argv[2]); t0 = time.time(); ok = 0; good = 0
for case in range(n_cases):
    fs = int(rng.choice([1, 2, 3, 5, 6, 7, 8, 9, 15, 16, 17, 63, 64, 100, 255, 256, 257, 1000]))
    n = int(rng.choice([0, 1, fs, fs + 1, 2 * fs - 1, int(rng.randint(1, 40 * fs + 2))]))
    n = min(n, 20000)
    level = int(rng.randint(0, 11)); ck = bool(rng.randint(0, 2))
    d = T._random_input(rng, n)
    st, ref = O.zra_compress(d, level, fs, ck)
    try:
        arc = Z.CompressBuffer(d, level, fs, ck)
        assert st == (0, 0) and arc == ref, ("compress", case, fs, n, level, ck, st)
        assert Z.DecompressBuffer(arc) == d, ("decompress", case, fs, n, level)
        good += 1
        if n > 2:
            off = int(rng.randint(0, n - 1)); sz = int(rng.randint(1, n - off))
            if off + sz < n: assert Z.DecompressRA(arc, off, sz) == d[off:off + sz], ("ra", case, off, sz)
    except Z.ZraError as e:
        assert (e.zra, e.zstd) == st, ("status", case, fs, n, level, ck, st, (e.zra, e.zstd))
    ok += 1
print("tiny-frame soak: %d cases ok (%d compressed + decoded, the rest refused with the oracle's status), %.0f s" % (ok, good, time.time() - t0))
