"""bring-up: differential soak of the dfast levels (3, 4) over ARBITRARY frame sizes — the fixed frame-size list of gpu_soak.py missed a
residue class (round 5: frameSize % 512 in 1..7 broke the bucket-flag sweep). usage: gpu_soak_dfast.py <first seed> <last seed>
(environment: ZRA_MF_LS=0 puts the calls on the table kernel with its flag sweep, span and pipeline modes; default = the LDS-source kernel)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import zra_amd as Z, oracle_lib as O, corpus as C
import test_gpu_parity as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); nc = 0; bad = 0
for seed in range(lo, hi):
    rng = np.random.RandomState(770000 + seed)
    for case in range(12):
        kind = int(rng.randint(0, 4))
        if kind == 0: fs = int(rng.randint(600, 150000))
        elif kind == 1: fs = 512 * int(rng.randint(2, 200)) + int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 63, 64, 65, 503, 504, 505, 510, 511]))
        elif kind == 2: fs = 64 * int(rng.randint(10, 2000)) + int(rng.randint(0, 64))
        else: fs = int(rng.choice([4096, 16384, 65536, 131072]))
        m = int(rng.randint(1, 5))
        tail = int(rng.choice([0, 1, 7, 8, 9, int(rng.randint(0, fs)), 512 * int(rng.randint(0, max(1, fs // 512))) + int(rng.randint(0, 10))]))
        n = min(m * fs + min(tail, fs - 1), 700000)
        level = int(rng.choice([3, 3, 4]))
        d = (C.random_lz_input_far if rng.randint(0, 3) == 0 else T._random_input)(rng, n)
        st, ref = O.zra_compress(d, level, fs, bool(case & 1))
        if st != (0, 0): continue
        arc = Z.CompressBuffer(d, level, fs, bool(case & 1))
        nc += 1
        if arc != ref:
            bad += 1; print("MISMATCH seed", seed, "case", case, "n", n, "fs", fs, "level", level, flush=True)
            if bad > 10: break
    if bad > 10: break
    if seed % 50 == 0: print("seed", seed, "ok  %.0f s" % (time.time() - t0), flush=True)
print("dfast soak: seeds %d..%d, %d cases, %d mismatches, %.0f s" % (lo, hi, nc, bad, time.time() - t0))
