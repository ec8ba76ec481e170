"""bring-up: long differential soak — the randomised GPU tests of test_gpu_parity.py over many more seeds (not collected by pytest)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import zra_amd as Z, oracle_lib as O
import test_gpu_parity as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] == "v2":
    import corpus as C
    gen2 = C.random_lz_input_far
    T._random_input = gen2
t0 = time.time(); nc = nd = 0
for seed in range(lo, hi):
    try:
        T.test_randomised_differential_compress.__wrapped__(Z, seed) if hasattr(T.test_randomised_differential_compress, "__wrapped__") else T.test_randomised_differential_compress(Z, seed)
        nc += 25
        if seed % 3 == 0:
            T.test_randomised_differential_decode(Z, seed); nd += 20
    except BaseException as e:
        print("FAIL seed", seed, type(e).__name__, str(e)[:300], flush=True)
        raise
    if seed % 20 == 0: print("seed", seed, "ok  %.0f s" % (time.time() - t0), flush=True)
print("soak done: %d compress cases, %d decode cases, %.0f s" % (nc, nd, time.time() - t0))
