"""bring-up: long differential soak — the randomised GPU tests of test_gpu_parity.py over many more seeds (not collected by pytest)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zra_amd as Z, oracle_lib as O
import test_gpu_parity as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] == "v2":
    # second generator: far offsets (up to the 256 KiB window), periodic data, long runs, text-like alphabets
    def gen2(rng, n):
        out = bytearray()
        words = [bytes(rng.randint(97, 123, size=int(rng.randint(2, 9))).astype(np.uint8).tolist()) for _ in range(int(rng.choice([8, 60, 500])))]
        while len(out) < n:
            r = rng.rand()
            if r < 0.3:
                for _ in range(int(rng.randint(1, 40))): out += words[int(rng.randint(0, len(words)))] + b" "
            elif r < 0.6 and len(out) > 16:
                off = int(rng.randint(1, min(len(out), 262000) + 1)); k = int(rng.choice([4, 5, 6, 7, 8, 9, 15, 33, 130, 1000, 20000]))
                st = len(out) - off
                for i in range(k): out.append(out[st + i])
            elif r < 0.7:
                per = bytes(rng.randint(0, 256, size=int(rng.choice([1, 2, 3, 5, 8, 13, 64, 257]))).astype(np.uint8).tolist())
                out += per * int(rng.randint(1, 3000 // len(per) + 2))
            elif r < 0.8:
                out += bytes(int(rng.choice([10, 1000, 70000, 200000])))
            else:
                out += bytes(rng.randint(0, 256, size=int(rng.choice([1, 10, 300, 5000]))).astype(np.uint8).tolist())
        return bytes(out[:n])
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
