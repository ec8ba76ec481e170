"""bring-up: damaged HEADER fields (tests/corpus.py mutated_headers) — DecompressBuffer and DecompressRA of the HIP path against the
oracle's container code over the real libzstd ("zl"); not collected by pytest (the suite runs a few seeds of the same test)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zra_amd as Z
import test_gpu_parity as T
lo, hi = int(sys.argv[1]), int(sys.argv[2]); t0 = time.time(); bad = 0
for seed in range(lo, hi):
    try:
        T.test_randomised_header_damage(Z, seed)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, str(e)[:300].replace("\n", " "), flush=True)
        if bad > 15: break
print("header soak: seeds %d..%d, %d seeds with a mismatch, %.0f s" % (lo, hi, bad, time.time() - t0))
