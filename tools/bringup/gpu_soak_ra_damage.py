"""bring-up: DecompressRA on archives with damaged frames against libzstd behind the container code (the suite runs a few seeds)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zra_amd as Z
import test_gpu_parity as T
lo, hi = int(sys.argv[1]), int(sys.argv[2]); t0 = time.time(); bad = 0
for seed in range(lo, hi):
    try:
        T.test_randomised_random_access_on_damaged_frames(Z, seed)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, str(e)[:300].replace("\n", " "), flush=True)
        if bad > 15: break
print("RA-on-damaged-frames soak: seeds %d..%d, %d seeds with a mismatch, %.0f s" % (lo, hi, bad, time.time() - t0))
