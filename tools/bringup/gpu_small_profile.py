"""bring-up: per-stage cycles of zra_ra_small_kernel (library built with ZRA_EXTRA_CFLAGS=-DZRA_SMALL_PROFILE) + host phases (ZRA_RA_TRACE=1)"""
import sys, os, time, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root)
import numpy as np, torch, zra_amd as Z, bench
dev = torch.device("cuda", 0); eng = Z.Engine(0)
N = 1 << 30; fs, qb = 65536, 4096
base = bench.synth_corpus(64 << 20, seed=1)
d_in = torch.from_numpy(base).to(dev).repeat(N // len(base))[:N].contiguous()
d_arc = torch.empty(Z.GetOutputBufferSize(N, fs) + 64, dtype=torch.uint8, device=dev)
n1 = eng.compress(d_in.data_ptr(), N, d_arc.data_ptr(), 3, fs, True)
lib = ctypes.CDLL(Z.LIB_PATH)
rng = np.random.RandomState(7)
buf = (ctypes.c_ulonglong * 64)()
for bs in (1, 64):
    d_o = torch.empty(bs * qb + 64, dtype=torch.uint8, device=dev)
    sizes = np.full(bs, qb, dtype=np.uint64); oo = np.arange(bs, dtype=np.uint64) * qb
    if hasattr(lib, "ZraHipDebugReadSmallProfile"): lib.ZraHipDebugReadSmallProfile(buf, 1)
    ts = []
    for r in range(20):
        offs = rng.randint(0, N - qb - 1, size=bs).astype(np.uint64)
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.decompress_ra_batch(d_arc.data_ptr(), n1, d_o.data_ptr(), offs, sizes, oo)
        ts.append(time.perf_counter() - t)
    ts = sorted(ts)
    print("batch %d: median %.1f us  kernel ms %.3f" % (bs, ts[len(ts) // 2] * 1e6, eng.last_kernel_ms()))
    if hasattr(lib, "ZraHipDebugReadSmallProfile"):
        lib.ZraHipDebugReadSmallProfile(buf, 0); v = list(buf); nj = max(v[4], 1)
        print("  parse, cycles per job: to the literals header %.0f  literals header + tree %.0f  sequences header %.0f  table descriptions %.0f  table builds %.0f  hand-over %.0f" % tuple(v[k] / nj for k in (8, 9, 10, 11, 12, 13)))
        print("  wave-wide Huffman: done %d, handed to the serial decoders %d (reached the checks %d, not converged %d)" % (v[14], v[15], v[6], v[7]))
        print("  wave-wide Huffman: lanes restarted after pass 1..6 (per job): %s" % " ".join("%.1f" % (v[16 + k] / nj) for k in range(6)))
        print("  literals header, lane 0, cycles per job: header fields %.0f  weight description %.0f  weight table %.0f  weight decode %.0f" % tuple(v[k] / nj for k in (20, 21, 22, 23)))
        print("  jobs %d; cycles per job: parse %.0f  huffman %.0f  chain producer %.0f  chain consumer %.0f  execute %.0f" % (nj, v[0] / nj, v[1] / nj, v[2] / nj, v[5] / nj, v[3] / nj))
