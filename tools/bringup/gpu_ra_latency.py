"""bring-up: latency of small random-access batches (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import zra_amd as Z
import bench

dev = torch.device("cuda", 0)
N = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 1 << 30
fs, qb = 65536, 4096
base = bench.synth_corpus(64 << 20, seed=1)
d_in = torch.from_numpy(base).to(dev).repeat(N // len(base))[:N].contiguous()
eng = Z.Engine(0)
d_arc = torch.empty(Z.GetOutputBufferSize(N, fs) + 64, dtype=torch.uint8, device=dev)
n1 = eng.compress(d_in.data_ptr(), N, d_arc.data_ptr(), 3, fs, True)
rng = np.random.RandomState(7)
for bs, reps in ((1, 30), (64, 20), (4096, 8)):
    d_o = torch.empty(bs * qb + 64, dtype=torch.uint8, device=dev)
    sizes = np.full(bs, qb, dtype=np.uint64); oo = np.arange(bs, dtype=np.uint64) * qb
    ts = []
    for r in range(reps + 2):
        offs = rng.randint(0, N - qb - 1, size=bs).astype(np.uint64)
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.decompress_ra_batch(d_arc.data_ptr(), n1, d_o.data_ptr(), offs, sizes, oo)
        ts.append(time.perf_counter() - t)
    assert torch.equal(d_o[:qb], d_in[int(offs[0]): int(offs[0]) + qb])
    ts = sorted(ts[2:])
    print("batch %5d: median %.1f us  min %.1f us  kernel ms %.3f" % (bs, ts[len(ts) // 2] * 1e6, ts[0] * 1e6, eng.last_kernel_ms() if hasattr(eng, "last_kernel_ms") else -1))
