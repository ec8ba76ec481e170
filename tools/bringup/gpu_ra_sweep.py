"""bring-up: batch size at which the lane-per-frame chain kernel overtakes the wave-per-frame one (ZRA_DEC_CHAIN_WAVE in the environment)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import zra_amd as Z
import bench
dev = torch.device("cuda", 0)
N, fs, qb = 4 << 30, 65536, 4096
base = bench.synth_corpus(64 << 20, seed=1)
d_in = torch.from_numpy(base).to(dev).repeat(N // len(base))[:N].contiguous()
eng = Z.Engine(0)
d_arc = torch.empty(Z.GetOutputBufferSize(N, fs) + 64, dtype=torch.uint8, device=dev)
n1 = eng.compress(d_in.data_ptr(), N, d_arc.data_ptr(), 3, fs, True)
rng = np.random.RandomState(7)
for bs in (1, 64, 1024, 4096, 8192, 16384, 32768):
    d_o = torch.empty(bs * qb + 64, dtype=torch.uint8, device=dev)
    sizes = np.full(bs, qb, dtype=np.uint64); oo = np.arange(bs, dtype=np.uint64) * qb
    ts = []
    for r in range(8):
        offs = rng.randint(0, N - qb - 1, size=bs).astype(np.uint64)
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.decompress_ra_batch(d_arc.data_ptr(), n1, d_o.data_ptr(), offs, sizes, oo)
        ts.append(time.perf_counter() - t)
    ts = sorted(ts[2:])
    print("batch %6d: median %8.1f us" % (bs, ts[len(ts) // 2] * 1e6))
