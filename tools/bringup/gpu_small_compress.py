"""bring-up: latency of compressing small device-resident buffers (the streaming Compressor's per-call cost)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import zra_amd as Z
import bench

dev = torch.device("cuda", 0)
base = bench.synth_corpus(256 << 20, seed=1)
d_all = torch.from_numpy(base).to(dev)
eng = Z.Engine(0)
for fs in (65536, 16384):
    for mb in (10, 40, 160):
        n = mb * 1000 * 1000 // fs * fs
        d_in = d_all[:n]
        d_body = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
        d_sizes = torch.empty(n // fs + 1, dtype=torch.int64, device=dev)
        ts = []
        for r in range(7):
            torch.cuda.synchronize(); t = time.perf_counter()
            eng.compress_frames(d_in.data_ptr(), n, d_body.data_ptr(), d_sizes.data_ptr(), 3, fs, True)
            ts.append(time.perf_counter() - t)
        ts = sorted(ts[2:])
        st = eng.kernel_stats() if hasattr(eng, "kernel_stats") else None
        print("fs %6d  %4d MB (%5d frames): %.2f ms  -> %.2f GB/s   kernel stats %s" % (fs, mb, n // fs, ts[len(ts) // 2] * 1e3, n / ts[len(ts) // 2] / 1e9, st))
