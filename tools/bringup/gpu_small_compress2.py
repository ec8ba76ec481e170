"""bring-up: latency of small device-resident compress calls (level 3, 64 KiB frames) with and without the LDS-source match finder.
usage: gpu_small_compress2.py  (reads ZRA_MF_LS / ZRA_MF_LS_MAX from the environment)"""
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
fs = 65536
for frames in (1, 16, 152, 256, 512, 768, 1024, 2048, 4096):
    n = frames * fs
    d_in = d_all[:n]
    d_body = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
    d_sizes = torch.empty(n // fs + 1, dtype=torch.int64, device=dev)
    ts = []
    for r in range(7):
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.compress_frames(d_in.data_ptr(), n, d_body.data_ptr(), d_sizes.data_ptr(), 3, fs, True)
        ts.append(time.perf_counter() - t)
    ts = sorted(ts[2:])
    st = eng.kernel_stats()
    print("LS=%s max=%s  %5d frames: %.2f ms  -> %.2f GB/s   mf %.2f ms ent %.2f ms" % (os.environ.get("ZRA_MF_LS", "1"), os.environ.get("ZRA_MF_LS_MAX", "-"), frames, ts[len(ts) // 2] * 1e3, n / ts[len(ts) // 2] / 1e9, st["mf_ms"], st["ent_ms"]), flush=True)
