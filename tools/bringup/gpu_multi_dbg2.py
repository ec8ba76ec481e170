"""bring-up: N ranks on ONE GPU over gloo: compress, gather to rank 0, check every frame on the CPU (libzstd) to see WHAT a bad shard holds"""
import os, sys, time, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch, torch.distributed as dist
import zra_amd as Z, bench, oracle_lib as O
from zra_amd import sharding
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
eng = Z.Engine(0)
comm = sharding.Comm.torch_dist(eng)
fs = 65536; N = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 1 << 30
base = bench.synth_corpus(64 << 20, seed=1 + rank)
d_in = torch.from_numpy(base).to(dev).repeat((N + len(base) - 1) // len(base))[:N].contiguous()
for it in range(2):
    shard = comm.compress(d_in.data_ptr(), N, N * world, 3, fs, True)
    asz = shard.archive_size()
    rootbuf = torch.empty(asz + 64, dtype=torch.uint8, device=dev) if rank == 0 else None
    comm.gather_archive(shard, 0, rootbuf.data_ptr() if rank == 0 else 0, asz + 64 if rank == 0 else 0)
    torch.cuda.synchronize()
    if rank == 0:
        arc = rootbuf[:asz].cpu().numpy()
        tsz = int.from_bytes(arc[26:30].tobytes(), "little"); hs = 38 + 5 * tsz
        tab = arc[38:hs].reshape(tsz, 5).astype(np.uint64)
        ent = tab[:, 0] | (tab[:, 1] << np.uint64(8)) | (tab[:, 2] << np.uint64(16)) | (tab[:, 3] << np.uint64(24)) | (tab[:, 4] << np.uint64(32))
        bad = []
        nfr = tsz - 1; per = nfr // world
        bases = {r: bench.synth_corpus(64 << 20, seed=1 + r) for r in range(world)}
        for f in range(nfr):
            fr = arc[hs + int(ent[f]): hs + int(ent[f + 1])].tobytes()
            out, code = O.decompress(fr, fs, "zl")
            r = f // per; lo = (f - r * per) * fs
            want = bases[r][np.arange(lo, lo + fs) % len(bases[r])].tobytes()
            if out is None:
                bad.append((f, "zstd error", code))
            elif out != want:
                k = next(i for i in range(len(want)) if i >= len(out) or out[i] != want[i])
                bad.append((f, "bytes differ at", k, len(out)))
            if len(bad) >= 12: break
        print("iteration", it, "frames", nfr, "bad", len(bad), bad[:12], flush=True)
    shard.close()
    dist.barrier()
