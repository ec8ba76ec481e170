"""bring-up: N ranks on ONE GPU over gloo (ZRA_BENCH_ONE_GPU-style): compress / serve sequences to localise a failure between steps"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root)
import numpy as np, torch, torch.distributed as dist
import zra_amd as Z, bench
from zra_amd import sharding
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
eng = Z.Engine(0)
comm = sharding.Comm.torch_dist(eng)
fs = 65536; N = 1 << 30; q = 100000; qb = 4096
base = bench.synth_corpus(64 << 20, seed=1 + rank)
d_in = torch.from_numpy(base).to(dev).repeat(N // len(base))[:N].contiguous()
rng = np.random.RandomState(42 + rank)
offs = rng.randint(0, N * world - qb - 1, size=q).astype(np.uint64); sizes = np.full(q, qb, dtype=np.uint64); oo = np.arange(q, dtype=np.uint64) * qb
d_ra = torch.empty(q * qb + 64, dtype=torch.uint8, device=dev)
plan = sys.argv[1] if len(sys.argv) > 1 else "csscss"
shard = None
for i, op in enumerate(plan):
    try:
        if op == "c":
            if shard is not None: shard.close()
            shard = comm.compress(d_in.data_ptr(), N, N * world, 3, fs, True)
        elif op == "s":
            comm.serve(shard, offs, sizes, oo, d_ra.data_ptr())
        elif op == "b":
            dist.barrier(); torch.cuda.synchronize()
        elif op == "w":
            time.sleep(1.5 if rank == 0 else 0.0)
        torch.cuda.synchronize()
        if rank == 0: print("step", i, op, "ok", flush=True)
    except Z.ZraError as e:
        if rank == 0: print("step", i, op, "FAILED", e.zra, e.zstd, flush=True)
dist.barrier()
