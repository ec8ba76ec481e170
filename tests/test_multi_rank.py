"""N>1 path on CPU: world_size-2 gloo processes run the same sharding/stitch/gather code the GPU bench uses
(zra_amd/sharding.py); the per-frame codec is replaced by the oracle here because there is no GPU in this container."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, fs, level, total, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import numpy as np
    import corpus as C
    import oracle_lib as O
    from zra_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = C.gen_E(1 << 20)[:total]
        nframes = (total + fs - 1) // fs
        lo, hi = sharding.shard_range(nframes, rank, world)
        body = bytearray()
        sizes = []
        for f in range(lo, hi):
            fr = O.compress_frame(data[f * fs:(f + 1) * fs], level, True)
            sizes.append(len(fr))
            body += fr
        tb = torch.frombuffer(bytearray(body) if body else bytearray(1), dtype=torch.uint8)
        arc, hdr, bases, totals = sharding.gather_archive(tb, torch.tensor(sizes, dtype=torch.int64), total, fs)
        if rank == 0:
            st, ref = O.zra_compress(data, level, fs, True)
            q.put(bytes(arc.numpy().tobytes()) == ref and st == (0, 0))
        # every rank holds the full seek table: lookup of an arbitrary frame works everywhere
        n = int.from_bytes(hdr[26:30], "little")
        assert n == nframes + 1
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,fs", [(400000, 65536), (70000, 16384), (65536 * 3, 65536)])
def test_two_rank_shard_stitch_gather(total, fs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + total) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fs, 3, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
