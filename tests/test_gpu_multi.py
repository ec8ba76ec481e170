"""Distributed archive on the GPU (include/zra_hip.h, zra_amd/csrc/zra_comm.hip): the real kernels behind ZraHipCommCompress /
ZraHipCommGatherArchive / ZraHipCommServe. The box has one GPU, so two ranks share it (two processes, two engines) and talk through
the host transport over gloo; RCCL itself is exercised with a communicator of one rank. Multi-GPU runs are the driver's."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, fs, level, total, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    import corpus as C
    import zra_amd as Z
    from zra_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        data = C.gen_loglike(total, seed=9)[:total]
        eng = Z.Engine(0)
        comm = sharding.Comm.torch_dist(eng)
        F = (total + fs - 1) // fs
        lo, hi = sharding.shard_range(F, rank, world)
        b0, b1 = min(total, lo * fs), min(total, hi * fs)
        d_local = torch.from_numpy(np.frombuffer(data[b0:b1], dtype=np.uint8).copy()).to(dev) if b1 > b0 else torch.empty(1, dtype=torch.uint8, device=dev)
        shard = comm.compress(d_local.data_ptr(), b1 - b0, total, level, fs, True)
        # the archive in one piece on rank 0 == what one engine writes
        ref = Z.CompressBuffer(data, level, fs, True)
        assert shard.header() == ref[: len(shard.header())] and shard.archive_size() == len(ref)
        d_arc = torch.zeros(len(ref) + 64, dtype=torch.uint8, device=dev) if rank == 0 else None
        n = comm.gather_archive(shard, 0, d_arc.data_ptr() if rank == 0 else 0, len(ref) + 64 if rank == 0 else 0)
        same = True
        if rank == 0:
            same = n == len(ref) and d_arc[:n].cpu().numpy().tobytes() == ref
            # ... and == what the oracle's container writes (the CPU restatement of zra.cpp:194-235): the multi-rank row stands on the
            # oracle itself, not only on the single-engine path
            import oracle_lib as O
            st_o, o_ref = O.zra_compress(data, level, fs, True)
            same = same and st_o == (0, 0) and d_arc[:n].cpu().numpy().tobytes() == o_ref
        # a root that cannot take the archive: every rank gets OutputBufferTooSmall, nobody hangs
        try:
            comm.gather_archive(shard, 0, d_arc.data_ptr() if rank == 0 else 0, 10 if rank == 0 else 0)
            refused = False
        except Z.ZraError as e:
            refused = e.zra == 6
        # serving: both ranks ask over the whole range at the same time
        rng = np.random.RandomState(300 + rank)
        nq = 500
        sizes = np.minimum(rng.choice([1, 100, 4096, fs, 2 * fs + 3, 9 * fs], size=nq), total - 1).astype(np.uint64)
        offs = np.array([rng.randint(0, total - int(s)) for s in sizes], dtype=np.uint64)
        bnd = sharding.shard_range(F, 0, world)[1] * fs
        if bnd >= 7 and bnd + 14 < total:
            offs[0] = bnd - 7; sizes[0] = 20                                       # straddles the ownership boundary
        offs[1] = 0; sizes[1] = total - 1                                          # everything a query may ask for
        oo = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
        d_out = torch.zeros(int(sizes.sum()) + 64, dtype=torch.uint8, device=dev)
        comm.serve(shard, offs, sizes, oo, d_out.data_ptr())
        host = d_out.cpu().numpy().tobytes()
        served = all(host[int(oo[i]): int(oo[i]) + int(sizes[i])] == data[int(offs[i]): int(offs[i]) + int(sizes[i])] for i in range(nq))
        # one rank asks out of bounds: the call fails on BOTH ranks with OutOfBoundsAccess
        bad = offs.copy()
        if rank == 1:
            bad[3] = total - int(sizes[3])
        try:
            comm.serve(shard, bad, sizes, oo, d_out.data_ptr())
            oob = False
        except Z.ZraError as e:
            oob = e.zra == 5
        # an empty batch on one rank while the other asks
        if rank == 0:
            comm.serve(shard, offs[:0], sizes[:0], oo[:0], d_out.data_ptr())
            empty_ok = True
        else:
            d_out.zero_()
            comm.serve(shard, offs[:50], sizes[:50], oo[:50], d_out.data_ptr())
            host = d_out.cpu().numpy().tobytes()
            empty_ok = all(host[int(oo[i]): int(oo[i]) + int(sizes[i])] == data[int(offs[i]): int(offs[i]) + int(sizes[i])] for i in range(50))
        # round 6: the gather on a second communicator's own stream WHILE both ranks serve on the first (the sharded bench step): the
        # archive on rank 0 and every answer as before
        side = sharding.Comm.torch_dist(eng, group=dist.new_group(backend="gloo")).use_own_stream()
        if rank == 0:
            d_arc.zero_()
        d_out.zero_()
        side.gather_archive_begin(shard, 0, d_arc.data_ptr() if rank == 0 else 0, len(ref) + 64 if rank == 0 else 0)
        comm.serve(shard, offs, sizes, oo, d_out.data_ptr())
        n2 = side.gather_archive_end()
        host = d_out.cpu().numpy().tobytes()
        overlap_ok = all(host[int(oo[i]): int(oo[i]) + int(sizes[i])] == data[int(offs[i]): int(offs[i]) + int(sizes[i])] for i in range(nq))
        if rank == 0:
            overlap_ok = overlap_ok and n2 == len(ref) and d_arc[:n2].cpu().numpy().tobytes() == ref
        same = same and overlap_ok
        q.put((rank, same, refused, served, oob, empty_ok))
        side.close()
        shard.close(); comm.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,fs,level", [(5 * 1024 * 1024 + 777, 65536, 3), (3 * 262144 + 200000, 262144, 9), (65536 * 2, 65536, 3), (40000, 65536, 3)])
def test_two_ranks_on_one_gpu_compress_gather_serve(zra, total, fs, level):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + total) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fs, level, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, True, True, True, True, True), (1, True, True, True, True, True)]


def test_rccl_transport_with_one_rank(zra, gpu_engine):
    """RCCL called by the library itself: unique id, ncclCommInitRank, ncclAllGather on the engine's stream. With one rank the
    distributed calls reduce to the single-GPU ones, byte for byte."""
    import torch
    import corpus as C
    from zra_amd import sharding
    dev = torch.device("cuda", 0)
    total, fs = 3 * 1024 * 1024 + 123, 65536
    data = C.gen_loglike(total, seed=2)[:total]
    comm = sharding.Comm.rccl(gpu_engine, 0, 1)
    d_in = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev)
    shard = comm.compress(d_in.data_ptr(), total, total, 3, fs, True)
    ref = zra.CompressBuffer(data, 3, fs, True)
    d_arc = torch.zeros(len(ref) + 64, dtype=torch.uint8, device=dev)
    n = comm.gather_archive(shard, 0, d_arc.data_ptr(), len(ref) + 64)
    assert n == len(ref) and d_arc[:n].cpu().numpy().tobytes() == ref
    rng = np.random.RandomState(1)
    nq = 300
    sizes = rng.choice([1, 4096, fs + 9], size=nq).astype(np.uint64)
    offs = np.array([rng.randint(0, total - int(s)) for s in sizes], dtype=np.uint64)
    oo = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    d_out = torch.zeros(int(sizes.sum()) + 64, dtype=torch.uint8, device=dev)
    comm.serve(shard, offs, sizes, oo, d_out.data_ptr())
    host = d_out.cpu().numpy().tobytes()
    assert all(host[int(oo[i]): int(oo[i]) + int(sizes[i])] == data[int(offs[i]): int(offs[i]) + int(sizes[i])] for i in range(nq))
    shard.close(); comm.close()


@pytest.mark.gpu
def test_rccl_messages_are_cut_into_pieces(zra):
    """zra_comm.hip sends a peer's message as pieces of at most 1 GiB inside the one group (at the headline configuration a rank's frames
    are 5.6 GiB: ZraHipCommGatherArchive, the final gather of zra.cpp:216-230 spread over ranks). With one GPU the only peer is the rank
    itself: ZraHipCommLoopback moves a buffer through grouped ncclSend / ncclRecv to its own rank, here with pieces of 4,000 bytes
    (ZRA_COMM_CHUNK_BYTES, read at the first exchange: a fresh process) over lengths around the piece boundaries."""
    import subprocess
    code = """
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch, zra_amd as Z
from zra_amd import sharding
eng = Z.Engine(0); comm = sharding.Comm.rccl(eng, 0, 1)
dev = torch.device('cuda', 0)
for n in (1, 3999, 4000, 4001, 8000, 123457, 4000 * 64 + 17):
    src = torch.from_numpy(np.random.RandomState(n).randint(0, 256, size=n).astype(np.uint8)).to(dev)
    dst = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    comm.loopback(src.data_ptr(), dst.data_ptr(), n)
    torch.cuda.synchronize()
    assert torch.equal(dst[:n], src) and int(dst[n:].sum()) == 0, n
comm.close(); print('ok')
""" % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZRA_COMM_CHUNK_BYTES="4000"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
