"""N>1 on the CPU: the host-side logic of the distributed archive (include/zra_hip.h: shard ranges, the query router, the header
stitch) checked directly, and world_size-2 gloo processes that shard an input by frame index, exchange the frame sizes, stitch the seek
table and serve random-access queries THROUGH THE ROUTER (slices to the owner, bytes back). There is no GPU in this container, so
the per-frame codec of the two ranks is the oracle; the real kernels run the same flow in tests/test_gpu_multi.py."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def test_shard_ranges_and_owner_cover_every_frame():
    from zra_amd import sharding
    for F in (0, 1, 2, 7, 8, 9, 1000, 262144, (1 << 32) + 5):
        for W in (1, 2, 3, 8):
            prev = 0
            for r in range(W):
                lo, hi = sharding.shard_range(F, r, W)
                assert lo == prev and hi >= lo
                prev = hi
                for f in {lo, (lo + hi) // 2, hi - 1} if hi > lo else ():
                    assert sharding.owner_of_frame(F, W, f) == r, (F, W, r, f)
            assert prev == F


def test_router_cuts_queries_at_ownership_boundaries():
    """every byte of every query in exactly one slice, the slice's frames all belong to its owner, slices grouped by owner in query
    order; a query that reaches the last byte is refused like DecompressRA refuses it (zra.cpp:260)"""
    import zra_amd as Z
    from zra_amd import sharding
    rng = np.random.RandomState(5)
    for case in range(60):
        fs = int(rng.choice([1024, 4096, 65536]))
        W = int(rng.choice([1, 2, 3, 8]))
        U = int(rng.randint(2, 200 * fs))
        F = (U + fs - 1) // fs
        nq = int(rng.randint(0, 50))
        sizes = np.minimum(rng.choice([0, 1, 100, fs, 3 * fs + 7, 40 * fs], size=nq), U - 1).astype(np.uint64)
        offs = np.array([rng.randint(0, U - int(s)) for s in sizes], dtype=np.uint64)
        sl, per = sharding.route_queries(U, fs, W, offs, sizes)
        assert list(sl["owner"]) == sorted(sl["owner"]) and [int((sl["owner"] == r).sum()) for r in range(W)] == [int(x) for x in per]
        cover = {q: [] for q in range(nq)}
        for s in sl:
            lo, hi = sharding.shard_range(F, int(s["owner"]), W)
            assert s["size"] > 0 and lo * fs <= s["offset"] and s["offset"] + s["size"] <= min(U, hi * fs)
            cover[int(s["query"])].append((int(s["within"]), int(s["offset"]), int(s["size"])))
        for q in range(nq):
            pos = 0
            for within, off, size in sorted(cover[q]):
                assert within == pos and off == int(offs[q]) + pos
                pos += size
            assert pos == int(sizes[q])
        for r in range(W):                                  # query order inside an owner
            qs = list(sl["query"][sl["owner"] == r])
            assert qs == sorted(qs)
    with pytest.raises(Z.ZraError) as e:
        sharding.route_queries(1000, 64, 2, np.array([990], dtype=np.uint64), np.array([10], dtype=np.uint64))
    assert e.value.zra == 5


def _worker(rank, world, port, fs, level, total, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import corpus as C
    import oracle_lib as O
    import zra_amd as Z
    from zra_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = C.gen_E(1 << 20)[:total]
        nframes = (total + fs - 1) // fs
        lo, hi = sharding.shard_range(nframes, rank, world)
        # --- compress the own frames, exchange the sizes, stitch (what ZraHipCommCompress does with the real kernels)
        frames = [O.compress_frame(data[f * fs:(f + 1) * fs], level, True) for f in range(lo, hi)]
        mine = torch.tensor([len(x) for x in frames], dtype=torch.int64)
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([len(frames)], dtype=torch.int64))
        mx = max(int(c) for c in counts)
        pad = torch.zeros(mx, dtype=torch.int64); pad[: len(frames)] = mine
        allsz = [torch.zeros(mx, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allsz, pad)
        flat = np.concatenate([a[: int(c)].numpy() for a, c in zip(allsz, counts)]).astype(np.uint64)
        header = Z.stitch_header(flat, total, fs)
        st, ref = O.zra_compress(data, level, fs, True)
        assert st == (0, 0) and header == ref[: len(header)]                       # every rank holds the whole seek table
        body_base = int(flat[:lo].sum())
        assert b"".join(frames) == ref[len(header) + body_base: len(header) + body_base + int(mine.sum())]
        # --- serve: this rank's queries over the whole range, through the router
        rng = np.random.RandomState(100 + rank)
        nq = 40
        sizes = np.minimum(rng.choice([1, 100, fs, 2 * fs + 3, 5 * fs], size=nq), total - 1).astype(np.uint64)
        offs = np.array([rng.randint(0, total - int(s)) for s in sizes], dtype=np.uint64)
        offs[0] = max(0, hi * fs - 5) if rank == 0 else max(0, lo * fs - 5); sizes[0] = min(10, total - 1 - int(offs[0]))   # straddles the boundary
        sl, per = sharding.route_queries(total, fs, world, offs, sizes)
        asks = [[(int(s["offset"]), int(s["size"])) for s in sl[sl["owner"] == r]] for r in range(world)]
        box = [None] * world
        dist.all_gather_object(box, asks)
        got_asks = [box[s][rank] for s in range(world)]
        local_arc_body = b"".join(frames)
        ent = np.concatenate([[0], np.cumsum(flat)]).astype(np.int64)

        def decode_own(off, size):
            out = b""
            f = off // fs
            while len(out) < size:
                assert lo <= f < hi, "router sent a frame this rank does not own"
                fr = local_arc_body[int(ent[f]) - body_base: int(ent[f + 1]) - body_base]
                raw, err = O.decompress(fr, fs)
                assert err == 0
                a = off + len(out) - f * fs
                out += raw[a: a + size - len(out)]
                f += 1
            return out
        answers = [[decode_own(o, s) for (o, s) in got_asks[s_]] for s_ in range(world)]
        box = [None] * world
        dist.all_gather_object(box, answers)
        back = [box[r][rank] for r in range(world)]                                  # what owner r decoded for me
        res = [bytearray(int(s)) for s in sizes]
        cur = [0] * world
        for s in sl:
            r = int(s["owner"])
            piece = back[r][cur[r]]; cur[r] += 1
            res[int(s["query"])][int(s["within"]): int(s["within"]) + int(s["size"])] = piece
        ok = all(bytes(res[i]) == data[int(offs[i]): int(offs[i]) + int(sizes[i])] for i in range(nq))
        q.put((rank, ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,fs", [(400000, 65536), (70000, 16384), (65536 * 3, 65536)])
def test_two_rank_shard_stitch_route_serve(total, fs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + total) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fs, 3, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, True), (1, True)]


def test_stitched_archive_of_a_tebibyte_is_rejected():
    """zra.cpp:227-228 after the all-gather of the frame sizes: header + bodies >= 2^40 bytes is CompressedSizeTooLarge on every rank
    (ZraHipCommCompress stitches with the same call). Synthetic sizes: no such archive is ever materialised."""
    import zra_amd as Z
    fs = 1 << 30
    sizes = np.array([1 << 38] * 4, dtype=np.uint64)                  # 2^40 bytes of frame bodies
    with pytest.raises(Z.ZraError) as e:
        Z.stitch_header(sizes, 4 * fs, fs)
    assert e.value.zra == 7
    sizes[3] -= np.uint64(4096)                                       # just below (header of 38 + 5 * 5 bytes included)
    h = Z.stitch_header(sizes, 4 * fs, fs)
    assert len(h) == 38 + 5 * 5 and int.from_bytes(h[38 + 20: 38 + 25], "little") == int(sizes.sum())


def _transport_worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        closed = []

        class FakeComm:
            def __init__(self, kind): self.kind = kind
            def close(self): closed.append(self.kind)

        class FakeSharding:
            class Comm:
                @staticmethod
                def rccl(eng, r, w):
                    if r == 0:
                        raise RuntimeError("ncclCommInitRank failed on this rank only")
                    return FakeComm("rccl")

                @staticmethod
                def torch_dist(eng):
                    return FakeComm("host")
        comm, transport = bench.choose_comm(None, rank, world, True, None, dist, FakeSharding, torch, "rccl")
        q.put((rank, comm.kind, transport.split("/")[0], tuple(closed)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_one_rank_without_rccl_sends_every_rank_to_the_host_transport():
    """bench.py's transport choice: the RCCL communicator fails on rank 0 only; rank 1, which got one, must close it and take the
    torch.distributed transport as well — a world with two transports would hang in the first collective."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 777) % 2000
    procs = [ctx.Process(target=_transport_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "host", "torch.distributed", ()), (1, "host", "torch.distributed", ("rccl",))]


@pytest.mark.parametrize("world,nframes", [(8, 8 * 37), (8, 8 * 37 + 5), (8, 3), (8, 1), (5, 12), (2, 7)])
def test_eight_rank_stitch_through_the_c_entry_points(world, nframes):
    """The size exchange of the sharded CompressBuffer (zra_comm.hip: comm_stitch, the all-gather of 8 bytes per frame that replaces the
    running offset of zra.cpp:216-230) through the C entry points themselves, W ranks as threads of this process talking through the host
    transport callbacks — no engine, no GPU: ragged F % W, ranks without a frame, and one rank that reports a failure. Every rank must
    end up with the same header + seek table, and it must be the one a single writer builds from the same sizes (Z.stitch_header, whose
    bytes the oracle's container is compared with in test_two_rank_shard_stitch_route_serve)."""
    import ctypes
    import threading
    import zra_amd as Z
    L = Z.load()
    L.ZraHipCommStitchSizes.restype = Z.ZraStatus if hasattr(Z, "ZraStatus") else L.ZraHipCommCreateHost.restype
    fs = 4096
    rng = np.random.RandomState(nframes * 31 + world)
    sizes = rng.randint(20, fs + 200, size=nframes).astype(np.uint64)
    total = (nframes - 1) * fs + 1234
    bar = threading.Barrier(world)
    board = [None] * world
    results = [None] * world

    def run(rank, fail_rank):
        def allgather(user, send, recv, nbytes):
            board[rank] = ctypes.string_at(send, nbytes)
            bar.wait()
            ctypes.memmove(recv, b"".join(board), nbytes * world)
            bar.wait()
            return 0

        def exchange(*a):
            return 1
        tr = Z.ZraHipHostTransport(None, Z.ALLGATHER_FN(allgather), Z.EXCHANGE_FN(exchange))
        h = ctypes.c_void_p()
        st = L.ZraHipCommCreateHost(ctypes.byref(h), None, ctypes.byref(tr), rank, world)
        assert (st.zra, st.zstd) == (0, 0)
        lo, hi = sharding.shard_range(nframes, rank, world)
        mine = np.ascontiguousarray(sizes[lo:hi])
        n_local = hi - lo + (1 if rank == fail_rank else 0)            # a wrong share: InputFrameSizeMismatch on that rank -> on every rank
        sh = ctypes.c_void_p()
        st = L.ZraHipCommStitchSizes(h, mine.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n_local), ctypes.c_uint64(total), ctypes.c_uint32(fs), ctypes.byref(sh))
        out = (st.zra, st.zstd, None, None)
        if st.zra == 0:
            L.ZraHipShardHeaderSize.restype = ctypes.c_size_t
            n = L.ZraHipShardHeaderSize(sh)
            buf = ctypes.create_string_buffer(n)
            L.ZraHipShardGetHeader(sh, buf)
            L.ZraHipShardArchiveSize.restype = ctypes.c_uint64
            out = (0, 0, buf.raw, L.ZraHipShardArchiveSize(sh))
            L.ZraHipShardDestroy(sh)
        L.ZraHipCommDestroy(h)
        results[rank] = out

    from zra_amd import sharding
    for fail_rank in (-1, world - 1):
        bar.reset()
        th = [threading.Thread(target=run, args=(r, fail_rank)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(60)
        assert all(r is not None for r in results)
        if fail_rank < 0:
            want = Z.stitch_header(sizes, total, fs)
            assert all(r[0] == 0 and r[2] == want and r[3] == len(want) + int(sizes.sum()) for r in results), [r[:2] for r in results]
            # the seek table is the running offset: entry i = sum of the sizes before frame i, the last one = the body size
            ent = [int.from_bytes(want[38 + 5 * i: 43 + 5 * i], "little") for i in range(nframes + 1)]
            assert ent == [int(x) for x in np.concatenate([[0], np.cumsum(sizes)])]
        else:
            assert all(r[:2] == results[0][:2] and r[0] != 0 for r in results), [r[:2] for r in results]
        results = [None] * world


def test_stitch_only_shard_has_no_body_and_bad_arguments_are_rejected():
    """ZraHipCommStitchSizes on a one-rank engine-less communicator (the all-gather of zra_comm.hip: comm_stitch degenerates to a copy):
    the shard it returns holds the header and NO frames — ZraHipShardGetBody must say so (NULL, 0 bytes) instead of an address built from
    a null pointer; a NULL size array with nLocal > 0 and a frame count beyond the format's u32 table (zra.cpp:118) are statuses, not
    exceptions or reads through NULL."""
    import ctypes
    import zra_amd as Z
    L = Z.load()
    L.ZraHipCommStitchSizes.restype = Z.ZraStatus

    def allgather(user, send, recv, nbytes):
        ctypes.memmove(recv, send, nbytes)
        return 0

    def exchange(*a):
        return 1
    tr = Z.ZraHipHostTransport(None, Z.ALLGATHER_FN(allgather), Z.EXCHANGE_FN(exchange))
    h = ctypes.c_void_p()
    st = L.ZraHipCommCreateHost(ctypes.byref(h), None, ctypes.byref(tr), 0, 1)
    assert (st.zra, st.zstd) == (0, 0)
    fs = 4096
    sizes = np.array([100, 200, 300], dtype=np.uint64)
    sh = ctypes.c_void_p()
    st = L.ZraHipCommStitchSizes(h, sizes.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(3), ctypes.c_uint64(2 * fs + 10), ctypes.c_uint32(fs), ctypes.byref(sh))
    assert (st.zra, st.zstd) == (0, 0)
    body, base, nbytes = ctypes.c_void_p(1), ctypes.c_uint64(7), ctypes.c_uint64(7)
    L.ZraHipShardGetBody(sh, ctypes.byref(body), ctypes.byref(base), ctypes.byref(nbytes))
    assert body.value is None and nbytes.value == 0 and base.value == 0
    L.ZraHipShardArchiveSize.restype = ctypes.c_uint64
    assert L.ZraHipShardArchiveSize(sh) == 38 + 5 * 4 + 600
    L.ZraHipShardDestroy(sh)
    # NULL sizes with a share to describe
    sh = ctypes.c_void_p()
    st = L.ZraHipCommStitchSizes(h, None, ctypes.c_size_t(3), ctypes.c_uint64(2 * fs + 10), ctypes.c_uint32(fs), ctypes.byref(sh))
    assert st.zra != 0 and not sh.value
    # more frames than the table's u32 count can hold: a status (the vectors sized from it would be tens of GiB)
    st = L.ZraHipCommStitchSizes(h, sizes.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(3), ctypes.c_uint64((1 << 40) - 1), ctypes.c_uint32(1), ctypes.byref(sh))
    assert st.zra != 0 and not sh.value
    L.ZraHipCommDestroy(h)
