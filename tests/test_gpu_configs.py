"""GPU tests at the sizes BASELINE.json names (C3, the one-GPU shard of C4, the one-GPU share of C5; C1 and C2 live in
test_gpu_parity.py). At these sizes the oracle cannot be run over everything, so parity is checked through properties the domain offers
(round trip, idempotence, seek-table invariants + CRC-32, random access == slices of the input) plus byte parity with the oracle on a
prefix / on the reference's anchor archives. Everything goes through the C ABI (ctypes); torch only holds the device buffers."""
import hashlib
import json
import os
import threading
import zlib

import numpy as np
import pytest

import corpus as C
import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GiB = 1 << 30


def _entries(head, n):
    e = np.frombuffer(head[38:38 + 5 * n], dtype=np.uint8).reshape(n, 5).astype(np.int64)
    return e[:, 0] | (e[:, 1] << 8) | (e[:, 2] << 16) | (e[:, 3] << 24) | (e[:, 4] << 32)


def _check_header(head, N, fs, arc_size):
    nent = (N + fs - 1) // fs + 1
    assert int.from_bytes(head[18:26], "little") == N and int.from_bytes(head[26:30], "little") == nent and int.from_bytes(head[30:34], "little") == fs
    ent = _entries(head, nent)
    assert ent[0] == 0 and np.all(np.diff(ent) > 0) and ent[-1] == arc_size - (38 + 5 * nent)
    assert int.from_bytes(head[14:18], "little") == zlib.crc32(head[18:38 + 5 * nent], zlib.crc32(head[:14]))     # CalculateHash, zra.cpp:128-133
    return ent


@pytest.fixture(autouse=True)
def _free_device_memory(gpu_engine):
    """these tests each fill a large part of the 288 GB: start from an empty torch cache and an engine without scratch, and leave it so"""
    import torch
    gpu_engine.release_scratch()
    torch.cuda.empty_cache()
    yield
    gpu_engine.release_scratch()
    torch.cuda.empty_cache()


def _all_queries_equal(torch, d_ra, d_in, offs, qb, chunk=1 << 16):
    """every answer of a fixed-size query batch against the resident input, on the device"""
    dev = d_in.device
    ar = torch.arange(qb, device=dev, dtype=torch.int64)
    for i in range(0, len(offs), chunk):
        o = torch.from_numpy(offs[i:i + chunk].astype(np.int64)).to(dev)
        want = d_in[(o[:, None] + ar[None, :]).reshape(-1)]
        got = d_ra[i * qb:(i + len(o)) * qb]
        if not torch.equal(want, got):
            return False
    return True


def test_config_c3_16gib_1m_random_access_queries(zra, gpu_engine):
    """BASELINE config C3 (the metric's configuration): 16 GiB corpus, frameSize 64 KiB, level 3, 1,000,000 random DecompressRA(offset, 4 KiB)
    queries on one MI355X. ALL one million answers are compared with the input on the device."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    fs, N, Q, qb = 65536, 16 * GiB, 1_000_000, 4096
    base = bench.synth_corpus(64 << 20, seed=1)
    d_in = torch.from_numpy(base).to(dev).repeat(N // len(base))[:N].contiguous()
    bound = zra.GetOutputBufferSize(N, fs)
    d_arc = torch.empty(bound + 64, dtype=torch.uint8, device=dev)
    n1 = gpu_engine.compress(d_in.data_ptr(), N, d_arc.data_ptr(), 3, fs, True)
    nent = N // fs + 1
    head = d_arc[: 38 + 5 * nent].cpu().numpy().tobytes()
    ent = _check_header(head, N, fs, n1)
    # the corpus repeats every 64 MiB = 1024 frames: so do the frame sizes (frames are independent, zra.cpp:216-225)
    sz = np.diff(ent)
    assert np.array_equal(sz[:1024], sz[1024:2048]) and np.array_equal(sz[:1024], sz[-1024:])
    # byte parity with the oracle on all 1,024 distinct frames of the corpus (the rest of the archive repeats them: the size check above)
    nchk = 1024
    st, ref = O.zra_compress(base[: nchk * fs].tobytes(), 3, fs, True)
    refbody = ref[38 + 5 * (nchk + 1):]
    assert d_arc[len(head): len(head) + len(refbody)].cpu().numpy().tobytes() == refbody
    # idempotence
    d_arc2 = torch.empty(bound + 64, dtype=torch.uint8, device=dev)
    n2 = gpu_engine.compress(d_in.data_ptr(), N, d_arc2.data_ptr(), 3, fs, True)
    assert n1 == n2 and torch.equal(d_arc[:n1], d_arc2[:n2])
    del d_arc2
    # round trip of the whole archive
    d_out = torch.empty(N, dtype=torch.uint8, device=dev)
    gpu_engine.decompress(d_arc.data_ptr(), n1, d_out.data_ptr(), N)
    assert torch.equal(d_out, d_in)
    del d_out
    # 1 M queries of 4 KiB, offsets from a fixed seed, every answer checked
    rng = np.random.RandomState(42)
    offs = rng.randint(0, N - qb - 1, size=Q).astype(np.uint64)
    sizes = np.full(Q, qb, dtype=np.uint64)
    oo = np.arange(Q, dtype=np.uint64) * qb
    d_ra = torch.zeros(Q * qb + 64, dtype=torch.uint8, device=dev)
    gpu_engine.decompress_ra_batch(d_arc.data_ptr(), n1, d_ra.data_ptr(), offs, sizes, oo)
    assert _all_queries_equal(torch, d_ra, d_in, offs, qb)
    # small batches (per-slice jobs, nothing proportional to the archive's 262,144 frames): 1, 64 and 4096 queries
    for k in (1, 64, 4096):
        d_ra[: k * qb].zero_()
        gpu_engine.decompress_ra_batch(d_arc.data_ptr(), n1, d_ra.data_ptr(), offs[1000:1000 + k], sizes[:k], oo[:k])
        assert _all_queries_equal(torch, d_ra, d_in, offs[1000:1000 + k], qb)
    # whole-frame mode (reference error behaviour: every touched frame decoded in full, checksums verified) returns the same bytes
    zra.load().ZraHipSetOptions(8)
    try:
        d_ra.zero_()
        gpu_engine.decompress_ra_batch(d_arc.data_ptr(), n1, d_ra.data_ptr(), offs[:200000], sizes[:200000], oo[:200000])
        assert _all_queries_equal(torch, d_ra, d_in, offs[:200000], qb)
    finally:
        zra.load().ZraHipSetOptions(0)


def test_config_c4_shard_8gib_level9_256k(zra, gpu_engine):
    """BASELINE config C4, the share of one GPU: 64 GiB of log-like data over 8 GPUs = 8 GiB per GPU, frameSize 256 KiB, level 9 (lazy2,
    two blocks per frame). The shard goes through ZraHipCompressFrames (what a rank runs) and through the whole-archive call; the
    reference's three level-9 / 256 KiB anchor archives are reproduced by the HIP path itself."""
    import torch
    dev = torch.device("cuda", 0)
    fs, N = 262144, 8 * GiB
    # (1) the reference's anchors at L9 / 256 KiB: archive size + sha256 recorded from the reference
    anc = json.load(open(os.path.join(GOLD, "anchors.json")))
    ci = anc["configs"].index([9, 262144])
    gens = {"A": C.gen_A, "B": C.gen_B, "C": C.gen_C, "D": C.gen_D, "E": C.gen_E}
    for name in ("C", "D", "E"):
        arc = zra.CompressBuffer(gens[name](anc["n"]), 9, fs, True)
        size, sha = anc["archives"][name][ci]
        assert len(arc) == size and hashlib.sha256(arc).hexdigest()[:16] == sha, name
    # (2) the 8 GiB shard
    base = np.frombuffer(C.gen_loglike(32 << 20, seed=4), dtype=np.uint8)
    d_in = torch.from_numpy(base.copy()).to(dev).repeat(N // len(base))[:N].contiguous()
    bound = zra.GetOutputBufferSize(N, fs)
    d_arc = torch.empty(bound + 64, dtype=torch.uint8, device=dev)
    n1 = gpu_engine.compress(d_in.data_ptr(), N, d_arc.data_ptr(), 9, fs, True)
    nent = N // fs + 1
    head = d_arc[: 38 + 5 * nent].cpu().numpy().tobytes()
    ent = _check_header(head, N, fs, n1)
    nchk = 12                                        # 3 MiB through the oracle's lazy2 (slow on the CPU)
    st, ref = O.zra_compress(base[: nchk * fs].tobytes(), 9, fs, True)
    assert st == (0, 0)
    refbody = ref[38 + 5 * (nchk + 1):]
    assert d_arc[len(head): len(head) + len(refbody)].cpu().numpy().tobytes() == refbody
    # the rank-side call of the sharded path gives the same frames: packed body + sizes
    d_body = torch.empty(bound + 64, dtype=torch.uint8, device=dev)
    d_sizes = torch.empty(N // fs, dtype=torch.int64, device=dev)
    blen = gpu_engine.compress_frames(d_in.data_ptr(), N, d_body.data_ptr(), d_sizes.data_ptr(), 9, fs, True)
    assert blen == n1 - len(head) and torch.equal(d_body[:blen], d_arc[len(head):n1])
    assert np.array_equal(d_sizes.cpu().numpy(), np.diff(ent))
    assert zra.stitch_header(d_sizes.cpu().numpy().astype(np.uint64), N, fs) == head      # the stitched header is the archive's
    del d_body
    # round trip
    d_out = torch.empty(N, dtype=torch.uint8, device=dev)
    gpu_engine.decompress(d_arc.data_ptr(), n1, d_out.data_ptr(), N)
    assert torch.equal(d_out, d_in)


def test_config_c5_share_32gib_resident_concurrent_streams(zra, gpu_engine):
    """BASELINE config C5, the share of one GPU: 256 GiB archive over 8 GPUs = 32 GiB of original data per GPU, archive resident in HBM,
    concurrent random-access streams: two engines on two host threads answer different query batches against the same resident archive
    at the same time; every answer is checked."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    fs, N, qb = 65536, 32 * GiB, 4096
    base = bench.synth_corpus(64 << 20, seed=3)
    d_in = torch.from_numpy(base).to(dev).repeat(N // len(base))[:N].contiguous()
    bound = zra.GetOutputBufferSize(N, fs)
    d_arc = torch.empty(bound + 64, dtype=torch.uint8, device=dev)
    n1 = gpu_engine.compress(d_in.data_ptr(), N, d_arc.data_ptr(), 3, fs, True)
    gpu_engine.release_scratch()                      # the compressor's scratch is not needed while serving
    engines = [gpu_engine, zra.Engine(0)]
    Q = 200000
    res, errs = [None, None], []

    def stream(k):
        try:
            rng = np.random.RandomState(100 + k)
            for it in range(3):
                offs = rng.randint(0, N - qb - 1, size=Q).astype(np.uint64)
                sizes = np.full(Q, qb, dtype=np.uint64)
                oo = np.arange(Q, dtype=np.uint64) * qb
                d_ra = torch.zeros(Q * qb + 64, dtype=torch.uint8, device=dev)
                engines[k].decompress_ra_batch(d_arc.data_ptr(), n1, d_ra.data_ptr(), offs, sizes, oo)
                res[k] = (d_ra, offs)
        except Exception as e:               # noqa: BLE001 (reported below, on the main thread)
            errs.append((k, repr(e)))

    th = [threading.Thread(target=stream, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(2):
        d_ra, offs = res[k]
        assert _all_queries_equal(torch, d_ra, d_in, offs, qb), k
    # a mixed batch on the resident archive: sizes from 1 byte to 1 MiB, crossing frame boundaries
    rng = np.random.RandomState(7)
    q = 20000
    sizes = rng.choice([1, 100, 4096, 65536, 70000, 1 << 20], size=q, p=[0.2, 0.2, 0.3, 0.2, 0.09, 0.01]).astype(np.uint64)
    offs = (rng.randint(0, N - (1 << 20) - 2, size=q)).astype(np.uint64)
    oo = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    d_ra = torch.zeros(int(sizes.sum()) + 64, dtype=torch.uint8, device=dev)
    engines[1].decompress_ra_batch(d_arc.data_ptr(), n1, d_ra.data_ptr(), offs, sizes, oo)
    for i in rng.randint(0, q, size=2000):
        o, s, w = int(offs[i]), int(sizes[i]), int(oo[i])
        assert torch.equal(d_ra[w: w + s], d_in[o: o + s]), i


def test_crafted_headers_cannot_write_out_of_bounds(zra, gpu_engine):
    """Device-pointer decode of archives whose header fields disagree: an inflated tableSize, a shrunk uncompressedSize. The
    reference's one multi-frame zstd call runs out of destination (dstSize_tooSmall); nothing may be written behind the caller's buffer
    (guard region), and the batched random access must refuse such a header instead of indexing past the table."""
    import torch
    dev = torch.device("cuda", 0)
    fs = 4096
    data = C.gen_C(10 * fs)
    st, arc = O.zra_compress(data, 3, fs, True)
    nent = len(data) // fs + 1
    guard = 1 << 20

    def run(a, cap):
        d_a = torch.from_numpy(np.frombuffer(a, dtype=np.uint8).copy()).to(dev)
        d_o = torch.full((cap + guard,), 0xA5, dtype=torch.uint8, device=dev)
        try:
            gpu_engine.decompress(d_a.data_ptr(), len(a), d_o.data_ptr(), cap)
            status = (0, 0)
        except zra.ZraError as e:
            status = (e.zra, e.zstd)
        assert bool((d_o[cap:] == 0xA5).all()), "bytes behind the output capacity were touched"
        return status, d_a

    # uncompressedSize shrunk to 2.5 frames: frames 3.. have no room
    a = bytearray(arc)
    a[18:26] = (2 * fs + fs // 2).to_bytes(8, "little")
    status, d_a = run(bytes(a), 2 * fs + fs // 2)
    assert status == (1, 70)
    with pytest.raises(zra.ZraError) as e:          # header fields disagree: frames != ceil(size / frameSize)
        gpu_engine.decompress_ra_batch(d_a.data_ptr(), len(a), d_a.data_ptr(), np.array([0], dtype=np.uint64), np.array([10], dtype=np.uint64), np.array([0], dtype=np.uint64))
    assert e.value.zra == 3
    # tableSize inflated (entries beyond the real table read frame bytes as offsets): must fail cleanly, whatever the code
    a = bytearray(arc)
    a[26:30] = (nent + 6).to_bytes(4, "little")
    a[4:8] = (int.from_bytes(arc[4:8], "little") + 30).to_bytes(4, "little")       # headerSize grows with the table
    status, _ = run(bytes(a), len(data))
    assert status[0] != 0
    # u64 wrap in offset + size
    d_a = torch.from_numpy(np.frombuffer(arc, dtype=np.uint8).copy()).to(dev)
    d_o = torch.zeros(1 << 16, dtype=torch.uint8, device=dev)
    with pytest.raises(zra.ZraError) as e:
        gpu_engine.decompress_ra_batch(d_a.data_ptr(), len(arc), d_o.data_ptr(), np.array([(1 << 64) - 5], dtype=np.uint64), np.array([10], dtype=np.uint64), np.array([0], dtype=np.uint64))
    assert e.value.zra == 5


def test_input_with_a_short_tail_stays_on_the_persistent_pipeline(zra, gpu_engine):
    """an input that is not a multiple of the frame size must take the same (benchmarked) persistent match-finder pipeline as
    an exact multiple — one match-finder launch — and still match the oracle byte for byte."""
    import torch
    dev = torch.device("cuda", 0)
    fs = 65536
    data = (C.gen_E(1 << 20) * 9)[: 8 * (1 << 20) + 1234]          # 128 full frames + a 1,234-byte tail (its own cparams: dfast, minMatch 4)
    st, ref = O.zra_compress(data, 3, fs, True)
    d_in = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev)
    d_arc = torch.empty(zra.GetOutputBufferSize(len(data), fs) + 64, dtype=torch.uint8, device=dev)
    n = gpu_engine.compress(d_in.data_ptr(), len(data), d_arc.data_ptr(), 3, fs, True)
    assert d_arc[:n].cpu().numpy().tobytes() == ref
    assert gpu_engine.kernel_stats()["mf_launches"] == 1


@pytest.mark.gpu
def test_multi_block_frames_in_two_half_batches(zra, gpu_engine):
    """Round 5: frames of several blocks whose batch would hold the whole call run as two half batches on the two scratch contexts (block
    b + 1's match finder waits for block b's entropy stage; two contexts fill each other's gaps) — from 32 frames per CU on. 8,192 frames
    of 132 KiB (two blocks each: 128 KiB + 4 KiB) at level 3, 1.03 GiB of log-like data: the archive is the oracle's, byte for byte
    (reference call site zra.cpp:216-225), the call took the match finder's 2 x 2 launches, and it decodes back."""
    import torch
    dev = torch.device("cuda", 0)
    fs, nfr = 135168, 8192
    base = np.frombuffer(C.gen_loglike(33 << 20, seed=9), dtype=np.uint8)
    data = np.resize(base, fs * nfr).tobytes()
    st, ref = O.zra_compress(data, 3, fs, True)
    assert st == (0, 0)
    d_in = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev)
    d_arc = torch.empty(zra.GetOutputBufferSize(len(data), fs) + 64, dtype=torch.uint8, device=dev)
    n = gpu_engine.compress(d_in.data_ptr(), len(data), d_arc.data_ptr(), 3, fs, True)
    assert n == len(ref)
    assert hashlib.sha256(d_arc[:n].cpu().numpy().tobytes()).digest() == hashlib.sha256(ref).digest()
    assert gpu_engine.kernel_stats()["mf_launches"] == 4
    d_back = torch.empty(len(data), dtype=torch.uint8, device=dev)
    gpu_engine.decompress(d_arc.data_ptr(), n, d_back.data_ptr(), len(data))
    assert torch.equal(d_back, d_in)
