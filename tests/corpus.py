"""Seeded synthetic inputs: generators A-E plus structured stress inputs.

Pure integer code + numpy; identical on every box, so archives can be pinned by sha256.
"""
import numpy as np


def xs32_stream(seed, n):
    """n successive xorshift32 states: x^=x<<13; x^=x>>17; x^=x<<5 (mod 2^32)."""
    out = np.empty(n, dtype=np.uint32)
    x = seed & 0xFFFFFFFF
    # vectorising a recurrence is not possible; do it in chunks with python ints on a small unrolled loop
    for i in range(n):
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        out[i] = x
    return out


def gen_A(n):
    return bytes(n)


_xs_cache = {}


def _xs(seed, n):
    key = seed
    if key not in _xs_cache or len(_xs_cache[key]) < n:
        _xs_cache[key] = xs32_stream(seed, max(n, 1 << 20))
    return _xs_cache[key][:n]


def gen_B(n):
    return (_xs(0x5A524130, n) & 0xFF).astype(np.uint8).tobytes()


def gen_C(n):
    out = bytearray()
    i = 0
    st = [b"OK", b"WARN", b"FAIL"]
    while len(out) < n:
        out += b"line %d of the zra golden corpus; value=%d status=%s\n" % (i, (i * 2654435761) & 0xFFFF, st[(i * 7) % 3])
        i += 1
    return bytes(out[:n])


def gen_D(n):
    x = _xs(0x5A524131, n)
    return ((x & (x >> 8) & (x >> 16)) & 0xFF).astype(np.uint8).tobytes()


def gen_E(n=1 << 20):
    e = gen_C(1 << 20)[:700001] + gen_B(1 << 20)[:100003] + gen_A(50000) + gen_D(1 << 20)[:198572]
    return e[:n]


def gen_struct(n, seed=12345):
    """2-symbol runs, byte runs, short periodic patterns and back-copies: hits every repcode path."""
    rng = np.random.RandomState(seed)
    out = bytearray()
    while len(out) < n:
        k = rng.randint(0, 6)
        if k == 0:
            out += bytes(rng.randint(0, 256, size=rng.randint(1, 40), dtype=np.uint8))
        elif k == 1:
            out += bytes([rng.randint(0, 256)]) * rng.randint(1, 300)
        elif k == 2:
            p = bytes(rng.randint(0, 256, size=rng.randint(2, 9), dtype=np.uint8))
            out += p * rng.randint(2, 40)
        elif k == 3 and len(out) > 16:
            d = rng.randint(1, min(len(out), 60000))
            ln = rng.randint(3, 200)
            s = len(out) - d
            for j in range(ln):
                out.append(out[s + j])
        elif k == 4:
            a, b = rng.randint(0, 256, size=2)
            out += bytes(rng.choice([a, b], size=rng.randint(4, 120)).astype(np.uint8))
        else:
            out += b"key=%d;" % rng.randint(0, 50)
    return bytes(out[:n])


def gen_alpha4(n, seed=7):
    rng = np.random.RandomState(seed)
    return bytes(rng.randint(0, 4, size=n, dtype=np.uint8))


def gen_loglike(n, seed=99):
    rng = np.random.RandomState(seed)
    out = bytearray()
    i = 0
    lv = [b"INFO", b"DEBUG", b"WARN", b"ERROR"]
    while len(out) < n:
        out += b"2026-10-01T12:%02d:%02d.%03d %s req=%08x user=%d path=/api/v1/items/%d latency_ms=%d\n" % (
            (i // 60) % 60, i % 60, rng.randint(0, 1000), lv[rng.randint(0, 4)], rng.randint(0, 1 << 32), rng.randint(0, 500), rng.randint(0, 100000), rng.randint(1, 900))
        i += 1
    return bytes(out[:n])


def gen_litrle(n=262144):
    """two-block 'literal-RLE' input (SURVEY Appendix A closing note)."""
    rng = np.random.RandomState(5)
    chunks = [bytes(rng.randint(0, 256, size=32, dtype=np.uint8)) for _ in range(512)]
    b1 = b"y" * (131072 - 512 * 32) + b"".join(chunks)
    b2 = b"".join(b"z" + c for c in chunks)
    return (b1 + b2 + b"z" * n)[:n]


ALL = {
    "A": gen_A, "B": gen_B, "C": gen_C, "D": gen_D, "E": gen_E,
}


def random_lz_input(rng, n):
    """Synthetic LZ-style data: literal runs of varying entropy interleaved with copies of earlier spans (incl. overlapping ones)."""
    out = bytearray()
    alpha = rng.choice([2, 4, 16, 64, 256])
    while len(out) < n:
        r = rng.rand()
        if r < 0.35 or len(out) < 8:
            k = int(rng.choice([1, 3, 7, 20, 100, 700]))
            out += bytes(rng.randint(0, alpha, size=k).astype(np.uint8).tolist())
        elif r < 0.85:
            off = int(min(len(out), rng.choice([1, 2, 3, 4, 8, 17, 64, 300, 5000, 70000])))
            off = max(1, min(off, len(out)))
            k = int(rng.choice([3, 4, 5, 8, 12, 40, 300, 3000]))
            st = len(out) - off
            for i in range(k):
                out.append(out[st + i])
        elif r < 0.93:
            out += bytes([int(rng.randint(0, 256))]) * int(rng.choice([5, 40, 600, 9000]))
        else:
            alpha = rng.choice([2, 4, 16, 64, 256])
    return bytes(out[:n])


def mutated_archives(seed, cases, compress):
    """Yields (case, damaged archive) for the corruption tests: a valid archive of seeded LZ data (written by `compress(data, level,
    frameSize, checksum) -> (status, bytes)`), then 1..5 bit flips / byte overwrites / truncations. Archives whose damaged header
    promises more than 16 MiB are skipped (the reference's C wrapper trusts that field for the caller's buffer, zra.cpp:519)."""
    rng = np.random.RandomState(3000 + seed)
    for case in range(cases):
        fs = int(rng.choice([1024, 4096, 16384, 65536]))
        n = int(rng.randint(1, 5 * fs))
        level = int(rng.choice([1, 3, 3, 5, 9]))
        d = random_lz_input(rng, n)
        st, arc = compress(d, level, fs, bool(case & 1))
        if st != (0, 0):
            continue                                   # cparams outside the restated set for this size (refused, not faked)
        a = bytearray(arc)
        for _ in range(int(rng.choice([1, 1, 1, 2, 5]))):
            mode = rng.rand()
            if mode < 0.7:
                a[int(rng.randint(38, len(a)))] ^= 1 << int(rng.randint(0, 8))
            elif mode < 0.85:
                a[int(rng.randint(0, len(a)))] = int(rng.randint(0, 256))
            else:
                a = a[: int(rng.randint(38, len(a)))]
            if len(a) < 44:
                break
        a = bytes(a)
        if len(a) < 38 or int.from_bytes(a[18:26], "little") > (1 << 24):
            continue
        yield case, a


def ra_defined_prefix(a, off, size, decompress):
    """How many leading bytes of DecompressRA(a, off, size)'s answer the reference's call DEFINES (zra.cpp:258-296), for an archive whose
    header seek_table_consistent() accepts and whose touched frames all decode without an error: all `size` of them — unless a touched
    frame regenerates another size than its slot (a damaged frame can regenerate fewer bytes and still end properly). Behind such a
    frame's own bytes the answer holds whatever the reference's buffers held: the frame buffer of the first / last phase is reused from
    one to the other (zra.cpp:280, :291), the middle phase packs its frames back to back (:288), and libzstd's copies overrun a
    sequence's end by up to 32 bytes — two correct decoders behind the same container code already differ there (zl / zo on one host:
    profiles/r06_soak_bisect3.txt). Everything IN FRONT of that point is full frames in their places and is compared byte for byte.
    `decompress(frame_bytes, cap) -> (bytes or None, error)` = the dependency's one-shot call (tests/oracle_lib.py decompress)."""
    U = int.from_bytes(a[18:26], "little"); ts = int.from_bytes(a[26:30], "little"); fs = int.from_bytes(a[30:34], "little")
    ents = [int.from_bytes(a[38 + 5 * k:43 + 5 * k], "little") for k in range(ts)]
    body = 38 + 5 * ts
    f, pos, start = off // fs, 0, off % fs
    while pos < size and f < ts - 1:
        slot = min(fs, U - f * fs)
        out, err = decompress(a[body + ents[f]: body + ents[f + 1]], fs)
        r = len(out) if out is not None else 0
        if r != slot:
            return pos + max(0, min(r, slot) - start)
        pos += slot - start
        f += 1; start = 0
    return size


def random_lz_input_far(rng, n):
    """second seeded generator of the differential tests: far offsets (up to the 256 KiB window), periodic data, long zero runs (very
    long matches: zstd's "limited update after a very long match" at block starts), text-like alphabets"""
    out = bytearray()
    words = [bytes(rng.randint(97, 123, size=int(rng.randint(2, 9))).astype(np.uint8).tolist()) for _ in range(int(rng.choice([8, 60, 500])))]
    while len(out) < n:
        r = rng.rand()
        if r < 0.3:
            for _ in range(int(rng.randint(1, 40))):
                out += words[int(rng.randint(0, len(words)))] + b" "
        elif r < 0.6 and len(out) > 16:
            off = int(rng.randint(1, min(len(out), 262000) + 1)); k = int(rng.choice([4, 5, 6, 7, 8, 9, 15, 33, 130, 1000, 20000]))
            st = len(out) - off
            for i in range(k):
                out.append(out[st + i])
        elif r < 0.7:
            per = bytes(rng.randint(0, 256, size=int(rng.choice([1, 2, 3, 5, 8, 13, 64, 257]))).astype(np.uint8).tolist())
            out += per * int(rng.randint(1, 3000 // len(per) + 2))
        elif r < 0.8:
            out += bytes(int(rng.choice([10, 1000, 70000, 200000])))
        else:
            out += bytes(rng.randint(0, 256, size=int(rng.choice([1, 10, 300, 5000]))).astype(np.uint8).tolist())
    return bytes(out[:n])


def far_repeat_input(rng, total, rep_len, dist):
    """compressible block R, `dist - rep_len` bytes of noise, R again, then more LZ data with pieces of R: repeats that lie just inside
    or just outside a window of `dist` bytes (frames larger than the level's window)"""
    R = random_lz_input(rng, rep_len)
    mid = bytes(rng.randint(0, 256, size=dist - rep_len, dtype=np.uint8)) if dist > rep_len else b""
    d = R + mid + R
    while len(d) < total:
        d += random_lz_input(rng, 50000) + R[:30000]
    return d[:total]


def mutated_headers(seed, cases, compress):
    """Yields (case, archive, ra_defined, ra_bytes_defined, what): a valid archive with 1-2 HEADER fields overwritten — frameSize, uncompressedSize,
    tableSize, headerSize, version, single seek-table entries — restricted to values for which the reference's behaviour is defined
    (it trusts the seek table: entries beyond the body, an inflated tableSize or a shifted table are out-of-bounds reads in
    DecompressRA, zra.cpp:265-296, and crash it). ra_defined: DecompressRA may be compared too (only frameSize, a smaller
    uncompressedSize, the version and in-range entry values were touched). ra_bytes_defined: its bytes too — with a damaged frameSize
    the reference copies out of a frameSize-sized buffer that the frames fill only partly, and what libzstd leaves behind the
    regenerated bytes (wildcopy overrun inside the capacity) is not specified; the status is."""
    rng = np.random.RandomState(70000 + seed)

    def put(a, off, n, v):
        a[off:off + n] = int(v).to_bytes(n, "little")

    for case in range(cases):
        fs = int(rng.choice([1024, 4096, 16384, 65536]))
        n = int(rng.randint(1, 5 * fs))
        d = random_lz_input(rng, n)
        st, arc = compress(d, int(rng.choice([1, 3, 3, 5])), fs, bool(case & 1))
        if st != (0, 0):
            continue
        a = bytearray(arc)
        ts = int.from_bytes(a[26:30], "little"); hs = int.from_bytes(a[4:8], "little")
        body = len(arc) - (hs + 8)
        what = []; ra_ok = True; ra_bytes = True
        for _ in range(int(rng.choice([1, 1, 2]))):
            m = int(rng.randint(0, 6))
            if m == 0:
                v = int(rng.choice([1, 2, 3, max(1, fs // 2), fs - 1, fs + 1, 2 * fs, 1792, 3 * fs + 5, 65535, 1 << 24, (1 << 31) + 5])); put(a, 30, 4, v); what.append(("frameSize", v)); ra_bytes = False
            elif m == 1:
                v = int(rng.choice([1, max(1, n - 1), n + 1, max(1, n // 2), n + fs, 2 * n])); put(a, 18, 8, v); what.append(("uncompressedSize", v)); ra_ok &= v <= n
            elif m == 2:
                v = int(rng.choice([1, max(1, ts - 1), ts + 1, ts + 7, 2 * ts])); put(a, 26, 4, v); what.append(("tableSize", v)); ra_ok = False
            elif m == 3:
                v = int(rng.choice([hs - 1, hs + 1, hs + 5, max(0, hs - 5)])); put(a, 4, 4, v); what.append(("headerSize", v)); ra_ok = False
            elif m == 4:
                v = int(rng.choice([0, 2, 3, 255])); put(a, 12, 2, v); what.append(("version", v))
            elif ts >= 2:
                k = int(rng.randint(0, ts)); e = int.from_bytes(a[38 + 5 * k:43 + 5 * k], "little")
                v = int(rng.choice([max(0, e - 1), min(body, e + 1), int(rng.randint(0, body + 1)), 0])); put(a, 38 + 5 * k, 5, v); what.append(("entry", k, v))
        # the reference's RA also needs every frame it touches to exist in the table: a frameSize that makes offset / frameSize run past
        # the table is an out-of-bounds vector read there
        fs2 = int.from_bytes(a[30:34], "little"); u2 = int.from_bytes(a[18:26], "little")
        if fs2 == 0 or u2 > (1 << 24):
            continue
        if (u2 + fs2 - 1) // fs2 + 1 > ts or fs2 > (1 << 22):
            ra_ok = False                             # (a huge frameSize: the reference's random access zero-fills a buffer of that size per call)
        ents = [int.from_bytes(a[38 + 5 * k:43 + 5 * k], "little") for k in range(ts)]
        if any(y < x for x, y in zip(ents, ents[1:])) or (ents and ents[-1] > body):
            ra_ok = False                             # a span that runs backwards is a wrapped size_t there
        yield case, bytes(a), ra_ok, ra_bytes, what


def seek_table_consistent(a):
    """True when the header of archive `a` still describes a table the reference's random access can follow without reading out of
    bounds: magic / version intact, tableSize and headerSize matching uncompressedSize and frameSize, entries non-decreasing and inside
    the bytes present (zra.cpp:265-296 trusts all of that)."""
    if len(a) < 43:
        return False
    hs = int.from_bytes(a[4:8], "little"); U = int.from_bytes(a[18:26], "little"); ts = int.from_bytes(a[26:30], "little")
    fs = int.from_bytes(a[30:34], "little"); ms = int.from_bytes(a[34:38], "little")
    if fs == 0 or ms != 0 or int.from_bytes(a[0:4], "little") != 0x184D2A50 or int.from_bytes(a[8:12], "little") != 0x3041525A or int.from_bytes(a[12:14], "little") != 1:
        return False
    if ts != (U + fs - 1) // fs + 1 or hs != 30 + 5 * ts or len(a) < hs + 8:
        return False
    ents = [int.from_bytes(a[38 + 5 * k:43 + 5 * k], "little") for k in range(ts)]
    return not any(y < x for x, y in zip(ents, ents[1:])) and ents[-1] <= len(a) - (hs + 8)
