"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/*.h declares, the host-only entry
points behave like the reference's, and the compute entry points fail LOUDLY (no CPU fallback) when no GPU is usable."""
import ctypes
import os
import re

import pytest

import corpus as C
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol(zra):
    L = zra.load()
    declared = set()
    for h in ("zra.h", "zra_hip.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        declared |= set(re.findall(r"ZRA_EXPORT[^;(]*?\b(Zra\w+)\s*\(", txt))
    assert len([s for s in declared if not s.startswith("ZraHip")]) == 29     # the reference's 29-function C ABI
    assert declared == set(zra.C_ABI_SYMBOLS) | set(zra.HIP_ABI_SYMBOLS)
    for s in sorted(declared):
        assert hasattr(L, s), s


def test_product_does_not_link_or_reference_the_oracle():
    # the product path must not route through oracle/: no include, no dlopen, no symbol
    pat = re.compile(r'#include\s+[<"](zo_|[^>"]*oracle)|dlopen\s*\(|\bzo_\w+\s*\(|import\s+oracle_lib|oracle_lib\.|libzra_oracle')
    for root, _, files in os.walk(os.path.join(ROOT, "zra_amd")):
        for f in files:
            if f.endswith((".hip", ".cpp", ".h", ".py")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert not pat.search(txt), (f, pat.search(txt).group(0))
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--undefined-only", os.path.join(ROOT, "zra_amd", "libzra_amd.so")]).decode()
    assert "ZSTD_" not in out and "zo_" not in out


def test_version_strings_and_bound(zra):
    L = zra.load()
    assert L.ZraGetVersion() == 1
    S = zra.ZraStatus
    assert L.ZraGetErrorString(S(0, 0)) == b"The operation was successful"
    assert L.ZraGetErrorString(S(5, 0)) == b"The specified offset and size are past the data contained within the buffer"
    assert L.ZraGetErrorString(S(1, 20)) == b"An error was returned by ZStandard: Corrupted block detected"
    assert L.ZraGetErrorString(S(1, 72)) == b"An error was returned by ZStandard: Src size is incorrect"
    # ZSTD_compressBound-derived sizes probed from the reference: C2 worst case and the three bounds
    assert L.ZraGetCompressedOutputBufferSize(1 << 30, 65536) == 1078542379
    for n, fs in ((0, 65536), (1, 4), (10, 4), (65536, 65536), (65537, 65536), (1 << 20, 16384), (300000, 262144)):
        assert L.ZraGetCompressedOutputBufferSize(n, fs) == O.lib().zo_zra_output_bound(n, fs, 0)


def test_header_object_on_golden_archive(zra):
    L = zra.load()
    arc = open(os.path.join(GOLD, "g1_abcdefghij_fs4.zra"), "rb").read()
    buf = ctypes.create_string_buffer(arc, len(arc))
    h = ctypes.c_void_p()
    st = L.ZraCreateHeader2(ctypes.byref(h), buf, len(arc))
    assert st.tup() == (0, 0)
    assert L.ZraGetVersionWithHeader(h) == 1 and L.ZraGetHeaderSizeWithHeader(h) == 58
    assert L.ZraGetUncompressedSizeWithHeader(h) == 10 and L.ZraGetFrameSizeWithHeader(h) == 4 and L.ZraGetMetadataSize(h) == 0
    L.ZraDeleteHeader(h)
    # validation order of the reference (zra.cpp:141-171): magic / version>1 -> HeaderInvalid(3); version 0 -> ZraVersionLow(2);
    # a buffer of <= 38 bytes -> OutOfBoundsAccess(5)
    bad = bytearray(arc); bad[8] ^= 0xFF
    assert L.ZraCreateHeader2(ctypes.byref(h), ctypes.create_string_buffer(bytes(bad), len(bad)), len(bad)).tup() == (3, 0)
    bad = bytearray(arc); bad[12] = 2
    assert L.ZraCreateHeader2(ctypes.byref(h), ctypes.create_string_buffer(bytes(bad), len(bad)), len(bad)).tup() == (3, 0)
    bad = bytearray(arc); bad[12] = 0
    assert L.ZraCreateHeader2(ctypes.byref(h), ctypes.create_string_buffer(bytes(bad), len(bad)), len(bad)).tup() == (2, 0)
    assert L.ZraCreateHeader2(ctypes.byref(h), buf, 38).tup() == (5, 0)
    # callback flavour
    def rd(off, size, out):
        ctypes.memmove(out, arc[off:off + size], size)
    cb = zra.READ_FN(rd)
    assert L.ZraCreateHeader(ctypes.byref(h), cb).tup() == (0, 0)
    assert L.ZraGetUncompressedSizeWithHeader(h) == 10
    L.ZraDeleteHeader(h)


def test_streaming_compressor_host_logic_without_gpu(zra):
    # header bookkeeping that needs no codec: sizes, HeaderIncomplete before the last frame, meta placement
    L = zra.load()
    c = ctypes.c_void_p()
    meta = b"hello-meta"
    st = L.ZraCreateCompressor(ctypes.byref(c), 100000, 3, 16384, True, ctypes.create_string_buffer(meta, len(meta)), len(meta))
    assert st.tup() == (0, 0)
    assert L.ZraGetHeaderSizeWithCompressor(c) == 38 + len(meta) + 5 * 8          # 7 frames + sentinel
    assert L.ZraGetOutputBufferSizeWithCompressor(c, 16384 * 3 + 5) == O.lib().zo_compress_bound(16384) * 4
    out = ctypes.create_string_buffer(64)
    assert L.ZraGetHeaderWithCompressor(c, out).tup() == (4, 0)                    # HeaderIncomplete
    L.ZraDeleteCompressor(c)


def test_stitch_header_matches_oracle_container(zra):
    # multi-GPU seek-table stitch (host side): header built from frame sizes == header the single-process path writes
    data = C.gen_E(1 << 20)[:400000]
    st, arc = O.zra_compress(data, 3, 65536, True)
    hs = int.from_bytes(arc[4:8], "little") + 8
    n = int.from_bytes(arc[26:30], "little")
    ent = [int.from_bytes(arc[38 + 5 * i: 43 + 5 * i], "little") for i in range(n)]
    sizes = [ent[i + 1] - ent[i] for i in range(n - 1)]
    assert zra.stitch_header(sizes, len(data), 65536) == arc[:hs]
    assert zra.stitch_header([], 0, 65536) == open(os.path.join(GOLD, "g1_empty_fs65536.zra"), "rb").read()


def test_compute_calls_fail_loudly_without_gpu(zra):
    L = zra.load()
    if L.ZraHipDeviceCount() > 0:
        pytest.skip("a GPU is present; covered by the -m gpu tests")
    with pytest.raises(zra.ZraError) as e:
        zra.CompressBuffer(b"x" * 1000, 3, 256, True)
    assert (e.value.zra, e.value.zstd) == (1, 1)      # ZStdError / GENERIC — never a silent CPU result
    arc = open(os.path.join(GOLD, "g1_abcdefghij_fs4.zra"), "rb").read()
    with pytest.raises(zra.ZraError):
        zra.DecompressBuffer(arc)
    with pytest.raises(zra.ZraError):
        zra.Engine(0)


def test_reference_cli_builds_against_our_headers_and_library(tmp_path):
    """Source compatibility of include/zra.hpp: the reference's own programs/zratool.cpp compiles and links UNMODIFIED against
    this repo's headers + libzra_amd.so (compile/link only — running it needs a GPU). Skipped where /root/reference is absent."""
    import subprocess
    src = "/root/reference/programs/zratool.cpp"
    if not os.path.exists(src):
        pytest.skip("/root/reference not present on this box")
    out = str(tmp_path / "ref_zratool")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", out,
                           "-L" + os.path.join(ROOT, "zra_amd"), "-lzra_amd", "-Wl,-rpath," + os.path.join(ROOT, "zra_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    assert os.path.exists(out)


def test_zstd_error_strings_match_dependency(zra):
    """ZraGetErrorString appends ZSTD_getErrorString(code) (zra.cpp:74-78): every code of zstd 1.4.9's enum, against the library."""
    import ctypes
    import oracle_lib as O
    if not O.have_libzstd():
        pytest.skip("libzstd 1.4.x not present")
    addr = O.libzstd_symbol("ZSTD_getErrorString")
    assert addr
    ref = ctypes.CFUNCTYPE(ctypes.c_char_p, ctypes.c_int)(addr)
    L = zra.load()
    for code in range(0, 121):
        want = b"An error was returned by ZStandard: " + ref(code)
        got = L.ZraGetErrorString(zra.ZraStatus(1, code))
        assert got == want, (code, got, want)


def test_bench_traffic_key_follows_the_code_not_the_comments(tmp_path):
    """profiles/traffic.json entries are tied to a build by a hash of zra_amd/csrc (bench.kernel_source_sha): a reworded comment must
    not make a PMC measurement stale, a changed statement must."""
    import shutil, sys
    sys.path.insert(0, ROOT)
    import bench
    here = bench.HERE
    try:
        base = bench.kernel_source_sha()
        dst = tmp_path / "zra_amd" / "csrc"
        shutil.copytree(os.path.join(ROOT, "zra_amd", "csrc"), dst)
        bench.HERE = str(tmp_path)
        assert bench.kernel_source_sha() == base
        f = dst / "zra_kernels.h"
        f.write_text("// a new remark\n" + f.read_text() + "\n/* and another\n   one */\n")
        assert bench.kernel_source_sha() == base
        f.write_text(f.read_text() + "\n#define ZRA_SOMETHING_ELSE 1\n")
        assert bench.kernel_source_sha() != base
    finally:
        bench.HERE = here
