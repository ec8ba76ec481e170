"""ctypes binding of the CPU oracle (oracle/libzra_oracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
`zo_*` = the C restatement of the reference hot path; `zl_*` = the same container code driving the
real dependency (libzstd 1.4.9 from the image) — used to pin the restatement and as the CPU baseline.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = os.path.join(ORACLE_DIR, "libzra_oracle.so")


class Status(ctypes.Structure):
    _fields_ = [("zra", ctypes.c_int), ("zstd", ctypes.c_int)]

    def tup(self):
        return (self.zra, self.zstd)


class Seq(ctypes.Structure):
    _fields_ = [("litLength", ctypes.c_uint32), ("matchLength", ctypes.c_uint32), ("offsetValue", ctypes.c_uint32)]


class CParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint) for n in "windowLog chainLog hashLog searchLog minMatch targetLength strategy".split()]

    def tup(self):
        return tuple(getattr(self, f[0]) for f in self._fields_)


def build():
    if not os.path.exists(_LIB) or any(
        os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(_LIB)
        for f in os.listdir(ORACLE_DIR)
        if f.endswith((".c", ".h"))
    ):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libzra_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        sz, vp, u32, i = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int
        L.zo_crc32.restype = u32
        L.zo_crc32.argtypes = [u32, vp, sz]
        L.zo_xxh64.restype = ctypes.c_uint64
        L.zo_xxh64.argtypes = [vp, sz, ctypes.c_uint64]
        L.zo_compress_bound.restype = sz
        L.zo_compress_bound.argtypes = [sz]
        L.zo_get_cparams.restype = i
        L.zo_get_cparams.argtypes = [i, sz, ctypes.POINTER(CParams)]
        for name in ("zo_compress_frame", "zl_compress_frame"):
            f = getattr(L, name)
            f.restype = sz
            f.argtypes = [vp, sz, vp, sz, i, i]
        for name in ("zo_decompress", "zl_decompress"):
            f = getattr(L, name)
            f.restype = sz
            f.argtypes = [vp, sz, vp, sz]
        L.zo_find_frame_size.restype = sz
        L.zo_find_frame_size.argtypes = [vp, sz]
        L.zo_generate_sequences.restype = sz
        L.zo_generate_sequences.argtypes = [ctypes.POINTER(Seq), sz, vp, sz, i]
        L.zo_zra_output_bound.restype = sz
        L.zo_zra_output_bound.argtypes = [sz, u32, u32]
        for p in ("zo", "zl"):
            f = getattr(L, p + "_zra_compress_buffer")
            f.restype = Status
            f.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz), i, u32, i, sz]
            f = getattr(L, p + "_zra_decompress_buffer")
            f.restype = Status
            f.argtypes = [vp, sz, vp, sz]
            f = getattr(L, p + "_zra_decompress_ra")
            f.restype = Status
            f.argtypes = [vp, sz, vp, sz, sz, sz]
        L.zo_libzstd_load.restype = i
        L.zo_libzstd_load.argtypes = [ctypes.c_char_p]
        L.zo_libzstd_version.restype = ctypes.c_char_p
        _lib = L
    return _lib


ERR_LIMIT = (1 << 64) - 120


def is_err(r):
    return r > ERR_LIMIT


def err_code(r):
    return (1 << 64) - r


def _buf(b):
    return (ctypes.c_char * len(b)).from_buffer_copy(b) if len(b) else ctypes.create_string_buffer(1)


def have_libzstd():
    return lib().zo_libzstd_load(None) == 0


def libzstd_symbol(name):
    """address of an export of the loaded dependency (0 when absent)"""
    L = lib()
    L.zo_libzstd_symbol.restype = ctypes.c_void_p
    L.zo_libzstd_symbol.argtypes = [ctypes.c_char_p]
    return L.zo_libzstd_symbol(name.encode()) or 0


def compress_frame(data, level=3, checksum=True, backend="zo"):
    L = lib()
    cap = L.zo_compress_bound(len(data)) + 64
    out = ctypes.create_string_buffer(cap)
    r = getattr(L, backend + "_compress_frame")(out, cap, _buf(data), len(data), level, int(checksum))
    if is_err(r):
        raise RuntimeError("zstd error %d" % err_code(r))
    return out.raw[:r]


def decompress(data, cap, backend="zo"):
    """returns (bytes, 0) or (None, zstd_error_code)"""
    L = lib()
    out = ctypes.create_string_buffer(max(cap, 1))
    r = getattr(L, backend + "_decompress")(out, cap, _buf(data), len(data))
    if is_err(r):
        return None, err_code(r)
    return out.raw[:r], 0


def zra_compress(data, level=3, frame_size=65536, checksum=True, meta_size=0, backend="zo"):
    L = lib()
    cap = L.zo_zra_output_bound(len(data), frame_size, 0)
    out = ctypes.create_string_buffer(cap)
    osz = ctypes.c_size_t(0)
    s = getattr(L, backend + "_zra_compress_buffer")(_buf(data), len(data), out, cap, ctypes.byref(osz), level, frame_size, int(checksum), meta_size)
    return s.tup(), out.raw[: osz.value]


def zra_decompress(arc, cap=None, backend="zo", defined_only=False):
    """(status, bytes). defined_only: just the bytes the decoder regenerated (a damaged header may promise more than the frames hold;
    the reference call returns void, and what lies beyond the regenerated prefix is whatever the buffer held)."""
    L = lib()
    if cap is None:
        cap = int.from_bytes(arc[18:26], "little")
    out = ctypes.create_string_buffer(max(cap, 1))
    s = getattr(L, backend + "_zra_decompress_buffer")(_buf(arc), len(arc), out, cap)
    if defined_only:
        L.zo_zra_last_produced.restype = ctypes.c_size_t
        return s.tup(), out.raw[:min(cap, L.zo_zra_last_produced())]
    return s.tup(), out.raw[:cap]


def zra_ra(arc, offset, size, backend="zo"):
    L = lib()
    out = ctypes.create_string_buffer(max(size, 1))
    s = getattr(L, backend + "_zra_decompress_ra")(_buf(arc), len(arc), out, size, offset, size)
    return s.tup(), out.raw[:size]


def sequences(data, level=3):
    L = lib()
    cap = len(data) // 3 + 16
    arr = (Seq * cap)()
    n = L.zo_generate_sequences(arr, cap, _buf(data), len(data), level)
    if is_err(n):
        raise RuntimeError("zstd error %d" % err_code(n))
    return [(arr[k].litLength, arr[k].matchLength, arr[k].offsetValue) for k in range(n)]


def cparams(level, size):
    cp = CParams()
    if lib().zo_get_cparams(level, size, ctypes.byref(cp)):
        return None
    return cp.tup()
