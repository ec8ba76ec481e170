"""zra_amd — Python binding of libzra_amd.so, the MI355X-native ZRA engine.

This module is plumbing only: it loads the in-tree C-ABI shared library (HIP kernels + C++ host engine)
with ctypes and mirrors the reference's interface names (include/zra.h of zraorg/ZRA: ZraCompressBuffer,
ZraDecompressBuffer, ZraDecompressRA, ... + the additive device-pointer calls of include/zra_hip.h).
There is NO CPU codec here: without the built extension or without a GPU the compute calls raise.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzra_amd.so")
if os.environ.get("ZRA_AMD_LIB") and os.environ.get("ZRA_AMD_BRINGUP") == "1":   # bring-up A/B of two builds on one box (tools/ab_lib.sh): both variables, never in production
    LIB_PATH = os.environ["ZRA_AMD_LIB"]

STATUS_NAMES = ["Success", "ZStdError", "ZraVersionLow", "HeaderInvalid", "HeaderIncomplete", "OutOfBoundsAccess",
                "OutputBufferTooSmall", "CompressedSizeTooLarge", "InputFrameSizeMismatch"]


class ZraStatus(ctypes.Structure):
    _fields_ = [("zra", ctypes.c_int), ("zstd", ctypes.c_int)]

    def tup(self):
        return (self.zra, self.zstd)


class ZraError(RuntimeError):
    def __init__(self, status, what=""):
        self.zra, self.zstd = status
        name = STATUS_NAMES[self.zra] if 0 <= self.zra < len(STATUS_NAMES) else str(self.zra)
        super().__init__("%s (zra=%d, zstd=%d) %s" % (name, self.zra, self.zstd, what))


READ_FN = ctypes.CFUNCTYPE(None, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p)


class ZraHipSlice(ctypes.Structure):
    """include/zra_hip.h: one piece of a query cut at ownership boundaries"""
    _fields_ = [("owner", ctypes.c_uint32), ("query", ctypes.c_uint64), ("offset", ctypes.c_uint64), ("size", ctypes.c_uint64), ("within", ctypes.c_uint64)]


ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                               ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t))


class ZraHipHostTransport(ctypes.Structure):
    _fields_ = [("user", ctypes.c_void_p), ("allgather", ALLGATHER_FN), ("exchange", EXCHANGE_FN)]

_lib = None


def load():
    """Loads libzra_amd.so. Raises (loudly) if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("zra_amd: %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)" % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64 with the same soname as /opt/rocm's: whichever is loaded first serves both.
    # torch only finds its GPUs through its own copy, so when torch is installed it goes first (tests and bench.py use both).
    if "torch" not in sys.modules and not os.environ.get("ZRA_NO_TORCH_PRELOAD"):
        try:
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        except Exception:
            pass
    L = ctypes.CDLL(LIB_PATH)
    sz, vp, u32, u64p = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)
    szp = ctypes.POINTER(ctypes.c_size_t)
    S = ZraStatus
    sig = {
        "ZraGetVersion": (ctypes.c_uint16, []),
        "ZraGetErrorString": (ctypes.c_char_p, [S]),
        "ZraCreateHeader": (S, [ctypes.POINTER(vp), READ_FN]),
        "ZraCreateHeader2": (S, [ctypes.POINTER(vp), vp, sz]),
        "ZraDeleteHeader": (None, [vp]),
        "ZraGetVersionWithHeader": (sz, [vp]),
        "ZraGetHeaderSizeWithHeader": (sz, [vp]),
        "ZraGetUncompressedSizeWithHeader": (sz, [vp]),
        "ZraGetFrameSizeWithHeader": (sz, [vp]),
        "ZraGetMetadataSize": (sz, [vp]),
        "ZraGetMetadata": (None, [vp, vp]),
        "ZraGetCompressedOutputBufferSize": (sz, [sz, sz]),
        "ZraCompressBuffer": (S, [vp, sz, vp, szp, ctypes.c_int8, u32, ctypes.c_bool, vp, sz]),
        "ZraDecompressBuffer": (S, [vp, sz, vp]),
        "ZraDecompressRA": (S, [vp, sz, vp, sz, sz]),
        "ZraCreateCompressor": (S, [ctypes.POINTER(vp), sz, ctypes.c_int8, u32, ctypes.c_bool, vp, sz]),
        "ZraDeleteCompressor": (None, [vp]),
        "ZraGetOutputBufferSizeWithCompressor": (sz, [vp, sz]),
        "ZraCompressWithCompressor": (S, [vp, vp, sz, vp, szp]),
        "ZraGetHeaderSizeWithCompressor": (sz, [vp]),
        "ZraGetHeaderWithCompressor": (S, [vp, vp]),
        "ZraCreateDecompressor": (S, [ctypes.POINTER(vp), READ_FN, sz]),
        "ZraDeleteDecompressor": (None, [vp]),
        "ZraGetHeaderWithDecompressor": (vp, [vp]),
        "ZraDecompressWithDecompressor": (S, [vp, sz, sz, vp]),
        "ZraCreateFullDecompressor": (S, [ctypes.POINTER(vp), READ_FN, sz]),
        "ZraDeleteFullDecompressor": (None, [vp]),
        "ZraGetHeaderWithFullDecompressor": (vp, [vp]),
        "ZraDecompressWithFullDecompressor": (S, [vp, vp, sz, szp]),
        # zra_hip.h
        "ZraHipDeviceCount": (ctypes.c_int, []),
        "ZraHipCreateEngine": (S, [ctypes.POINTER(vp), ctypes.c_int]),
        "ZraHipDestroyEngine": (None, [vp]),
        "ZraHipSynchronize": (S, [vp]),
        "ZraHipGetStream": (vp, [vp]),
        "ZraHipWaitStream": (S, [vp, vp]),
        "ZraHipReleaseScratch": (S, [vp]),
        "ZraHipLastKernelMs": (ctypes.c_double, [vp]),
        "ZraHipGetKernelStats": (None, [vp, ctypes.POINTER(ctypes.c_double)]),
        "ZraHipGetDecodeStageStats": (None, [vp, ctypes.POINTER(ctypes.c_double)]),
        "ZraHipGetLaunchTelemetry": (sz, [vp, u64p, sz]),
        "ZraHipCompressBuffer": (S, [vp, vp, sz, vp, szp, ctypes.c_int8, u32, ctypes.c_bool]),
        "ZraHipDecompressBuffer": (S, [vp, vp, sz, vp, sz]),
        "ZraHipDecompressRABatch": (S, [vp, vp, sz, vp, u64p, u64p, u64p, sz]),
        "ZraHipCompressFrames": (S, [vp, vp, sz, vp, vp, szp, ctypes.c_int8, u32, ctypes.c_bool]),
        "ZraHipStitchHeader": (S, [u64p, sz, ctypes.c_uint64, u32, vp, szp]),
        # distributed archive
        "ZraHipShardRange": (None, [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, u64p, u64p]),
        "ZraHipOwnerOfFrame": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64]),
        "ZraHipRouteQueries": (S, [ctypes.c_uint64, u32, ctypes.c_int, u64p, u64p, sz, ctypes.POINTER(ZraHipSlice), sz, szp, u64p]),
        "ZraHipCommGetUniqueId": (S, [vp]),
        "ZraHipCommCreateRccl": (S, [ctypes.POINTER(vp), vp, vp, ctypes.c_int, ctypes.c_int]),
        "ZraHipCommCreateHost": (S, [ctypes.POINTER(vp), vp, ctypes.POINTER(ZraHipHostTransport), ctypes.c_int, ctypes.c_int]),
        "ZraHipCommLoopback": (S, [vp, vp, vp, sz]),
        "ZraHipCommDestroy": (None, [vp]),
        "ZraHipCommCompress": (S, [vp, vp, sz, ctypes.c_uint64, ctypes.c_int8, u32, ctypes.c_bool, ctypes.POINTER(vp)]),
        "ZraHipShardDestroy": (None, [vp]),
        "ZraHipShardHeaderSize": (sz, [vp]),
        "ZraHipShardGetHeader": (None, [vp, vp]),
        "ZraHipShardArchiveSize": (ctypes.c_uint64, [vp]),
        "ZraHipShardGetBody": (None, [vp, ctypes.POINTER(vp), u64p, u64p]),
        "ZraHipCommGatherArchive": (S, [vp, vp, ctypes.c_int, vp, sz, szp]),
        "ZraHipCommUseOwnStream": (S, [vp]),
        "ZraHipCommGatherArchiveBegin": (S, [vp, vp, ctypes.c_int, vp, sz]),
        "ZraHipCommGatherArchiveEnd": (S, [vp, szp]),
        "ZraHipCommServe": (S, [vp, vp, u64p, u64p, u64p, sz, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError here == a symbol include/*.h declares is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


C_ABI_SYMBOLS = [
    "ZraGetVersion", "ZraGetErrorString", "ZraCreateHeader", "ZraCreateHeader2", "ZraDeleteHeader", "ZraGetVersionWithHeader",
    "ZraGetHeaderSizeWithHeader", "ZraGetUncompressedSizeWithHeader", "ZraGetFrameSizeWithHeader", "ZraGetMetadataSize", "ZraGetMetadata",
    "ZraGetCompressedOutputBufferSize", "ZraCompressBuffer", "ZraDecompressBuffer", "ZraDecompressRA", "ZraCreateCompressor",
    "ZraDeleteCompressor", "ZraGetOutputBufferSizeWithCompressor", "ZraCompressWithCompressor", "ZraGetHeaderSizeWithCompressor",
    "ZraGetHeaderWithCompressor", "ZraCreateDecompressor", "ZraDeleteDecompressor", "ZraGetHeaderWithDecompressor",
    "ZraDecompressWithDecompressor", "ZraCreateFullDecompressor", "ZraDeleteFullDecompressor", "ZraGetHeaderWithFullDecompressor",
    "ZraDecompressWithFullDecompressor",
]
HIP_ABI_SYMBOLS = ["ZraHipDeviceCount", "ZraHipCreateEngine", "ZraHipDestroyEngine", "ZraHipSynchronize", "ZraHipGetStream", "ZraHipWaitStream", "ZraHipReleaseScratch", "ZraHipLastKernelMs", "ZraHipGetKernelStats", "ZraHipGetDecodeStageStats", "ZraHipGetLaunchTelemetry",
                   "ZraHipCompressBuffer", "ZraHipDecompressBuffer", "ZraHipDecompressRABatch", "ZraHipCompressFrames", "ZraHipStitchHeader", "ZraHipDebugReadSeqs", "ZraHipSetOptions", "ZraHipGetOptions",
                   "ZraHipShardRange", "ZraHipOwnerOfFrame", "ZraHipRouteQueries", "ZraHipCommGetUniqueId", "ZraHipCommCreateRccl", "ZraHipCommCreateHost", "ZraHipCommLoopback", "ZraHipCommDestroy",
                   "ZraHipCommCompress", "ZraHipCommStitchSizes", "ZraHipShardDestroy", "ZraHipShardHeaderSize", "ZraHipShardGetHeader", "ZraHipShardArchiveSize", "ZraHipShardGetBody",
                   "ZraHipCommGatherArchive", "ZraHipCommUseOwnStream", "ZraHipCommGatherArchiveBegin", "ZraHipCommGatherArchiveEnd", "ZraHipCommServe"]


def _chk(st, what=""):
    if st.zra != 0:
        raise ZraError(st.tup(), what)


def _cbuf(b):
    return (ctypes.c_char * max(len(b), 1)).from_buffer_copy(bytes(b) if len(b) else b"\0")


# ---------------------------------------------------------------------------------------------------------------
# host-pointer calls — same names and argument meaning as the reference's zra:: free functions (zra.hpp:139-194)
def GetOutputBufferSize(input_size, frame_size):
    return load().ZraGetCompressedOutputBufferSize(input_size, frame_size)


def CompressBuffer(data, compressionLevel=0, frameSize=16384, checksum=True, meta=b""):
    L = load()
    cap = L.ZraGetCompressedOutputBufferSize(len(data), frameSize)
    out = ctypes.create_string_buffer(cap)
    osz = ctypes.c_size_t(0)
    mb = _cbuf(meta)
    _chk(L.ZraCompressBuffer(_cbuf(data), len(data), out, ctypes.byref(osz), compressionLevel, frameSize, checksum, mb if len(meta) else None, len(meta)))
    return out.raw[: osz.value]


def DecompressBuffer(archive):
    L = load()
    n = int.from_bytes(bytes(archive[18:26]), "little") if len(archive) >= 26 else 0
    out = ctypes.create_string_buffer(max(n, 1))
    _chk(L.ZraDecompressBuffer(_cbuf(archive), len(archive), out))
    return out.raw[:n]


def DecompressRA(archive, offset, size):
    L = load()
    out = ctypes.create_string_buffer(max(size, 1))
    _chk(L.ZraDecompressRA(_cbuf(archive), len(archive), out, offset, size))
    return out.raw[:size]


# ---------------------------------------------------------------------------------------------------------------
class Engine:
    """Device-pointer engine (include/zra_hip.h). Pointers are plain integers (e.g. torch tensor .data_ptr())."""

    def __init__(self, device=0):
        self.L = load()
        h = ctypes.c_void_p()
        _chk(self.L.ZraHipCreateEngine(ctypes.byref(h), device), "ZraHipCreateEngine")
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            self.L.ZraHipDestroyEngine(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self):
        return self.L.ZraHipGetStream(self.h)

    def wait_stream(self, stream_handle=None):
        """Orders the engine's streams behind work queued on `stream_handle` (a hipStream_t as int; None = the null stream)."""
        _chk(self.L.ZraHipWaitStream(self.h, ctypes.c_void_p(stream_handle or 0)))

    def release_scratch(self):
        """Hands the engine's (grow-only) scratch back to the device."""
        _chk(self.L.ZraHipReleaseScratch(self.h))

    def _order(self):
        # inputs are usually torch tensors produced asynchronously on torch's current stream: make the engine wait for that stream
        # (event wait, no host sync). Plumbing only; C callers use ZraHipWaitStream / their own synchronisation (zra_hip.h).
        t = sys.modules.get("torch")
        if t is not None and t.cuda.is_available():
            self.wait_stream(t.cuda.current_stream(device=self.device).cuda_stream)      # the ENGINE's device, whatever torch's current one is

    def last_kernel_ms(self):
        return self.L.ZraHipLastKernelMs(self.h)

    def kernel_stats(self):
        """{mf_ms, mf_launches, ent_ms, ent_launches, dec_ms, dec_launches} of the last call (HIP events on the engine stream)."""
        a = (ctypes.c_double * 6)()
        self.L.ZraHipGetKernelStats(self.h, a)
        return dict(mf_ms=a[0], mf_launches=int(a[1]), ent_ms=a[2], ent_launches=int(a[3]), dec_ms=a[4], dec_launches=int(a[5]))

    def launch_telemetry(self):
        """What the waves of the last persistent level-3/4 match-finder launch recorded (zra_hip.h: ZraHipGetLaunchTelemetry), reduced:
        effective shader MHz, waves per compute unit (min / max / histogram), waves, frames and frames per wave-second per XCD, the stagger
        of the wave starts and ends. None when the last call took another path."""
        cap = 32 + 2048 + 8 + 2048
        a = (ctypes.c_uint64 * cap)()
        n = self.L.ZraHipGetLaunchTelemetry(self.h, a, cap)
        if n < 32 or a[2] == 0:
            return None
        M = (1 << 64) - 1
        cuw = [int(v) for v in a[32:32 + 2048] if v]
        simd = [[(v >> (16 * k)) & 0xFFFF for k in range(4)] for v in cuw]        # a CU's word: four 16-bit counts, one per SIMD
        cu = [sum(q) for q in simd]
        shist = {}
        for q in simd:
            k = "/".join(str(x) for x in sorted(q, reverse=True)); shist[k] = shist.get(k, 0) + 1
        ent = None
        if n >= 2080 + 8:
            ecu = [int(v) for v in a[2088:n] if v]
            ent = dict(workgroups=int(a[2080]), cus=len(ecu), workgroups_per_cu_max=max(ecu) if ecu else 0, frames=int(a[2083]),
                       resident_ms_mean=round(a[2081] / max(1, a[2080]) / 1e5, 3), waiting_frac=round(a[2082] / max(1, a[2081]), 4),
                       ms_per_frame=round((a[2081] - a[2082]) / max(1, a[2083]) / 1e5, 4))
        hist = {}
        for v in cu:
            hist[v] = hist.get(v, 0) + 1
        first_start, last_end, last_start, first_end = M - a[4], a[5], a[6], M - a[7]
        return dict(shader_mhz=round(a[0] / max(1, a[1]) * 100.0, 1), waves=int(a[2]), cus=len(cu),
                    waves_per_cu_min=min(cu) if cu else 0, waves_per_cu_max=max(cu) if cu else 0,
                    waves_per_cu_hist={str(k): hist[k] for k in sorted(hist)}, waves_per_simd_hist=shist,
                    waves_per_xcd=[int(v) for v in a[8:16]], frames_per_xcd=[int(v) for v in a[16:24]],
                    frames_per_wave_ms_per_xcd=[round(a[16 + i] / (a[24 + i] / 1e5), 4) if a[24 + i] else 0.0 for i in range(8)],
                    span_ms=round((last_end - first_start) / 1e5, 3), start_stagger_ms=round((last_start - first_start) / 1e5, 3),
                    end_stagger_ms=round((last_end - first_end) / 1e5, 3), longest_wave_ms=round(a[3] / 1e5, 3), entropy=ent)

    def decode_stage_stats(self):
        """{parse_ms, huf_ms, chain_ms, exec_ms, rounds, small_ms, small_launches} of the last decode / random-access call."""
        a = (ctypes.c_double * 8)()
        self.L.ZraHipGetDecodeStageStats(self.h, a)
        return dict(parse_ms=a[0], huf_ms=a[1], chain_ms=a[2], exec_ms=a[3], rounds=int(a[4]), small_ms=a[5], small_launches=int(a[6]))

    def debug_read_seqs(self, frame, cap=65536):
        """bring-up: [(litLength, matchLength, offsetValue)] of `frame` in the last batch + (nbSeq, lastLL, skip)."""
        buf = (ctypes.c_uint64 * cap)(); meta = (ctypes.c_uint32 * 3)()
        self.L.ZraHipDebugReadSeqs.restype = ctypes.c_uint32
        n = self.L.ZraHipDebugReadSeqs(self.h, ctypes.c_uint32(frame), buf, ctypes.c_uint32(cap), meta)
        return [(int(v) & 0xFFFFF, (int(v) >> 20) & 0xFFFFF, int(v) >> 40) for v in buf[:n]], tuple(meta)

    def compress(self, d_in, in_size, d_out, level=3, frame_size=65536, checksum=True):
        osz = ctypes.c_size_t(0)
        self._order()
        _chk(self.L.ZraHipCompressBuffer(self.h, d_in, in_size, d_out, ctypes.byref(osz), level, frame_size, checksum))
        return osz.value

    def decompress(self, d_in, in_size, d_out, out_cap):
        self._order()
        _chk(self.L.ZraHipDecompressBuffer(self.h, d_in, in_size, d_out, out_cap))

    def decompress_ra_batch(self, d_in, in_size, d_out, offsets, sizes, out_offsets):
        import numpy as np
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        s = np.ascontiguousarray(sizes, dtype=np.uint64)
        oo = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
        self._order()
        _chk(self.L.ZraHipDecompressRABatch(self.h, d_in, in_size, d_out, p(o), p(s), p(oo), len(o)))

    def compress_frames(self, d_in, in_size, d_body, d_sizes, level=3, frame_size=65536, checksum=True):
        bsz = ctypes.c_size_t(0)
        self._order()
        _chk(self.L.ZraHipCompressFrames(self.h, d_in, in_size, d_body, d_sizes, ctypes.byref(bsz), level, frame_size, checksum))
        return bsz.value


def stitch_header(frame_sizes, uncompressed_size, frame_size):
    """Host-side seek-table stitch for sharded compression: all ranks' frame sizes -> full ZRA header."""
    import numpy as np
    L = load()
    fs = np.ascontiguousarray(frame_sizes, dtype=np.uint64)
    out = ctypes.create_string_buffer(38 + 5 * (len(fs) + 1))
    hsz = ctypes.c_size_t(0)
    _chk(L.ZraHipStitchHeader(fs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), len(fs), uncompressed_size, frame_size, out, ctypes.byref(hsz)))
    return out.raw[: hsz.value]
