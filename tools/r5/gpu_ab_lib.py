"""round 5: compress speed of the headline configuration through a GIVEN build of the library (plain ctypes, only entry points every
round's build has): usage gpu_ab_lib.py <path to libzra_amd.so> [GiB] [iterations]"""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
lib = ctypes.CDLL(sys.argv[1])
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
class St(ctypes.Structure):
    _fields_ = [("zra", ctypes.c_int), ("zstd", ctypes.c_int)]
vp = ctypes.c_void_p
lib.ZraHipCreateEngine.restype = St; lib.ZraHipCreateEngine.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
lib.ZraHipCompressBuffer.restype = St
lib.ZraHipCompressBuffer.argtypes = [vp, vp, ctypes.c_size_t, vp, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int8, ctypes.c_uint32, ctypes.c_bool]
lib.ZraHipGetKernelStats.restype = None; lib.ZraHipGetKernelStats.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
lib.ZraGetCompressedOutputBufferSize.restype = ctypes.c_size_t; lib.ZraGetCompressedOutputBufferSize.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
eng = vp(); st = lib.ZraHipCreateEngine(ctypes.byref(eng), 0); assert st.zra == 0
dev = torch.device("cuda", 0); N = int(gib * (1 << 30)); fs = 65536
t = torch.from_numpy(np.resize(bench.synth_corpus(64 << 20, 1), N)).to(dev)
out = torch.empty(lib.ZraGetCompressedOutputBufferSize(N, fs) + 64, dtype=torch.uint8, device=dev)
import hashlib, threading, glob
# optional: sample the clock levels and the power the driver reports, every 50 ms, while the calls run (SAMPLE=1)
samples = []
def sampler(stop):
    devs = sorted(glob.glob("/sys/class/drm/card*/device"))
    d = [x for x in devs if os.path.exists(x + "/pp_dpm_fclk")]
    if not d: return
    d = d[0]; hw = glob.glob(d + "/hwmon/hwmon*/power1_average") + glob.glob(d + "/hwmon/hwmon*/power1_input")
    def cur(f):
        try:
            for l in open(d + "/" + f):
                if "*" in l: return l.split(":")[1].strip().replace(" *", "")
        except Exception as e: return "?"
        return "-"
    while not stop.is_set():
        pw = "?"
        try: pw = str(int(open(hw[0]).read()) // 1000000) if hw else "?"
        except Exception: pass
        samples.append((time.time(), cur("pp_dpm_sclk"), cur("pp_dpm_mclk"), cur("pp_dpm_fclk"), cur("pp_dpm_socclk"), pw))
        time.sleep(0.05)
stop = threading.Event(); th = None
if os.environ.get("SAMPLE"): th = threading.Thread(target=sampler, args=(stop,)); th.start()
for it in range(iters):
    osz = ctypes.c_size_t(out.numel())
    torch.cuda.synchronize(); t0 = time.time()
    st = lib.ZraHipCompressBuffer(eng, t.data_ptr(), N, out.data_ptr(), ctypes.byref(osz), 3, fs, True)
    torch.cuda.synchronize(); dt = time.time() - t0
    assert st.zra == 0, (st.zra, st.zstd)
    ks = (ctypes.c_double * 6)(); lib.ZraHipGetKernelStats(eng, ks)
    h = hashlib.sha256(out[:osz.value].cpu().numpy().tobytes()).hexdigest()[:12] if it == 0 else ""
    print("%s: %.1f ms = %.2f GiB/s | mf %.1f ms ent %.1f ms | %d bytes %s" % (os.path.basename(sys.argv[1]), dt * 1e3, gib / dt, ks[0], ks[2], osz.value, h), flush=True)
if th:
    stop.set(); th.join()
    import collections
    c = collections.Counter((a, b, cc, d) for _, a, b, cc, d, _ in samples)
    pw = [int(x[5]) for x in samples if x[5].isdigit()]
    print("   clocks seen (sclk, mclk, fclk, socclk): %s | power W min %s max %s mean %s" % (dict(c), min(pw) if pw else "?", max(pw) if pw else "?", sum(pw) // len(pw) if pw else "?"), flush=True)
