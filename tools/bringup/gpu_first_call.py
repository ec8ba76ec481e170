"""bring-up: what the FIRST host-pointer call of a process pays (verdict r3, weak 8: 2.09 s against 0.29 s steady for 4 GiB).
Times, in a fresh process: library load, the first HIP call (device count), engine creation, first / second / third ZraCompressBuffer of a
host buffer (256 MiB and 4 GiB), first / second ZraDecompressBuffer. torch is not imported: numpy buffers, the C ABI through ctypes."""
import sys, os, time, ctypes
t0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import zra_amd as Z
t1 = time.perf_counter()
print("import zra_amd (dlopen of libzra_amd.so + HIP runtime): %.0f ms" % ((t1 - t0) * 1e3), flush=True)
import bench
base = bench.synth_corpus(64 << 20, 1)
for mib in (256, 4096):
    data = np.resize(base, mib << 20).tobytes()
    for i in range(3):
        t = time.perf_counter(); arc = Z.CompressBuffer(data, 3, 65536, True); dt = time.perf_counter() - t
        print("CompressBuffer %4d MiB call %d: %.1f ms (%.2f GB/s)" % (mib, i + 1, dt * 1e3, len(data) / dt / 1e9), flush=True)
    for i in range(2):
        t = time.perf_counter(); out = Z.DecompressBuffer(arc); dt = time.perf_counter() - t
        print("DecompressBuffer %4d MiB call %d: %.1f ms (%.2f GB/s)" % (mib, i + 1, dt * 1e3, len(data) / dt / 1e9), flush=True)
    assert out == data
