#!/usr/bin/env python3
"""bench.py — hot-path benchmark of the MI355X-native ZRA engine (BASELINE.json metric).

One "step" = one pass of the hot path over the synthetic archive held in HBM:
    CompressBuffer(N bytes @ frameSize, level)  +  batched DecompressRA(Q random queries)
value = uncompressed GiB moved through the path per second = (N + bytes returned by RA) / step time, whole job.
Inputs and outputs are device-resident when the timed region starts (torch is used only for device memory,
streams and torch.distributed). N>1: one process per GPU, frames sharded by index, RCCL all-gather of the
per-rank frame sizes to stitch the global seek table + variable-length gather of the bodies to rank 0.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

GiB = float(1 << 30)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 measured-achievable is reported alongside


def synth_corpus(nbytes, seed=1):
    """'Silesia stand-in' (SURVEY §8d S4): log-like text, structured binary records, skewed bytes and zero runs,
    interleaved in 256 KiB..1 MiB segments. Vectorised numpy so 64 MiB builds in a few seconds."""
    rng = np.random.RandomState(seed)
    out = np.empty(nbytes, dtype=np.uint8)
    pos = 0
    kind = 0
    words = np.frombuffer(b"GET POST PUT /api/v1/items /api/v2/users /static/img/logo.png /healthz status=200 status=404 status=500 "
                          b"INFO WARN ERROR DEBUG latency_ms= bytes= user= session= trace= host=node", dtype=np.uint8)
    while pos < nbytes:
        seg = min(nbytes - pos, int(rng.randint(256, 1025)) << 10)
        k = kind % 4
        kind += 1
        if k == 0 or k == 2:  # log-like text lines built from a small vocabulary + decimal/hex fields
            nlines = seg // 64 + 2
            buf = np.empty((nlines, 96), dtype=np.uint8)
            buf[:] = 32
            ts = (np.arange(nlines) // 7 + 1_700_000_000).astype(np.int64)
            for d in range(10):
                buf[:, 9 - d] = 48 + (ts // (10 ** d)) % 10
            starts = rng.randint(0, len(words) - 24, size=nlines)
            for j in range(24):
                buf[:, 11 + j] = words[starts + j]
            ids = rng.randint(0, 1 << 24, size=nlines)
            hexd = np.frombuffer(b"0123456789abcdef", dtype=np.uint8)
            for d in range(6):
                buf[:, 36 + d] = hexd[(ids >> (4 * d)) & 15]
            starts2 = rng.randint(0, len(words) - 30, size=nlines)
            for j in range(30):
                buf[:, 43 + j] = words[starts2 + j]
            lat = rng.randint(1, 5000, size=nlines)
            for d in range(4):
                buf[:, 77 - d] = 48 + (lat // (10 ** d)) % 10
            lens = rng.randint(60, 96, size=nlines)
            buf[np.arange(nlines), lens - 1] = 10
            mask = np.arange(96)[None, :] < lens[:, None]
            flat = buf[mask]
            out[pos:pos + seg] = flat[:seg] if len(flat) >= seg else np.resize(flat, seg)
        elif k == 1:  # structured binary records: 32-byte records with slowly varying fields
            nrec = seg // 32 + 1
            rec = np.zeros((nrec, 32), dtype=np.uint8)
            ctr = np.arange(nrec, dtype=np.uint32)
            rec[:, 0:4] = ctr.view(np.uint8).reshape(-1, 4)
            rec[:, 4:8] = (ctr // 17 * 2654435761 & 0xFFFFFFFF).astype(np.uint32).view(np.uint8).reshape(-1, 4)
            rec[:, 8:12] = rng.randint(0, 16, size=(nrec, 4))
            rec[:, 16:24] = np.frombuffer(b"RECORD\x00\x01", dtype=np.uint8)
            rec[:, 24:32] = rng.randint(0, 256, size=(nrec, 8)) & rng.randint(0, 256, size=(nrec, 8))
            out[pos:pos + seg] = rec.reshape(-1)[:seg]
        else:  # skewed bytes (Huffman-only) with zero runs
            a = rng.randint(0, 256, size=seg).astype(np.uint8) & rng.randint(0, 256, size=seg).astype(np.uint8) & rng.randint(0, 256, size=seg).astype(np.uint8)
            z = rng.randint(0, seg - 4096) if seg > 8192 else 0
            a[z:z + 4096] = 0
            out[pos:pos + seg] = a
        pos += seg
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cgroup_cores():
    """CPU quota of the container in cores (cgroup v2 cpu.max), None when unlimited or unreadable."""
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if a == "max" else round(int(a) / float(b), 2)
    except Exception:
        return None


def _socket_cores():
    """Physical cores of one socket as /proc/cpuinfo states them (the host's, not the container's quota)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("cpu cores"):
                return int(line.split(":", 1)[1])
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_baseline(sample, frame_size, level, nq, qsize, seed=42):
    """Reference CPU path timed on the host: ZRA container logic (oracle/zo_zra.c, a port of zra.cpp:194-296) over the real
    dependency libzstd 1.4.9 when the image has it, else over the oracle's own restatement. 1 thread (the reference is single-threaded).
    The timed spans contain the C calls only: buffers are allocated and touched before, the query loop runs in C."""
    sys.path.insert(0, os.path.join(HERE, "tests"))
    import oracle_lib as O
    backend = "zl" if O.have_libzstd() else "zo"
    L = O.lib()
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    data = np.ascontiguousarray(sample)
    n = data.size
    cap = L.zo_zra_output_bound(n, frame_size, 0)
    out = np.zeros(cap, dtype=np.uint8)                       # touched: no page faults inside the timed call
    osz = sz(0)
    fcomp = getattr(L, backend + "_zra_compress_buffer")
    fcomp.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz), ctypes.c_int, ctypes.c_uint32, ctypes.c_int, sz]
    t0 = time.perf_counter()
    st = fcomp(data.ctypes.data, n, out.ctypes.data, cap, ctypes.byref(osz), level, frame_size, 1, 0)
    t1 = time.perf_counter()
    assert st.tup() == (0, 0), st.tup()
    arc = out[: osz.value].tobytes()
    rng = np.random.RandomState(seed)
    offs = np.ascontiguousarray(rng.randint(0, n - qsize - 1, size=nq).astype(np.uint64))
    fra = getattr(L, backend + "_zra_decompress_ra_many")
    fra.restype = O.Status
    fra.argtypes = [vp, sz, vp, sz, vp, sz]
    abuf = np.frombuffer(arc, dtype=np.uint8)
    obuf = np.zeros(qsize, dtype=np.uint8)
    t2 = time.perf_counter()
    st = fra(abuf.ctypes.data, len(arc), obuf.ctypes.data, qsize, offs.ctypes.data, nq)
    t3 = time.perf_counter()
    assert st.tup() == (0, 0), st.tup()
    # (ii) of SURVEY §8d: best-case CPU, NOT the reference (which is single-threaded): frames statically partitioned over T threads,
    # one libzstd context per thread (ctypes releases the GIL). Bounded: every thread compresses a 64 MiB slice.
    import threading
    T = max(1, min(64, (os.cpu_count() or 1)))
    sl = min(n, 64 << 20) // frame_size * frame_size
    outs = [np.zeros(L.zo_zra_output_bound(sl, frame_size, 0), dtype=np.uint8) for _ in range(T)]
    def work(k):
        o = sz(0)
        fcomp(data.ctypes.data, sl, outs[k].ctypes.data, outs[k].size, ctypes.byref(o), level, frame_size, 1, 0)
    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    t4 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    t5 = time.perf_counter()
    mt = T * sl / GiB / (t5 - t4)
    comp = n / GiB / (t1 - t0)
    ra = nq * qsize / GiB / (t3 - t2)
    combined = (n + nq * qsize) / GiB / ((t1 - t0) + (t3 - t2))
    return {"value": round(combined, 4), "unit": "GiB/s", "cores": 1, "kind": "port",
            "sample": "%d MiB of the same corpus: CompressBuffer L%d/%d KiB + %d DecompressRA queries of %d B (the step's own blend of bytes to queries); container port (oracle/zo_zra.c) over %s, one ZSTD_CCtx per CompressBuffer and one ZSTD_DCtx per DecompressRA like zra.cpp:209,271; timed spans = the C calls only"
                      % (n >> 20, level, frame_size >> 10, nq, qsize, "libzstd " + O.lib().zo_libzstd_version().decode() if backend == "zl" else "the oracle's C restatement"),
            "compress_gibs": round(comp, 4), "ra_gibs": round(ra, 4), "ra_us_per_query": round((t3 - t2) / nq * 1e6, 1),
            "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(),
            "best_case_all_threads": {"compress_gibs": round(mt, 3), "threads": T, "host_cpus": os.cpu_count(),
                                      "cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                                      "cgroup_cpu_max": (open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None),
                                      "cpu_limit_cores": _cgroup_cores(),
                                      "one_thread_in_this_run_gibs": round(comp, 4),
                                      "full_socket_extrapolation_gibs": round(comp * _socket_cores(), 2), "socket_cores_assumed": _socket_cores(),
                                      "note": "not the reference (single-threaded): %d threads x %d MiB, one libzstd context each; the threads share the container's CPU "
                                              "quota (cpu_limit_cores), so this is NOT what the whole socket would do — full_socket_extrapolation_gibs = the 1-thread rate x "
                                              "the socket's physical cores (frames are independent: an upper bound that ignores memory-bandwidth and turbo effects)" % (T, sl >> 20)}}, arc


def c4_share(Z, eng, torch, dev, gib=2.0, cpu_mib=32):
    """BASELINE config 4's one-GPU share, outside the timed region: log-like synthetic data (tests/corpus.py: gen_loglike), level 9 at
    256 KiB frames — the second compress configuration BASELINE.json names (reference call site zra.cpp:219 at :210 level 9). GPU:
    CompressBuffer and full DecompressBuffer on `gib` GiB; CPU leg: the same container port as cpu_baseline() over a `cpu_mib` MiB sample
    (level 9 runs at ~0.03 GiB/s on one core), which is also the bit-exactness gate of this configuration."""
    sys.path.insert(0, os.path.join(HERE, "tests"))
    import corpus as C
    import oracle_lib as O
    fs, level = 262144, 9
    base = np.frombuffer(C.gen_loglike(32 << 20, seed=4), dtype=np.uint8)
    N = int(gib * GiB) // fs * fs
    d_in = torch.from_numpy(np.resize(base, N)).to(dev)
    d_arc = torch.empty(Z.GetOutputBufferSize(N, fs) + 64, dtype=torch.uint8, device=dev)
    n = eng.compress(d_in.data_ptr(), N, d_arc.data_ptr(), level, fs, True)                 # warm (scratch of this configuration)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = eng.compress(d_in.data_ptr(), N, d_arc.data_ptr(), level, fs, True)
    torch.cuda.synchronize(); tc = time.perf_counter() - t0
    ks = eng.kernel_stats()
    d_out = torch.empty(N, dtype=torch.uint8, device=dev)
    eng.decompress(d_arc.data_ptr(), n, d_out.data_ptr(), N)                                   # warm
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.decompress(d_arc.data_ptr(), n, d_out.data_ptr(), N)
    torch.cuda.synchronize(); td = time.perf_counter() - t0
    if not torch.equal(d_out, d_in):
        raise SystemExit("c4_share: GPU decompress does not restore the input")
    # CPU leg + gate on a sample
    samp = min(N, cpu_mib << 20) // fs * fs
    backend = "zl" if O.have_libzstd() else "zo"
    L = O.lib()
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    data = np.ascontiguousarray(np.resize(base, samp))
    cap = L.zo_zra_output_bound(samp, fs, 0)
    out = np.zeros(cap, dtype=np.uint8)
    osz = sz(0)
    fcomp = getattr(L, backend + "_zra_compress_buffer")
    fcomp.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz), ctypes.c_int, ctypes.c_uint32, ctypes.c_int, sz]
    t0 = time.perf_counter()
    st = fcomp(data.ctypes.data, samp, out.ctypes.data, cap, ctypes.byref(osz), level, fs, 1, 0)
    tcpu = time.perf_counter() - t0
    assert st.tup() == (0, 0), st.tup()
    d_s = torch.empty(Z.GetOutputBufferSize(samp, fs) + 64, dtype=torch.uint8, device=dev)
    ns = eng.compress(d_in.data_ptr(), samp, d_s.data_ptr(), level, fs, True)
    same = d_s[:ns].cpu().numpy().tobytes() == out[: osz.value].tobytes()
    if not same:
        raise SystemExit("c4_share: GPU archive differs from the CPU path")
    del d_in, d_arc, d_out, d_s
    eng.release_scratch()
    # roofline of this configuration's dominant kernel (zra_mf_hc_kernel, the wave-cooperative hash-chain finder): algorithmic bytes =
    # N_in + C_out per call over its launches' HIP-event time in THIS run; traffic = HBM-side requests of the same kernel
    # from the counter passes stored in profiles/traffic.json (c4_hc_r06: level 9 @ 256 KiB, 1 GiB), scaled by input size
    roof = None
    try:
        mf_ms, nl = float(ks["mf_ms"]), max(1, int(ks["mf_launches"]))
        alg = N + n
        ach = alg / 1e9 / (mf_ms / 1e3) if mf_ms > 0 else 0.0
        traffic, reqs = None, None
        tj = json.load(open(os.path.join(HERE, "profiles", "traffic.json"))).get("c4_hc_r06")
        if tj:
            e = tj["zra_mf_hc_kernel"]; sc = N / float(tj["input_bytes"])
            traffic = int((e["ea_rdreq"] * 64 + e["ea_wrreq"] * 32) * sc / nl)
            rd_s, wr_s = e["ea_rdreq"] * sc / (mf_ms / 1e3), e["ea_wrreq"] * sc / (mf_ms / 1e3)
            reqs = {"reads_per_s": round(rd_s / 1e9, 2), "read_ceiling_per_s": 48.3, "writes_per_s": round(wr_s / 1e9, 2), "write_ceiling_per_s": 23.0, "unit": "1e9 requests/s",
                    "sum_of_fractions": round(rd_s / 48.3e9 + wr_s / 23.0e9, 3), "l2_hit_rate": round(e["tcc_hit"] / (e["tcc_hit"] + e["tcc_miss"]), 3),
                    "wave_cycles_waiting": round(e["sq_wait_any"] / e["sq_wave_cycles"], 3),
                    "note": "requests from profiles/r06_pmc_hc.txt (1 GiB of the same data and configuration) over this run's launch time; ceilings: profiles/r02_pmc_calibration.txt"}
        roof = {"bound": "hbm", "kernel": "zra_mf_hc_kernel", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6),
                "traffic": traffic, "launch_ms": round(mf_ms / nl, 3), "launches_per_call": nl, "algorithmic_bytes_per_launch": int(alg / nl), "requests": reqs}
    except Exception:
        roof = None
    return {"roofline": roof, "workload": "%.3g GiB log-like synthetic (tests/corpus.py gen_loglike, 32 MiB tiled), frameSize=256 KiB, level 9, checksum on" % (N / GiB),
            "compress_gibs": round(N / GiB / tc, 3), "decompress_gibs": round(N / GiB / td, 3), "compression_ratio": round(N / n, 3),
            "mf_ms": round(ks["mf_ms"], 1), "mf_launches": ks["mf_launches"],
            "cpu": {"compress_gibs": round(samp / GiB / tcpu, 4), "cores": 1, "kind": "port", "sample_mib": samp >> 20,
                    "over": "libzstd " + O.lib().zo_libzstd_version().decode() if backend == "zl" else "the oracle's C restatement"},
            "bit_exact_gate": "archive bytes identical to CPU path (%d MiB sample)" % (samp >> 20)}


def request_roofline(req, launch_ms):
    """HBM-side request rates of the dominant launch (PMC bytes of profiles/traffic.json over the launch time measured in THIS run) as
    fractions of the chip's measured random-request ceilings. None without a traffic entry."""
    if not req or launch_ms <= 0:
        return None
    t = launch_ms / 1e3
    r = dict(req)
    r["fetch_per_s"] = round(req["fetch_lines_per_launch"] / t / 1e9, 2)
    r["write_per_s_low_high"] = [round(req["write_bytes_per_launch"] / 64 / t / 1e9, 2), round(req["write_bytes_per_launch"] / 32 / t / 1e9, 2)]
    r["unit"] = "1e9 requests/s"
    fr = req["fetch_lines_per_launch"] / t / req["read_ceiling_per_s"]
    r["frac_of_ceilings_low_high"] = [round(fr + req["write_bytes_per_launch"] / 64 / t / req["write_ceiling_per_s"], 3),
                                      round(fr + req["write_bytes_per_launch"] / 32 / t / req["write_ceiling_per_s"], 3)]
    return r


def kernel_source_sha():
    """sha256 over the kernel sources (zra_amd/csrc, sorted by name) with comments and white space taken out: ties a PMC measurement
    in profiles/traffic.json to the CODE of a build (a reworded comment does not make a measurement stale)."""
    import hashlib, re
    h = hashlib.sha256()
    d = os.path.join(HERE, "zra_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            t = open(os.path.join(d, f), "r", errors="replace").read()
            t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
            t = re.sub(r"//[^\n]*", " ", t)
            t = re.sub(r"\s+", " ", t)
            h.update(f.encode()); h.update(t.encode())
    return h.hexdigest()[:16]


def kernel_resources(names):
    """VGPRs / scratch / spill counts of the named kernels as the compiler reported them for THIS build (zra_amd/build/kernel_resources.json,
    written by zra_amd/build.py from -Rpass-analysis=kernel-resource-usage): in the line so that a reader sees them without a rebuild."""
    try:
        res = json.load(open(os.path.join(HERE, "zra_amd", "build", "kernel_resources.json")))
        return {k: {f: res[k].get(f) for f in ("vgprs", "sgpr_spill", "vgpr_spill", "scratch_bytes", "lds_bytes", "waves_per_simd")} for k in names if k in res}
    except Exception:
        return None


def under_profiler():
    """rocprofv3 preloads a library that initialises the GPU in every process of the tree: programs started from here (rocm-smi, the
    CLI counterpart) would then be 'an exec after the GPU was initialised', which the GPU pool refuses. The probes that start programs
    are skipped under the profiler (the profiled run is there for the kernel table, the plain run carries the probes)."""
    return any(k.startswith(("ROCPROFILER", "ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def gpu_clocks():
    """Current shader / memory clocks, power and performance level of GPU 0 as rocm-smi reports them (None when it cannot be asked):
    recorded before and after the timed region, because the same build reads two different match-finder times on different boxes."""
    import subprocess
    if under_profiler():
        return None
    try:
        r = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showperflevel", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        j = json.loads(r.stdout)
        c = j.get("card0", next(iter(j.values())))
        keep = {}
        for k, v in c.items():
            kl = k.lower()
            if "sclk" in kl or "mclk" in kl or "fclk" in kl or "power" in kl or "performance level" in kl or ("temperature" in kl and ("junction" in kl or "edge" in kl)):
                keep[k] = v
        return keep or None
    except Exception:
        return None


def host_pointer_calls(level, fs, gib=4):
    """The reference's own calling convention (host pointers, zra.h) measured OUTSIDE the timed region through the CLI counterpart's
    benchmark mode (zra_amd/tools/zratool_amd b): in-memory CompressBuffer / DecompressBuffer on `gib` GiB of the bench corpus (steady
    state of the C ABI, third repetition) and the streaming Compressor / FullDecompressor with the reference tool's 10 MB buffers."""
    import subprocess, re
    tool = os.path.join(HERE, "zra_amd", "tools", "zratool_amd")
    if under_profiler() or not os.path.exists(tool):
        return None
    path = "/tmp/zra_bench_host_%d.bin" % os.getpid()
    try:
        base = synth_corpus(64 << 20, seed=1).tobytes()
        with open(path, "wb") as f:
            for _ in range(int(gib * 16)):
                f.write(base)
        r = subprocess.run([tool, "b", path, str(level), str(fs), "10"], capture_output=True, text=True, timeout=600)
        out = {"bytes": int(gib * 16) * len(base), "tool": "zratool_amd b (level %d, frameSize %d, 10 MB streaming buffers)" % (level, fs)}
        for line in r.stdout.splitlines():
            m = re.match(r"(in-memory compress|in-memory decompress|streaming compress|streaming decompress)\s*:\s*([\d.]+) ms\s+([\d.]+) MB/s", line)
            if m:
                out[m.group(1).replace(" ", "_").replace("-", "_") + "_first_call"] = {"ms": float(m.group(2)), "gbs": round(float(m.group(3)) / 1e3, 3)}
            m = re.match(r"C ABI rep 2: compress\s+([\d.]+) ms\s+([\d.]+) MB/s \(status (\d+)\)\s+decompress\s+([\d.]+) ms\s+([\d.]+) MB/s \(status (\d+), (\w+)\)", line)
            if m:
                out["ZraCompressBuffer"] = {"ms": float(m.group(1)), "gbs": round(float(m.group(2)) / 1e3, 3), "status": int(m.group(3))}
                out["ZraDecompressBuffer"] = {"ms": float(m.group(4)), "gbs": round(float(m.group(5)) / 1e3, 3), "status": int(m.group(6)), "roundtrip": m.group(7)}
            m = re.match(r"random access (\d+) B @ (\d+) \((in-memory|streaming)\) :\s+([\d.]+) ms\s+(\w+)", line)
            if m:
                out["random_access_%s" % m.group(3).replace("-", "_")] = {"bytes": int(m.group(1)), "ms": float(m.group(4)), "check": m.group(5)}
        # the streaming classes are named as in zra.hpp
        if "streaming_compress_first_call" in out: out["Compressor_10MB_buffers"] = out.pop("streaming_compress_first_call")
        if "streaming_decompress_first_call" in out: out["FullDecompressor_10MB_buffers"] = out.pop("streaming_decompress_first_call")
        return out
    except Exception as e:
        return {"error": repr(e)}
    finally:
        for suffix in ("", ".bench.zra", ".bench.out"):
            try:
                os.remove(path + suffix)
            except OSError:
                pass


def ra_latency_probe(Z, eng, d_arc, arc_size, d_in, N, qb, torch):
    """Latency of small random-access batches (outside the timed region): batch sizes 1 / 64 / 4096 through the device-pointer call
    (ZraHipDecompressRABatch) and, per query, through the reference's host-pointer call (ZraDecompressRA on a host copy of the archive)."""
    out = {}
    rng = np.random.RandomState(7)
    dev = d_in.device
    for bs, reps in ((1, 40), (64, 20), (4096, 8)):
        d_o = torch.empty(bs * qb + 64, dtype=torch.uint8, device=dev)
        sizes = np.full(bs, qb, dtype=np.uint64); oo = np.arange(bs, dtype=np.uint64) * qb
        ts = []
        for r in range(reps + 2):
            offs = rng.randint(0, N - qb - 1, size=bs).astype(np.uint64)
            torch.cuda.synchronize(); t = time.perf_counter()
            eng.decompress_ra_batch(d_arc.data_ptr(), arc_size, d_o.data_ptr(), offs, sizes, oo)
            ts.append(time.perf_counter() - t)
        if not torch.equal(d_o[:qb], d_in[int(offs[0]): int(offs[0]) + qb]):
            raise SystemExit("RA latency probe: wrong bytes")
        ts = sorted(ts[2:])
        out["device_batch_%d" % bs] = {"us_per_call_median": round(ts[len(ts) // 2] * 1e6, 1), "us_per_query": round(ts[len(ts) // 2] * 1e6 / bs, 2)}
    # host-pointer ABI, one query per call (the reference's calling convention); the archive copy to the host is not timed
    h_arc = d_arc[:arc_size].cpu().numpy()
    L = Z.load()
    abuf = ctypes.c_void_p(h_arc.ctypes.data)
    obuf = ctypes.create_string_buffer(qb)
    ts = []
    for r in range(42):
        o = int(rng.randint(0, N - qb - 1))
        t = time.perf_counter()
        st = L.ZraDecompressRA(abuf, arc_size, obuf, o, qb)
        ts.append(time.perf_counter() - t)
        if st.zra != 0:
            raise SystemExit("ZraDecompressRA failed in the latency probe")
    ts = sorted(ts[2:])
    out["host_ZraDecompressRA"] = {"us_per_call_median": round(ts[len(ts) // 2] * 1e6, 1)}
    return out


def choose_comm(eng, rank, world, one_gpu, dev, dist, sharding, torch, want, group=None):
    """The communicator of a multi-rank run: RCCL called by the library itself (ncclAllGather / grouped ncclSend+ncclRecv on the engine's
    stream), or the host transport over torch.distributed — the fallback, and the only choice when every rank sits on one GPU (RCCL
    refuses duplicate devices). Every rank leaves with the SAME transport: one rank without an RCCL communicator sends everybody to the
    fallback (tests/test_multi_rank.py runs this with a communicator that fails on one rank only)."""
    comm, transport = None, "none"
    if want == "rccl":
        try:
            comm, transport = sharding.Comm.rccl(eng, rank, world), "rccl (direct)"
        except Exception as e:
            sys.stderr.write("bench: direct RCCL communicator failed (%r); falling back to torch.distributed\n" % (e,))
        flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cpu" if one_gpu else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and comm is not None:
            comm.close()
            comm = None
    if comm is None:
        comm = sharding.Comm.torch_dist(eng, group=group) if group is not None else sharding.Comm.torch_dist(eng)
        transport = "torch.distributed/" + dist.get_backend()
    return comm, transport


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size-gib", type=float, default=16.0, help="uncompressed bytes per GPU (BASELINE metric: 16 GiB)")
    ap.add_argument("--frame-kib", type=int, default=64)
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--queries", type=int, default=1_000_000)
    ap.add_argument("--query-bytes", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="nothing but the timed steps (PMC passes: tools/pmc_bench.sh)")
    ap.add_argument("--no-host-calls", action="store_true", help="skip the host-pointer extras (4 GiB through zratool_amd b, outside the timed region)")
    ap.add_argument("--no-c4", action="store_true", help="skip BASELINE config 4's one-GPU share (2 GiB log-like, level 9 @ 256 KiB, outside the timed region)")
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON: libraries that print banners on first use (RCCL's version block, gloo's rank lines)
    # get stderr instead; the real stdout is kept aside for the last line
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import zra_amd as Z

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # ZRA_BENCH_ONE_GPU=1: dry-run the N>1 control flow with every rank on GPU 0 and gloo (boxes with a single GPU); never used for numbers
    one_gpu = os.environ.get("ZRA_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    Z.load()
    eng = Z.Engine(local_rank)
    comm, comm_side, transport = None, None, "none"
    if world > 1:
        from zra_amd import sharding
        want_tr = os.environ.get("ZRA_BENCH_TRANSPORT", "torch" if one_gpu else "rccl")
        comm, transport = choose_comm(eng, rank, world, one_gpu, dev, dist, sharding, torch, want_tr)
        # a second communicator with a stream of its own: the archive's bodies travel to the root on it WHILE the ranks serve queries from their
        # shards on the first one (round 6; ZRA_BENCH_GATHER_SYNC=1: gather, then serve, as before). Its host transport gets a group of its own:
        # the two communicators' collectives run from two threads.
        if os.environ.get("ZRA_BENCH_GATHER_SYNC") != "1":
            try:
                g2 = dist.new_group(backend=dist.get_backend()) if not transport.startswith("rccl") else None
                comm_side, tr2 = choose_comm(eng, rank, world, one_gpu, dev, dist, sharding, torch, want_tr if transport.startswith("rccl") else "torch", group=g2)
                comm_side.use_own_stream()
                transport += " + a second communicator (%s) for the gather, beside the serving" % tr2
            except Exception as e:
                sys.stderr.write("bench: no second communicator (%r): gather and serve in sequence\n" % (e,))
                comm_side = None

    fs = args.frame_kib << 10
    N = int(args.size_gib * GiB) // fs * fs            # per-GPU bytes (weak scaling: fixed per rank)
    base = synth_corpus(64 << 20, seed=1 + rank)
    tbase = torch.from_numpy(base).to(dev)
    reps = (N + len(base) - 1) // len(base)
    d_in = tbase.repeat(reps)[:N].contiguous()
    del tbase
    nframes = N // fs
    bound = Z.GetOutputBufferSize(N, fs)
    d_arc = torch.empty(bound + 64, dtype=torch.uint8, device=dev) if world == 1 else None    # N>1: the shard object holds the frames
    # RA workload: offsets uniform over the whole (all ranks') uncompressed range from a fixed seed; N>1: routed to the owners
    q = args.queries
    qb = args.query_bytes
    rng = np.random.RandomState(42 + rank)
    offs = rng.randint(0, N * world - qb - 1, size=q).astype(np.uint64)
    sizes = np.full(q, qb, dtype=np.uint64)
    oofs = (np.arange(q, dtype=np.uint64) * qb)
    d_ra = torch.empty(q * qb + 64, dtype=torch.uint8, device=dev)

    # ---- bit-exactness gate + CPU baseline (rank 0): first 1 GiB of the same corpus through the CPU path
    cpu = None
    gate = "skipped"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # N=1 only (bench contract); N>1 runs reuse the N=1 gate
        samp = min(N, 1 << 30)
        # the CPU sample keeps the step's own blend of compressed bytes to queries (16 GiB : 1 M -> 1 GiB : 62.5 k)
        cpu_nq = max(1000, int(round(q * samp / float(N))))
        cpu, cpu_arc = cpu_baseline(np.resize(base, samp), fs, args.level, cpu_nq, qb)
        d_s = d_in[:samp]
        d_o = torch.empty(Z.GetOutputBufferSize(samp, fs) + 64, dtype=torch.uint8, device=dev)
        n = eng.compress(d_s.data_ptr(), samp, d_o.data_ptr(), args.level, fs, True)
        gpu_arc = d_o[:n].cpu().numpy().tobytes()
        gate = "archive bytes identical to CPU path (%d MiB sample)" % (samp >> 20) if gpu_arc == cpu_arc else "MISMATCH"
        if gpu_arc != cpu_arc:
            raise SystemExit("bit-exactness gate failed: GPU archive differs from the CPU path")
        d_back = torch.empty(samp, dtype=torch.uint8, device=dev)
        eng.decompress(d_o.data_ptr(), n, d_back.data_ptr(), samp)
        if not torch.equal(d_back, d_s):
            raise SystemExit("bit-exactness gate failed: GPU decompress does not restore the input")
        del d_o, d_back

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    comp_ms = []
    ra_ms = []
    mf_ms = []
    dec_stats = []
    dec_stage = []
    arc_size = 0

    def step():
        nonlocal arc_size
        t0 = time.perf_counter()
        if world == 1:
            arc_size = eng.compress(d_in.data_ptr(), N, d_arc.data_ptr(), args.level, fs, True)
            mf_ms.append(eng.kernel_stats())
            step.tele = eng.launch_telemetry()
        else:
            # sharded (include/zra_hip.h, distributed archive): local frames through the same kernels, all-gather of the frame sizes,
            # seek table stitched on every rank, frame bodies gathered to rank 0 behind the header (north_star: "final gather")
            old = getattr(step, "shard", None)
            if old is not None:
                old.close()
            step.shard = comm.compress(d_in.data_ptr(), N, N * world, args.level, fs, True)
            mf_ms.append(eng.kernel_stats())
            arc_size = step.shard.archive_size()
            if rank == 0 and (getattr(step, "root", None) is None or step.root.numel() < arc_size):
                step.root = torch.empty(arc_size + (arc_size >> 4) + 64, dtype=torch.uint8, device=dev)
            if comm_side is not None:
                comm_side.gather_archive_begin(step.shard, 0, step.root.data_ptr() if rank == 0 else 0, step.root.numel() if rank == 0 else 0)
            else:
                comm.gather_archive(step.shard, 0, step.root.data_ptr() if rank == 0 else 0, step.root.numel() if rank == 0 else 0)
        if comm_side is None:
            torch.cuda.synchronize()
        # (with the gather on its own stream a DEVICE-wide wait here would wait for it too and put it back in front of the serving; the
        #  compress call has returned, so its outputs are complete (include/zra_hip.h) — t1 only splits the step for the two reported legs)
        t1 = time.perf_counter()
        # RA over this rank's own shard (the archive stays sharded for serving; queries are routed to the owner)
        if world == 1:
            eng.decompress_ra_batch(d_arc.data_ptr(), arc_size, d_ra.data_ptr(), offs, sizes, oofs)
            dec_stats.append(eng.kernel_stats()); dec_stage.append(eng.decode_stage_stats())
        else:
            # sharded serving: this rank's queries go over the WHOLE range; slices travel to the owners, answers come back
            comm.serve(step.shard, offs, sizes, oofs, d_ra.data_ptr())
            dec_stats.append(eng.kernel_stats()); dec_stage.append(eng.decode_stage_stats())
            if comm_side is not None:
                t_s = time.perf_counter()
                got = comm_side.gather_archive_end()       # one join at the end of the step
                step.gather_wait_ms = getattr(step, "gather_wait_ms", []) + [(time.perf_counter() - t_s) * 1e3]
                if rank == 0 and got != arc_size:
                    raise SystemExit("gathered archive has %d bytes, the shards say %d" % (got, arc_size))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        comp_ms.append((t1 - t0) * 1e3)
        ra_ms.append((t2 - t1) * 1e3)

    # The timed RA leg decodes every touched frame IN FULL and verifies its XXH64 (ZRA_HIP_OPT_RA_WHOLE_FRAMES), which is what the reference's
    # DecompressRA does (zra.cpp:280-293) and what cpu_baseline times; the library's default for batches — stop a frame at the last byte a
    # query needs — is reported beside it as ra_early_stop, outside the timed region. (ZRA_BENCH_RA_EARLY=1: round 5's headline, for A/Bs.)
    lib_opts = Z.load()
    opts_before = lib_opts.ZraHipGetOptions()
    ra_whole_timed = os.environ.get("ZRA_BENCH_RA_EARLY") != "1"
    if ra_whole_timed:
        lib_opts.ZraHipSetOptions(opts_before | 8)
    for _ in range(args.warmup):
        step()
    comp_ms.clear(); ra_ms.clear(); mf_ms.clear(); dec_stats.clear(); dec_stage.clear()
    clocks_before = gpu_clocks() if rank == 0 else None
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    clocks_after = gpu_clocks() if rank == 0 else None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # spot-check RA output of the last step against the resident input
    chk = rng.randint(0, q, size=64)
    other = {}                                            # N>1: the corpus of the rank that owns the bytes, regenerated on the host
    for i in chk:
        o = int(offs[i])
        got = d_ra[int(oofs[i]): int(oofs[i]) + qb]
        for r in range(o // N, (o + qb - 1) // N + 1):
            a, b = max(o, r * N), min(o + qb, (r + 1) * N)
            if r == rank:
                want = d_in[a - r * N: b - r * N]
            else:
                if r not in other:
                    other[r] = synth_corpus(64 << 20, seed=1 + r)
                idx = (np.arange(a - r * N, b - r * N) % len(other[r]))
                want = torch.from_numpy(other[r][idx]).to(dev)
            if not torch.equal(got[a - o: b - o], want):
                raise SystemExit("RA result mismatch at query %d" % i)

    # SURVEY §8d RA size classes (outside the timed region, single rank): 64 KiB unaligned (touches 2 frames) and 1 MiB queries
    ra_classes = {}
    if world == 1 and not args.timed_only:
        for qsz, nq2 in ((65536, 100000), (1 << 20, 8000)):
            o2 = rng.randint(0, N - qsz - 1, size=nq2).astype(np.uint64)
            s2 = np.full(nq2, qsz, dtype=np.uint64); oo2 = (np.arange(nq2, dtype=np.uint64) * qsz)
            d2 = torch.empty(nq2 * qsz, dtype=torch.uint8, device=dev)
            eng.decompress_ra_batch(d_arc.data_ptr(), arc_size, d2.data_ptr(), o2, s2, oo2)     # warm
            torch.cuda.synchronize(); tq = time.perf_counter()
            eng.decompress_ra_batch(d_arc.data_ptr(), arc_size, d2.data_ptr(), o2, s2, oo2)
            torch.cuda.synchronize(); dq = time.perf_counter() - tq
            k = int(rng.randint(0, nq2))
            if not torch.equal(d2[k * qsz:(k + 1) * qsz], d_in[int(o2[k]): int(o2[k]) + qsz]):
                raise SystemExit("RA size-class result mismatch")
            ra_classes["%d_KiB" % (qsz >> 10)] = {"queries": nq2, "us_per_query": round(dq / nq2 * 1e6, 3), "gibs_returned": round(nq2 * qsz / GiB / dq, 3)}
            del d2

    # the other RA mode, outside the timed region: the same 1 M queries with the option flipped (whole frames when the timed leg stopped
    # early, early stop when the timed leg — the default since round 6 — decoded whole frames)
    ra_other = None
    if world == 1 and not args.timed_only:
        lib_opts.ZraHipSetOptions((opts_before & ~8) if ra_whole_timed else (opts_before | 8))
        try:
            eng.decompress_ra_batch(d_arc.data_ptr(), arc_size, d_ra.data_ptr(), offs, sizes, oofs)     # warm
            torch.cuda.synchronize(); tq = time.perf_counter()
            eng.decompress_ra_batch(d_arc.data_ptr(), arc_size, d_ra.data_ptr(), offs, sizes, oofs)
            torch.cuda.synchronize(); dq = time.perf_counter() - tq
            ra_other = {"us_per_query": round(dq / q * 1e6, 3), "gibs_returned": round(q * qb / GiB / dq, 3),
                        "note": ("a frame stops at the last byte a query needs and its XXH64 is not seen (the library's default for batches); the timed leg decodes whole frames"
                                 if ra_whole_timed else
                                 "every touched frame decoded in full and its XXH64 verified (ZRA_HIP_OPT_RA_WHOLE_FRAMES): the like-for-like of cpu_baseline.ra_us_per_query")}
        finally:
            lib_opts.ZraHipSetOptions(opts_before | 8 if ra_whole_timed else opts_before)
    ra_whole = ra_other if not ra_whole_timed else None
    ra_early = ra_other if ra_whole_timed else None

    lib_opts.ZraHipSetOptions(opts_before)              # (the probes below measure the library's defaults)
    ra_latency = None
    if world == 1 and not args.no_cpu_baseline:
        ra_latency = ra_latency_probe(Z, eng, d_arc, arc_size, d_in, N, qb, torch)

    host_calls = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_host_calls:
        host_calls = host_pointer_calls(args.level, fs)

    # BASELINE config 4's one-GPU share (level 9 @ 256 KiB, log-like), outside the timed region; needs the oracle for its gate + CPU leg
    c4 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.timed_only and not args.no_c4:
        c4 = c4_share(Z, eng, torch, dev)

    # compressed bytes of the frames the timed queries touch (for the random-access leg's roofline): the archive's own seek table
    ra_touched = None
    if rank == 0 and world == 1:
        try:
            hdr = d_arc[:38].cpu().numpy().tobytes()
            tsz = int.from_bytes(hdr[26:30], "little")
            tab = d_arc[38:38 + 5 * tsz].cpu().numpy().reshape(tsz, 5).astype(np.uint64)
            ent = tab[:, 0] | (tab[:, 1] << np.uint64(8)) | (tab[:, 2] << np.uint64(16)) | (tab[:, 3] << np.uint64(24)) | (tab[:, 4] << np.uint64(32))
            f0 = (offs // np.uint64(fs)).astype(np.int64); f1 = ((offs + np.uint64(qb - 1)) // np.uint64(fs)).astype(np.int64)
            hit = np.zeros(tsz - 1, dtype=bool); hit[f0] = True; hit[f1] = True
            ra_touched = {"frames": int(hit.sum()), "compressed_bytes": int((ent[1:][hit] - ent[:-1][hit]).sum())}
        except Exception:
            ra_touched = None

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        moved = (N + q * qb) * world
        value = moved / GiB / (elapsed / args.steps)
        ratio = N / max(1, (arc_size if world == 1 else arc_size / world))
        # roofline of the dominant kernel = the match finder (zra_mf_dfast_kernel at levels 3-4): per-launch duration from HIP events recorded on the
        # engine's stream around each launch; algorithmic bytes = N_in + C_out of the frames of that launch (SURVEY §8d:
        # 65536*(1+1/ratio) per 64 KiB frame x frames per launch)
        st = mf_ms[-1] if mf_ms else dict(mf_ms=0, mf_launches=0, ent_ms=0, ent_launches=0)
        c_out = (arc_size if world == 1 else arc_size / world)
        alg_bytes = N + c_out
        launches = max(1, st["mf_launches"])
        mf_launch_ms = float(np.mean([x["mf_ms"] / max(1, x["mf_launches"]) for x in mf_ms])) if mf_ms else 0.0
        ent_launch_ms = float(np.mean([x["ent_ms"] / max(1, x["ent_launches"]) for x in mf_ms])) if mf_ms else 0.0
        dec_launch_ms = float(np.mean([x["dec_ms"] / max(1, x["dec_launches"]) for x in dec_stats])) if dec_stats else 0.0
        alg_per_launch = alg_bytes / launches
        achieved = alg_per_launch / 1e9 / (mf_launch_ms / 1e3) if mf_launch_ms > 0 else 0.0
        # HBM-side bytes of one match-finder launch: PMC passes (FETCH_SIZE, WRITE_SIZE, separate runs) of THIS workload, stored with the
        # commit they were taken at (profiles/traffic.json, tools/pmc_bench.sh); per-frame scaling only if the size differs
        # which match finder the level runs (frame sizes of the 128 KiB class; the kernel-per-strategy table of zra_encode.hip)
        lv = 3 if args.level == 0 else args.level
        mf_kernel = ("zra_mf_fast_kernel" if lv <= 2 else "zra_mf_dfast_fl_kernel" if lv <= 4 else "zra_mf_hc_kernel" if lv <= 10 else
                     "zra_mf_kernel" if lv <= 12 or lv == 15 else "zra_mf_opt_kernel")
        if os.environ.get("ZRA_MF_FLAGS", "1") in ("0",) and mf_kernel == "zra_mf_dfast_fl_kernel":
            mf_kernel = "zra_mf_dfast_kernel"
        traffic, traffic_src, traffic_stale, ra_traffic, ra_traffic_chain, ra_traffic_mode = None, None, None, None, None, None
        req = None      # the launch's HBM-side REQUESTS against the chip's random-request ceilings (what actually bounds the kernel: DESIGN.md 4, "Round 5")
        tpath = os.path.join(HERE, "profiles", "traffic.json")
        wkey = "L%d_fs%d" % (args.level, fs)
        if os.path.exists(tpath):
            try:
                tall = json.load(open(tpath))
                cands = [v for k, v in tall.items() if isinstance(v, dict) and v.get("workload_key", "L3_fs65536" if k.startswith("bench16g") else None) == wkey and mf_kernel in v]
                if cands:
                    tj = cands[-1]
                    per_frame = tj[mf_kernel]["hbm_bytes_per_launch"] * tj[mf_kernel]["launches"] / tj["frames"]
                    traffic = int(per_frame * nframes / launches)
                    traffic_stale = tj.get("kernel_source_sha") != kernel_source_sha()
                    traffic_src = "rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE (separate passes) of bench.py --steps 1 (%s); kernel sources %s at measurement, %s now" % (
                        tj.get("raw", "profiles/"), tj.get("kernel_source_sha", "unrecorded (commit %s)" % tj.get("measured_at_commit")), kernel_source_sha())
                    e = tj[mf_kernel]
                    if "fetch_kib" in e and "write_kib" in e:
                        scale = nframes / launches / tj["frames"] * e["launches"]
                        fl, wb = e["fetch_kib"] * 1024 / 64 * scale, e["write_kib"] * 1024 * scale
                        req = {"fetch_lines_per_launch": int(fl), "write_bytes_per_launch": int(wb), "read_ceiling_per_s": 48.3e9, "write_ceiling_per_s": 23.0e9,
                               "ceilings_from": "profiles/r02_pmc_calibration.txt (tools/pmc_calib.cpp, 4 GiB region: random 4-byte reads / partial-line writes per second)",
                               "note": "fetches as 64-byte lines; writes between 64 B (full lines: the table clears) and 32 B (partial-line stores) per request: low / high"}
                    # the decode pass: every stage kernel's bytes, the two chain kernels (HBM tables / LDS tables: a frame runs on one of them,
                    # and under --pmc the split differs from pass to pass) summed INSIDE each counter pass — never one kernel's halves
                    # from two different executions (round 5's figure fell below the bytes the stage must move that way)
                    dpass = tj.get("decode_one_pass_of_16GiB", {})
                    if dpass:
                        tot = sum((e.get("fetch_kib", 0) + e.get("write_kib", 0)) for k, e in dpass.items() if k.startswith("zra_dec_"))
                        chn = sum((e.get("fetch_kib", 0) + e.get("write_kib", 0)) for k, e in dpass.items() if k.startswith("zra_dec_chain"))
                        ra_traffic = int(tot * 1024 * nframes / tj["frames"])
                        ra_traffic_chain = int(chn * 1024 * nframes / tj["frames"])
                        ra_traffic_mode = tj.get("ra_mode", "early stop (measured before round 6's whole-frame headline)")
            except Exception:
                traffic = None
        # second roofline object: the random-access leg's dominant kernel (the sequence chains). Algorithmic bytes = compressed bytes
        # of the frames the queries touch + the bytes returned; duration = its HIP-event span per decode pass
        chain_ms = float(np.mean([x["chain_ms"] for x in dec_stage])) if dec_stage else 0.0
        ra_alg = (ra_touched["compressed_bytes"] + q * qb) if ra_touched else None
        ra_ach = (ra_alg / 1e9 / (chain_ms / 1e3)) if (ra_alg and chain_ms > 0) else 0.0
        stage_ms = {k: round(float(np.mean([x[k] for x in dec_stage])), 3) for k in ("parse_ms", "huf_ms", "chain_ms", "exec_ms", "small_ms")} if dec_stage else None
        line = {
            "metric": "compress + RA-decompress GiB/s, 16 GiB @ 64 KiB frames, 1/2/4/8 MI355X",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%.3g GiB/GPU synthetic 'Silesia stand-in' corpus (64 MiB mixed text/binary/skew, tiled), frameSize=%d KiB, level %d, checksum on: "
                                   "CompressBuffer + %d random DecompressRA(offset,%d B) queries per GPU" % (N / GiB, fs >> 10, args.level, q, qb),
                       "frame_size": fs, "level": args.level, "bytes_per_gpu": N, "queries_per_gpu": q, "query_bytes": qb,
                       "parallelism": "frames sharded by index, %d rank(s)" % world, "transport": transport, "compression_ratio": round(ratio, 3),
                       "bit_exact_gate": gate},
            "gather_overlap": ({"mode": "gather on a second communicator's stream beside the serving, one join at the end of the step",
                                "join_wait_ms": round(float(np.mean(getattr(step, "gather_wait_ms", [0.0])[-args.steps:])), 3)} if comm_side is not None else
                               ({"mode": "in sequence"} if world > 1 else None)),
            "compress_gibs": round(N * world / GiB / (np.mean(comp_ms) / 1e3), 3),
            "ra_gibs_returned": round(q * qb * world / GiB / (np.mean(ra_ms) / 1e3), 3),
            "ra_us_per_query": round(np.mean(ra_ms) * 1e3 / q, 3),
            "ra_mode_timed": "whole frames, XXH64 verified (zra.cpp:280-293)" if ra_whole_timed else "early stop",
            "ra_early_stop": ra_early,
            "ra_whole_frames": ra_whole if not ra_whole_timed else {"us_per_query": round(np.mean(ra_ms) * 1e3 / q, 3), "note": "= the timed leg"},
            "ra_whole_frames_us_per_query": (ra_whole["us_per_query"] if ra_whole else None) if not ra_whole_timed else round(np.mean(ra_ms) * 1e3 / q, 3),
            "ra_size_classes": ra_classes,
            "ra_latency": ra_latency,
            "roofline": {"bound": "hbm", "kernel": mf_kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_of_measured_6290": round(achieved / 6290.0, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_stale": traffic_stale, "requests": request_roofline(req, mf_launch_ms),
                         "launch_ms": round(mf_launch_ms, 3), "launches_per_call": launches, "algorithmic_bytes_per_launch": int(alg_per_launch),
                         "other_kernels_launch_ms": {"zra_entropy_kernel": round(ent_launch_ms, 3), "zra_dec_parse+huf+chain+exec (one decode pass)": round(dec_launch_ms, 3)}},
            "roofline_ra": {"bound": "hbm", "kernel": "zra_dec_chain_kernel", "achieved": round(ra_ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ra_ach / HBM_PEAK_GBS, 5),
                            "achieved_whole_pass": round(ra_alg / 1e9 / (dec_launch_ms / 1e3), 2) if (ra_alg and dec_launch_ms > 0) else None,
                            "frac_whole_pass": round(ra_alg / 1e9 / (dec_launch_ms / 1e3) / HBM_PEAK_GBS, 5) if (ra_alg and dec_launch_ms > 0) else None,
                            "whole_pass_ms": round(dec_launch_ms, 3),
                            "traffic": ra_traffic, "traffic_chain_kernels": ra_traffic_chain, "traffic_measured_in_mode": ra_traffic_mode,
                            "traffic_stale": traffic_stale, "launch_ms": round(chain_ms, 3),
                            "algorithmic_bytes_per_launch": ra_alg, "touched": ra_touched, "stage_ms_per_pass": stage_ms,
                            "note": "one decode pass of the touched frames per step; algorithmic = compressed bytes of the touched frames + bytes returned; achieved / frac divide by the chain "
                                    "kernel's span alone (the pass's dominant kernel), achieved_whole_pass / frac_whole_pass by all four kernels of the pass; traffic = PMC bytes of "
                                    "ALL stage kernels of the pass (a query returns 4 KiB of a 64 KiB frame that is decoded whole: the pass writes the frames, not the answers)"},
            "clocks": {"before_timed_region": clocks_before, "after_timed_region": clocks_after},
            # what the waves of the last timed match-finder launch recorded themselves (ZraHipGetLaunchTelemetry): the effective shader
            # clock DURING the launch (cycles against the constant 100 MHz clock), where the waves sat, frames taken per XCD
            "launch": getattr(step, "tele", None),
            "kernel_resources": kernel_resources([mf_kernel, "zra_entropy_kernel", "zra_dec_chain_kernel", "zra_dec_chain_lds_kernel", "zra_dec_exec_kernel"]),
            "c4_share": c4,
            "host_pointer_calls": host_calls,
            "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        sh = getattr(step, "shard", None)
        if sh is not None:
            sh.close()
        if comm_side is not None:
            comm_side.close()
        comm.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
