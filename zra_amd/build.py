"""Builds libzra_amd.so (HIP kernels + host engine + C ABI) for gfx950, in-tree, with hipcc."""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzra_amd.so")
RESOURCES = os.path.join(HERE, "build", "kernel_resources.json")     # per kernel: VGPRs, scratch, spills, occupancy (from the compiler's remarks)
SOURCES = ["zra_decode.hip", "zra_encode_mf.hip", "zra_encode_ent.hip", "zra_encode.hip", "zra_engine.hip", "zra_hostpipe.hip", "zra_comm.hip", "zra_capi.cpp"]


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(RESOURCES):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "tools", "zratool_amd.cpp")] + [os.path.join(HERE, "..", "include", f) for f in ("zra.h", "zra.hpp", "zra_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


_REMARK = re.compile(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]):\s*(\S+)")
_KEYS = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes", "Occupancy [waves/SIMD]": "waves_per_simd",
         "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds_bytes"}


def parse_resource_remarks(text, source):
    """-Rpass-analysis=kernel-resource-usage -> {kernel: {vgprs, scratch_bytes, ...}} (what tests/test_kernel_budgets.py checks)"""
    res, cur = {}, None
    for m in _REMARK.finditer(text):
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = res.setdefault(v, {"source": source})
        elif cur is not None:
            cur[_KEYS[k]] = int(v)
    return res


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        o = os.path.join(HERE, "build", s + ".o")
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-x", "hip",
               "-Rpass-analysis=kernel-resource-usage",
               "-I" + os.path.join(HERE, "..", "include"), "-I" + CSRC, "-c", os.path.join(CSRC, s), "-o", o]
        cmd[3:3] = os.environ.get("ZRA_EXTRA_CFLAGS", "").split()      # bring-up only, e.g. -DZRA_MF_PROFILE
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    resources = {}
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on " + s)
        resources.update(parse_resource_remarks(out.decode(errors="replace"), s))
    with open(RESOURCES, "w") as f:
        json.dump(resources, f, indent=1, sort_keys=True)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl"]   # RCCL: zra_comm.hip
    subprocess.check_call(link)
    # command-line counterpart of the reference's zratool (C++ API consumer)
    tool = os.path.join(HERE, "tools", "zratool_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(HERE, "..", "include"), os.path.join(HERE, "tools", "zratool_amd.cpp"),
                           "-o", tool, "-L" + HERE, "-lzra_amd", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
