"""Builds libzra_amd.so (HIP kernels + host engine + C ABI) for gfx950, in-tree, with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzra_amd.so")
SOURCES = ["zra_decode.hip", "zra_encode_mf.hip", "zra_encode_ent.hip", "zra_encode.hip", "zra_engine.hip", "zra_hostpipe.hip", "zra_comm.hip", "zra_capi.cpp"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "tools", "zratool_amd.cpp")] + [os.path.join(HERE, "..", "include", f) for f in ("zra.h", "zra.hpp", "zra_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


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
               "-I" + os.path.join(HERE, "..", "include"), "-I" + CSRC, "-c", os.path.join(CSRC, s), "-o", o]
        cmd[3:3] = os.environ.get("ZRA_EXTRA_CFLAGS", "").split()      # bring-up only, e.g. -DZRA_MF_PROFILE
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on " + s)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl"]   # RCCL: zra_comm.hip
    subprocess.check_call(link)
    # command-line counterpart of the reference's zratool (C++ API consumer)
    tool = os.path.join(HERE, "tools", "zratool_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(HERE, "..", "include"), os.path.join(HERE, "tools", "zratool_amd.cpp"),
                           "-o", tool, "-L" + HERE, "-lzra_amd", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
