import sys, os, subprocess
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); root = os.path.dirname(here)
code = r'''
import sys, os, time
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch, zra_amd as Z, bench
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1); fs = 65536; n = 3 << 30
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
for i in range(2):
    torch.cuda.synchronize(); t = time.time(); asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True); dt = time.time() - t
st = eng.kernel_stats()
print("LDS", os.environ.get("ZRA_MF_LDS"), "FILTER", os.environ.get("ZRA_MF_FILTER"), "compress 3GiB: %%.1f ms  mf %%.1f ms/launch ent %%.1f ms/launch" %% (dt * 1e3, st["mf_ms"]/st["mf_launches"], st["ent_ms"]/st["ent_launches"]))
''' % (here, root)
for spec in (sys.argv[1:] or ["0"]):             # each argument: extra LDS bytes, or NAME=VALUE[;NAME=VALUE]
    env = dict(os.environ)
    if "=" in spec:
        for kv in spec.split(";"): env[kv.split("=")[0]] = kv.split("=")[1]
    else: env["ZRA_MF_LDS"] = spec
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
