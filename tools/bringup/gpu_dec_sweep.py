import sys, os, subprocess
here = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"); root = os.path.dirname(here)
code = r'''
import sys, os, time
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch, zra_amd as Z, bench
dev = torch.device("cuda", 0); eng = Z.Engine(0)
base = bench.synth_corpus(64 << 20, 1); fs = 65536; n = 2 << 30
d_in = torch.from_numpy(np.resize(base, n)).to(dev)
d_arc = torch.empty(Z.GetOutputBufferSize(n, fs) + 64, dtype=torch.uint8, device=dev)
asz = eng.compress(d_in.data_ptr(), n, d_arc.data_ptr(), 3, fs, True)
d_out = torch.empty(n, dtype=torch.uint8, device=dev)
for i in range(3):
    eng.decompress(d_arc.data_ptr(), asz, d_out.data_ptr(), n)
    st = eng.kernel_stats()
print("DEC_WAVES", os.environ.get("ZRA_DEC_WAVES"), "decode kernel 2 GiB: %%.1f ms" %% st["dec_ms"])
''' % (here, root)
for w in (sys.argv[1:] or ["0"]):
    env = dict(os.environ, ZRA_DEC_WAVES=w)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
