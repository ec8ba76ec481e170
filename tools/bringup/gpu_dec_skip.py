"""bring-up: execute-kernel stage ablation (ZRA_DEC_SKIP bits: 1 short literal runs, 2 short match copies, 4 round fences, 8 long literal runs); output is wrong by design."""
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
    try: eng.decompress(d_arc.data_ptr(), asz, d_out.data_ptr(), n)
    except Exception as e: pass
    st = eng.kernel_stats()
print("SKIP", os.environ.get("ZRA_DEC_SKIP"), "decode 2 GiB kernels %%.1f ms" %% st["dec_ms"])
''' % (here, root)
for spec in sys.argv[1:]:
    env = dict(os.environ); env["ZRA_DEC_SKIP"] = spec
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
