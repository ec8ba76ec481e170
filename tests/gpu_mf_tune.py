import sys, os, subprocess
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(here)
code = open(os.path.join(here, "gpu_mf_sweep.py")).read().split("code = r'''")[1].split("''' % (here, root)")[0] % (here, root)
for tune in [int(x) for x in sys.argv[1:]]:
    env = dict(os.environ, ZRA_MF_TUNE=str(tune))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print("TUNE", tune, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
