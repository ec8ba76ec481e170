"""bring-up: one archive (n bytes, frame size fs, level) through DecompressBuffer"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, oracle_lib as O, corpus as C, zra_amd as Z
n, fs, level = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.RandomState(1)
d = C.random_lz_input(rng, n)
st, arc = O.zra_compress(d, level, fs, True, 0, "zl")
print("n", n, "fs", fs, "level", level, "arc", len(arc), flush=True)
assert Z.DecompressBuffer(arc) == d
print("ok", flush=True)
