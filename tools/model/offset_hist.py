"""CPU (round 6): how far back do the matches of the bench corpus reach? The question behind "a history ring behind the parse in LDS": the
share of a frame's match-side compare loads whose source lies within R bytes of the parse position = the share of sequences with
offset <= R (level 3, 64 KiB frames, the oracle's sequences with repeat codes resolved). usage: python tools/model/offset_hist.py [frames]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
import numpy as np, oracle_lib as O, bench
base = bench.synth_corpus(64 << 20, 1); fs = 65536
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(0); dist = []
for f in rng.randint(0, (64 << 20) // fs, size=nfr):
    rep = [1, 4, 8]
    for ll, ml, ov in O.sequences(base[f * fs:(f + 1) * fs].tobytes(), 3):
        if ov > 3: off = ov - 3; rep = [off, rep[0], rep[1]]
        else:
            idx = ov - 1 + (1 if ll == 0 else 0)
            if idx == 0: off = rep[0]
            elif idx == 1: off = rep[1]; rep = [off, rep[0], rep[2]]
            elif idx == 2: off = rep[2]; rep = [off, rep[0], rep[1]]
            else: off = max(rep[0] - 1, 1); rep = [off, rep[0], rep[1]]
        dist.append(off)
dist = np.array(dist)
print("sequences per frame %.0f" % (len(dist) / nfr))
for r in (64, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
    print("offset <= %5d: %.1f %%" % (r, 100 * (dist <= r).mean()))
