"""bring-up helper: find frames of the bench corpus the GPU decoder rejects and print their anatomy."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, oracle_lib as O, zra_amd as Z, bench

def anatomy(f):
    p = 6; out = []
    while True:
        bh = f[p] | f[p+1] << 8 | f[p+2] << 16; p += 3
        bt = (bh >> 1) & 3; bs = bh >> 3
        d = {"type": bt, "size": bs}
        if bt == 2:
            q = p; lt = f[q] & 3; sf = (f[q] >> 2) & 3
            if lt < 2:
                if sf in (0, 2): rs = f[q] >> 3; lh = 1
                elif sf == 1: rs = (f[q] | f[q+1] << 8) >> 4; lh = 2
                else: rs = (f[q] | f[q+1] << 8 | f[q+2] << 16) >> 4; lh = 3
                cs = rs if lt == 0 else 1
            else:
                if sf in (0, 1): v = f[q] | f[q+1] << 8 | f[q+2] << 16; rs = (v >> 4) & 0x3ff; cs = v >> 14; lh = 3
                elif sf == 2: v = int.from_bytes(f[q:q+4], 'little'); rs = (v >> 4) & 0x3fff; cs = v >> 18; lh = 4
                else: v = int.from_bytes(f[q:q+5], 'little'); rs = (v >> 4) & 0x3ffff; cs = v >> 22; lh = 5
            d.update(lit=lt, sf=sf, regen=rs, comp=cs)
            if lt == 2: d["hufhdr"] = f[q + lh]
            q += lh + cs
            n = f[q]
            if n == 0: ns = 0; q += 1
            elif n < 128: ns = n; q += 1
            elif n < 255: ns = ((n - 128) << 8) + f[q+1]; q += 2
            else: ns = f[q+1] + (f[q+2] << 8) + 0x7f00; q += 3
            d["nbSeq"] = ns
            if ns: d["modes"] = (f[q] >> 6, (f[q] >> 4) & 3, (f[q] >> 2) & 3)
        out.append(d)
        p += 1 if bt == 1 else bs
        if bh & 1: break
    return out

base = bench.synth_corpus(64 << 20, 1)
fs = 65536
data = base[: 32 << 20].tobytes()
st, arc = O.zra_compress(data, 3, fs, True, 0, "zl" if O.have_libzstd() else "zo")
hs = int.from_bytes(arc[4:8], 'little') + 8; ts = int.from_bytes(arc[26:30], 'little')
tab = [int.from_bytes(arc[38+5*i:43+5*i], 'little') for i in range(ts)]
bad = 0
for k in range(ts - 1):
    fr = arc[hs + tab[k]: hs + tab[k+1]]
    one = Z.stitch_header([len(fr)], fs, fs) + fr
    try:
        ok = Z.DecompressBuffer(one) == data[k*fs:(k+1)*fs]; err = None
    except Exception as e:
        ok = False; err = str(e)
    if not ok:
        bad += 1
        if bad <= 6: print("frame", k, "len", len(fr), err, anatomy(fr), flush=True)
print("frames", ts - 1, "bad", bad)
