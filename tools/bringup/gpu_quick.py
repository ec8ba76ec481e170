"""Ad-hoc GPU check used during bring-up (not a pytest file): HIP path vs oracle on a few inputs."""
import sys, os, time, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle_lib as O, corpus as C
import zra_amd as Z

def main():
    print("devices", Z.load().ZraHipDeviceCount())
    data = {"C": C.gen_C(1 << 20), "E": C.gen_E(1 << 20), "D": C.gen_D(1 << 19), "B": C.gen_B(1 << 18), "A": C.gen_A(1 << 18), "F": C.gen_struct(1 << 19)}
    bad = 0
    for name, d in data.items():
        for lvl, fs in ((3, 65536), (3, 16384), (1, 65536), (3, 262144), (5, 65536), (9, 262144)):
            st, ref = O.zra_compress(d, lvl, fs, True, 0, "zo")
            # decode of the oracle archive on the GPU
            t = time.time()
            try:
                out = Z.DecompressBuffer(ref)
                okd = out == d
            except Exception as e:
                okd = False; print("  decode exc", e)
            td = time.time() - t
            t = time.time()
            try:
                arc = Z.CompressBuffer(d, lvl, fs, True)
                okc = arc == ref
                if not okc:
                    # locate first differing frame
                    n = min(len(arc), len(ref)); i = next((k for k in range(n) if arc[k] != ref[k]), n)
                    print("  first diff at byte", i, "sizes", len(arc), len(ref))
            except Exception as e:
                okc = False; print("  compress exc", e)
            tc = time.time() - t
            print(name, lvl, fs, "decode", "OK" if okd else "FAIL", "%.3fs" % td, "compress", "OK" if okc else "FAIL", "%.3fs" % tc, flush=True)
            bad += (not okd) + (not okc)
    # RA
    d = data["E"]; st, ref = O.zra_compress(d, 3, 65536, True, 0, "zo")
    for off, sz in ((0, 10), (65530, 20), (100000, 300000), (5, 65531), (len(d) - 11, 10)):
        try:
            r = Z.DecompressRA(ref, off, sz); ok = r == d[off:off + sz]
        except Exception as e:
            ok = False; print("  ra exc", e)
        print("RA", off, sz, "OK" if ok else "FAIL"); bad += not ok
    print("TOTAL BAD", bad)
    return bad

if __name__ == "__main__":
    sys.exit(1 if main() else 0)
