"""bring-up: the link dfast kernels (zra_encode_lk.hip) against the oracle on a spread of inputs, then their speed on the bench corpus.
usage: gpu_lk_check.py [speed GiB]   (environment: ZRA_MF_LK, ZRA_LK_MODE, ZRA_LK_GROUP, ZRA_LK_PP_CUS)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle_lib as O, corpus as C
import zra_amd as Z


def first_diff(eng, d, level, fs):
    for f in range((len(d) + fs - 1) // fs):
        seqs, (nb, last_ll, skip) = eng.debug_read_seqs(f)
        ref = O.sequences(d[f * fs:(f + 1) * fs], level)
        mine = seqs + [(last_ll, 0, 0)]
        if mine != ref and not skip:
            k = next((i for i in range(min(len(mine), len(ref))) if mine[i] != ref[i]), min(len(mine), len(ref)))
            pos = sum(a + b for a, b, _ in ref[:k])
            print("    frame", f, "nb", nb, "ref", len(ref) - 1, "first diff at seq", k, "pos", pos, "mine", mine[max(0, k - 1):k + 2], "ref", ref[max(0, k - 1):k + 2])
            return


def main():
    speed_gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    eng = Z.Engine(0); dev = torch.device("cuda", 0)
    import bench
    corp = bench.synth_corpus(8 << 20, 1).tobytes()
    data = {"bench": corp[:3 << 20], "C": C.gen_C(1 << 20), "E": C.gen_E(1 << 20), "D": C.gen_D(1 << 19), "B": C.gen_B(1 << 18), "A": C.gen_A(1 << 18),
            "F": C.gen_struct(1 << 19), "log": C.gen_loglike(1 << 20), "a4": C.gen_alpha4(1 << 19), "rle": C.gen_litrle(262144)}
    bad = 0
    for name, d in data.items():
        for lvl, fs in ((3, 65536), (4, 65536), (3, 16384), (3, 4096), (4, 40000), (3, 65535), (3, 1000), (4, 300)):
            for cut in (0, 12345):
                dd = d[:len(d) - cut] if cut else d
                t = torch.from_numpy(np.frombuffer(dd, dtype=np.uint8).copy()).to(dev)
                out = torch.empty(Z.GetOutputBufferSize(len(dd), fs) + 64, dtype=torch.uint8, device=dev)
                try:
                    n = eng.compress(t.data_ptr(), len(dd), out.data_ptr(), lvl, fs, True)
                    st, ref = O.zra_compress(dd, lvl, fs, True)
                    ok = bytes(out[:n].cpu().numpy()) == ref
                except Exception as e:
                    ok = False; print("  exc", e)
                if not ok:
                    bad += 1
                    print(name, lvl, fs, cut, "FAIL", flush=True)
                    first_diff(eng, dd, lvl, fs)
        print(name, "done, bad so far", bad, flush=True)
    print("TOTAL BAD", bad, flush=True)
    if speed_gib > 0:
        N = int(speed_gib * (1 << 30))
        c64 = bench.synth_corpus(64 << 20, 1)
        t = torch.from_numpy(np.resize(c64, N)).to(dev)
        out = torch.empty(Z.GetOutputBufferSize(N, 65536) + 64, dtype=torch.uint8, device=dev)
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            n = eng.compress(t.data_ptr(), N, out.data_ptr(), 3, 65536, True)
            torch.cuda.synchronize(); dt = time.time() - t0
            print("compress %.2f GiB: %.1f ms = %.2f GiB/s, ratio %.3f, kernel stats %s" % (speed_gib, dt * 1e3, speed_gib / dt, N / n, eng.kernel_stats()), flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
