"""The CPU restatements of the wave-parallel dfast parses (tools/model/) against the oracle, a few seeds each: the kernels' exactness
arguments are fuzzed here on every CPU run of the suite, not only when somebody remembers to run tools/model/run*.sh.
(Bring-up models of two dfast formulations that were built, measured and deleted from the library again — the link parse of round 4 and the
mask parse of round 3, both last present at commit d26905f — and of round 5's bucket-flag statistics; test infrastructure, like the oracle.)"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MODEL = os.path.join(ROOT, "tools", "model")


def _build(src, exe):
    cmd = ["gcc", "-O2", "-std=gnu11", "-Wno-unused-function", "-I" + os.path.join(ROOT, "oracle"), "-o", exe, os.path.join(MODEL, src),
           os.path.join(ROOT, "oracle", "zo_entropy.c"), os.path.join(ROOT, "oracle", "zo_decode.c"), "-lm", "-ldl"]
    subprocess.check_call(cmd)


@pytest.mark.parametrize("src,args", [("dfast_link_model.c", ["1", "60"]), ("dfast_mask_model.c", ["1", "60"])])
def test_lane_array_models_match_the_oracle(tmp_path, src, args):
    exe = str(tmp_path / src.replace(".c", ""))
    _build(src, exe)
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:]
    assert " bad 0" in r.stdout


def test_link_lookup_equals_the_table_lookup(tmp_path):
    """The claim the link formulation rests on: walking a position's bucket chain to the first INSERTED predecessor gives what the hash
    table holds, for every lookup of the serial parse (dfast's insert positions never decrease)."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    exe = str(tmp_path / "dfast_link_stats")
    _build("dfast_link_stats.c", exe)
    data = bench.synth_corpus(4 << 20, 1)
    f = tmp_path / "corpus.bin"
    data.tofile(str(f))
    r = subprocess.run([exe, str(f), "65536", "64", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:]
    assert "lookup mismatches 0" in r.stdout


def test_window_duplicate_detection_in_rounds_is_exact():
    """mf_dfast_lean's duplicate detection (zra_encode_mf.hip, round 4) restated lane by lane: every lane enters its bucket's slot with
    max((round << 6) | (63 - lane)), so the slot's winner is the lowest lane of the round; a lane that finds another lane there compares
    buckets with it — equal: it has an earlier bucket-mate (marked, done); different: next round. The window ends before the first marked
    lane. Claim: that lane is the first lane that shares a bucket (of either table) with an earlier lane — for any slot count, including
    slots shared by many buckets and buckets held by many lanes."""
    import random
    rng = random.Random(20260104)
    for case in range(4000):
        n = rng.randint(2, 64)
        slots = 1 << rng.randint(0, 8)
        nb = rng.choice([2, 5, 17, 64, 300, 5000])                # few buckets: many real duplicates; many buckets: only slots collide
        bL = [rng.randrange(nb) for _ in range(n)]
        bS = [rng.randrange(nb) for _ in range(n)]
        want = next((j for j in range(n) if any(bL[i] == bL[j] or bS[i] == bS[j] for i in range(j))), n)
        dL, dS = [0] * slots, [0] * slots
        pL, pS = [True] * n, [True] * n
        marked, rnd = [False] * n, 0
        while any(pL) or any(pS):
            rnd += 1
            assert rnd <= 64
            for l in range(n):                                       # all lanes' LDS max, then all lanes' reads (one instruction each)
                key = (rnd << 6) | (63 - l)
                if pL[l]: dL[bL[l] % slots] = max(dL[bL[l] % slots], key)
                if pS[l]: dS[bS[l] % slots] = max(dS[bS[l] % slots], key)
            for l in range(n):
                wl = 63 - (dL[bL[l] % slots] & 63) if pL[l] else l
                ws = 63 - (dS[bS[l] % slots] & 63) if pS[l] else l
                lostL, lostS = wl != l, ws != l
                sameL, sameS = lostL and bL[wl] == bL[l], lostS and bS[ws] == bS[l]
                marked[l] = marked[l] or sameL or sameS
                pL[l], pS[l] = lostL and not sameL, lostS and not sameS
        got = next((j for j in range(n) if marked[j]), n)
        assert got == want, (case, n, slots, bL, bS)


def test_four_link_chain_slots_visit_the_reference_candidates():
    """hcw_insert / hcw_search_window (zra_encode_mf.hip, round 4) restated: chain slots hold the next FOUR links of their index's chain,
    captured when the index is inserted (64 indices per step, ahead of the searches; the first link from the hash table or from the
    nearest earlier bucket-mate of the step, the other three from the slot of the index linked to as it is BEFORE the step's stores, or
    from the mate); slots overwritten by indices inserted ahead are kept in a ring of 128. Claim: for every search position the
    candidates visited, in order, are those of the serial walk ZSTD_HcFindBestMatch makes over single links (chain table smaller than
    the input, attempts, lower bound), whatever garbage the untouched slots hold."""
    import random
    rng = random.Random(7)
    for case in range(300):
        N = rng.randint(70, 700)
        clog = rng.randint(7, 9); csize = 1 << clog; cmask = csize - 1
        nb = rng.choice([3, 40, 400])
        attempts0 = rng.choice([1, 2, 3, 4, 5, 8, 32])
        h = [0] + [rng.randrange(nb) for _ in range(N + 70)]          # bucket of index 1..
        # ---- reference: single links, inserted up to the search position only
        ref = {}
        headR, chainR = {}, [rng.randrange(1 << 30) for _ in range(csize)]
        for curr in range(1, N + 1):
            chainR[curr & cmask] = headR.get(h[curr], 0); headR[h[curr]] = curr
            minChain = curr - csize if curr > csize else 0
            out, mi, att = [], chainR[curr & cmask], attempts0
            while mi >= 1 and att > 0:
                out.append(mi)
                if mi <= minChain: break
                mi = chainR[mi & cmask]; att -= 1
            ref[curr] = out
        # ---- four links per slot, windows of 64 positions inserted ahead (indices <= last position of the window + 1)
        GARB = lambda: tuple(rng.randrange(1 << 30) for _ in range(4))
        head, chain, ring = {}, [GARB() for _ in range(csize)], [GARB() for _ in range(128)]
        insEnd = 1
        w = 1
        while w <= N:
            endIdx = min(w + 63, N) + 2                                  # (hcw_search_window: lastPos + 2)
            while insEnd < endIdx:
                lanes = [i for i in range(insEnd, min(endIdx, insEnd + 64))]
                link, frm = {}, {}
                for i in lanes:
                    mates = [j for j in lanes if j < i and h[j] == h[i]]
                    frm[i] = mates[-1] if mates else None
                    link[i] = mates[-1] if mates else head.get(h[i], 0)
                E = {i: (chain[link[i] & cmask] if (frm[i] is None and link[i]) else (0, 0, 0, 0)) for i in lanes}   # before the stores
                l2 = {i: (link[frm[i]] if frm[i] is not None else E[i][0]) for i in lanes}
                l3 = {i: (l2[frm[i]] if frm[i] is not None else E[i][1]) for i in lanes}
                l4 = {i: (l3[frm[i]] if frm[i] is not None else E[i][2]) for i in lanes}
                old = {i: chain[i & cmask] for i in lanes}
                for i in lanes:
                    ring[i & 127] = old[i] if i > csize else (0, 0, 0, 0)
                    chain[i & cmask] = (link[i], l2[i], l3[i], l4[i])
                for i in lanes:
                    if not any(j > i and h[j] == h[i] for j in lanes): head[h[i]] = i
                insEnd = lanes[-1] + 1
            for curr in range(w, min(w + 63, N) + 1):
                minChain = curr - csize if curr > csize else 0
                e, att, out = chain[curr & cmask], attempts0, []
                while e[0] >= 1 and att > 0:
                    t = [True, False, False, False]
                    t[1] = e[0] > minChain and att > 1 and e[1] >= 1
                    t[2] = t[1] and e[1] > minChain and att > 2 and e[2] >= 1
                    t[3] = t[2] and e[2] > minChain and att > 3 and e[3] >= 1
                    en = None
                    if t[3]:
                        over = e[3] + csize
                        en = ring[over & 127] if (curr < over < insEnd) else chain[e[3] & cmask]
                    stop = False
                    for k in range(4):
                        if not t[k]: stop = True; break
                        out.append(e[k])
                        if e[k] <= minChain: stop = True; break
                        att -= 1
                    if stop: break
                    e = en
                assert out == ref[curr], (case, curr, out, ref[curr])
            w += 64
