"""Generates the golden fixtures under tests/golden/ (run in the dev container; needs /opt/conda libzstd 1.4.9).

Provenance of each fixture:
  g1_*.zra, A_zeros.*.zra : bytes produced by the REFERENCE itself (source/zra.cpp built during the survey with zlib standing in
                            for CRCpp). Small enough to commit verbatim. They are data, not source.
  anchors.json            : size + sha256[:16] of full reference archives for the seeded generators A-E.
  frames.json             : zstd frames produced by the real dependency libzstd 1.4.9 (through oracle/zo_zra.c's dlopen backend) for
                            decoder coverage (levels -5..22, every block/literal/sequence mode seen in the census), with the sha256
                            of the input each must regenerate. Inputs come from tests/corpus.py generators.
  errors.json             : (mutation -> ZraStatus) table produced by the libzstd-backed container path.
"""
import base64, hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O, corpus as C

def main():
    assert O.have_libzstd() and O.lib().zo_libzstd_version() == b"1.4.9"
    # --- G1 known answers, typed from the survey's hex dump
    g1 = bytes.fromhex(
        "502a4d18" "32000000" "5a524130" "0100" "e745b1fd" "0a00000000000000" "04000000" "04000000" "00000000"
        "0000000000" "1100000000" "2200000000" "3100000000"
        "28b52ffd0400210000" "61626364" "cc925dd2"
        "28b52ffd0400210000" "65666768" "c9672e04"
        "28b52ffd0400110000" "696a" "74b59aa0")
    assert len(g1) == 107
    open(os.path.join(HERE, "g1_abcdefghij_fs4.zra"), "wb").write(g1)
    g1e = bytes.fromhex("502a4d18" "23000000" "5a524130" "0100" "924940d5" "0000000000000000" "01000000" "00000100" "00000000" "0000000000")
    assert len(g1e) == 43
    open(os.path.join(HERE, "g1_empty_fs65536.zra"), "wb").write(g1e)
    for name in ("A_zeros.l3.zra", "A_zeros.l9.zra"):
        src = os.path.join("/tmp/probe", name)
        if os.path.exists(src):
            open(os.path.join(HERE, name), "wb").write(open(src, "rb").read())
    anchors = {"A": [[491, "43d185c2f2695ad5"], [167, "818e5bd72bfe30d9"], [1771, "d7f8b0dab0c531a1"]],
               "B": [[1048907, "c177037d39524915"], [1048703, "3a67b78debf7a9f7"], [1049771, "25e3cc281e26d015"]],
               "C": [[121043, "f0d6c811150a5b97"], [88486, "b55eb36ca4a9b387"], [121209, "5123513a5edae10c"]],
               "D": [[617701, "22d9ea2b66d6d77f"], [649303, "10776bb9219cc67a"], [643463, "a5ef8b564a48c079"]],
               "E": [[300420, "4db93f63fb79a723"], [286381, "87cead4cc3a0372a"], [303696, "2292777534fac9dd"]]}
    inputs = {"A": "30e14955ebf13522", "B": "8f2a39d78dc184d2", "C": "24178e52320aca34", "D": "75fe1fd604a6778f", "E": "3f19bdfc9d8ce921"}
    json.dump({"configs": [[3, 65536], [9, 262144], [3, 16384]], "archives": anchors, "inputs_sha256_16": inputs, "n": 1 << 20},
              open(os.path.join(HERE, "anchors.json"), "w"), indent=1)
    # --- decoder coverage frames from libzstd 1.4.9
    gens = {"C": C.gen_C(1 << 20), "D": C.gen_D(1 << 20), "E": C.gen_E(1 << 20), "F": C.gen_struct(1 << 19), "G": C.gen_alpha4(1 << 18),
            "H": C.gen_litrle(1 << 19), "L": C.gen_loglike(1 << 20), "A": C.gen_A(1 << 18), "B": C.gen_B(1 << 18)}
    frames = []
    picks = [("C", 0, 65536, 3), ("C", 100000, 16384, 3), ("C", 5, 300, 3), ("C", 77, 2561, 3), ("E", 690000, 65536, 3), ("D", 0, 20000, 3),
             ("F", 1000, 65536, 3), ("G", 0, 30000, 3), ("L", 0, 65536, 1), ("L", 70000, 65536, 5), ("L", 0, 40000, 9), ("C", 0, 30000, 16),
             ("L", 0, 30000, 19), ("F", 0, 40000, 22), ("L", 0, 20000, -5), ("H", 0, 262144, 9), ("H", 0, 262144, 3), ("F", 0, 262144, 6),
             ("A", 0, 65536, 3), ("A", 0, 200000, 9), ("B", 0, 5000, 3), ("E", 699000, 4000, 3), ("C", 0, 6, 3), ("C", 0, 1, 3), ("F", 300000, 140000, 4)]
    for g, off, n, lvl in picks:
        data = gens[g][off:off + n]
        for ck in (True, False) if n in (300, 20000) else (True,):
            fr = O.compress_frame(data, lvl, ck, "zl")
            frames.append({"gen": g, "off": off, "n": n, "level": lvl, "checksum": ck, "sha256": hashlib.sha256(data).hexdigest(),
                           "frame_b64": base64.b64encode(fr).decode()})
    json.dump(frames, open(os.path.join(HERE, "frames.json"), "w"))
    # --- G5 error table from the libzstd-backed container path
    data = C.gen_E(1 << 20)[650000:650000 + 300000]
    st, arc = O.zra_compress(data, 3, 65536, True, 0, "zl")
    hs = int.from_bytes(arc[4:8], "little") + 8
    muts = []
    def add(name, pos=None, xor=None, trunc=None, setbytes=None):
        muts.append({"name": name, "pos": pos, "xor": xor, "trunc": trunc, "set": setbytes})
    add("identity")
    add("checksum_flip", pos=len(arc) - 1, xor=0x5A)
    for k, p in enumerate((hs + 0, hs + 4, hs + 5, hs + 6, hs + 9, hs + 14, hs + 40, hs + 400, hs + 3000, hs + 20000, len(arc) - 200)):
        add("body_flip_%d" % k, pos=p, xor=0x10)
    add("bad_zra_magic", pos=8, xor=0xFF)
    add("version_2", setbytes=[12, 2, 0])
    add("version_0", setbytes=[12, 0, 0])
    add("crc_flip", pos=15, xor=0xFF)
    add("seektable_flip", pos=38 + 6, xor=0x01)
    add("truncate_100", trunc=len(arc) - 100)
    add("truncate_mid", trunc=hs + 30000)
    table = []
    for m in muts:
        a = bytearray(arc)
        if m["pos"] is not None: a[m["pos"]] ^= m["xor"]
        if m["set"]: a[m["set"][0]:m["set"][0] + 2] = bytes(m["set"][1:])
        if m["trunc"] is not None: a = a[:m["trunc"]]
        stf, _ = O.zra_decompress(bytes(a), len(data), "zl")
        table.append({"mutation": m, "full": list(stf)})
    ra = []
    for off, sz in ((len(data) - 10, 10), (len(data) - 11, 10), (0, 1), (65535, 2), (65536, 65536), (100, 200000), (len(data), 0), (len(data) - 1, 0)):
        s, _ = O.zra_ra(arc, off, sz, "zl")
        ra.append({"offset": off, "size": sz, "status": list(s)})
    json.dump({"input": {"gen": "E", "off": 650000, "n": 300000}, "level": 3, "frame_size": 65536, "archive_sha256": hashlib.sha256(arc).hexdigest(),
               "mutations": table, "ra_bounds": ra}, open(os.path.join(HERE, "errors.json"), "w"), indent=1)
    print("golden fixtures written:", sorted(os.listdir(HERE)))

if __name__ == "__main__":
    main()
