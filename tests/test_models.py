"""The CPU restatements of the wave-parallel dfast parses (tools/model/) against the oracle, a few seeds each: the kernels' exactness
arguments are fuzzed here on every CPU run of the suite, not only when somebody remembers to run tools/model/run*.sh.
(Bring-up models of zra_amd/csrc/zra_encode_lk.hip and of mf_dfast_mask in zra_encode_mf.hip — test infrastructure, like the oracle.)"""
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
