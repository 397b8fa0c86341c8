"""End to end against the UNMODIFIED reference binary run beside it on the GPU box's host cores (oracle/_ref/mecat2pw,
built in the build container from /root/reference by oracle/Makefile and shipped with the snapshot): same FASTA in,
same multiset of output lines out, for every task x technology combination, at a size beyond the committed goldens."""
import os
import subprocess

import pytest

import helpers as H

pytestmark = pytest.mark.gpu
BIN = os.path.join(H.ROOT, "mecat_amd", "bin", "mecat2pw")
REF = os.path.join(H.ROOT, "oracle", "_ref", "mecat2pw")


def _lines(binary, tmp_path, tag, fa, args):
    out = str(tmp_path / (tag + ".out"))
    wrk = tmp_path / ("w_" + tag)
    wrk.mkdir()
    r = subprocess.run([binary, "-d", fa, "-o", out, "-w", str(wrk), "-t", "16"] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return sorted(open(out).read().splitlines())


@pytest.mark.skipif(not os.path.exists(REF), reason="reference binary not built (oracle/_ref)")
@pytest.mark.parametrize("tech,nreads,L,err,genome,seed", [(0, 2500, 7000, 0.15, 900000, 91), (1, 1200, 6000, 0.12, 400000, 92)])
def test_same_output_as_reference_binary(tmp_path, tech, nreads, L, err, genome, seed):
    codes, lens = H.synth_reads(nreads, L, err, genome, seed, tech)
    fa = str(tmp_path / "reads.fa")
    H.write_fasta(fa, codes, lens)
    for args in (["-j", "0"], ["-j", "1"], ["-j", "1", "-g", "1", "-n", "30"]):
        a = args + ["-x", str(tech)]
        tag = "".join(x.strip("-") for x in a)
        want = _lines(REF, tmp_path, "ref" + tag, fa, a)
        got = _lines(BIN, tmp_path, "hip" + tag, fa, a)
        assert len(want) > 1000
        assert got == want, "%s: %d vs %d lines" % (a, len(got), len(want))


def test_n1500_on_the_dense_set(tmp_path):
    """`-n 1500 -k 2` on helpers.dense_reads(): 17 query reads keep 1 000 candidates or more, 14 keep 1 000 overlaps or more — the
    reference's per-read m4 sort (pw_impl.cpp:581) then works on lists beyond libstdc++ parallel mode's 1 000-element threshold
    (oracle/_ref is built with the reference's own release flags, oracle/Makefile).  Pins: the sorted outputs of
    the unmodified reference (tests/golden/dense.json, make_golden_dense.py: six minutes of 32 host threads, not repeated here); the
    candidate stage is also run side by side where the reference binary travelled."""
    import json
    g = json.load(open(os.path.join(H.GOLDEN, "dense.json")))
    codes, lens = H.dense_reads()
    fa = str(tmp_path / "dense.fa")
    H.write_fasta(fa, codes, lens)
    assert H.sha256_lines(open(fa).read().splitlines()) == g["fasta_sha256"]
    for name in ("can", "m4_g1"):
        got = _lines(BIN, tmp_path, "hip_" + name, fa, g[name]["args"])
        assert len(got) == g[name]["lines"], name
        assert H.sha256_lines(got) == g[name]["sorted_sha256"], name
        assert g[name]["query_reads_with_1000_lines_or_more"] >= 10
    if os.path.exists(REF):
        assert _lines(REF, tmp_path, "ref_can", fa, g["can"]["args"]) == _lines(BIN, tmp_path, "hip_can2", fa, g["can"]["args"])
