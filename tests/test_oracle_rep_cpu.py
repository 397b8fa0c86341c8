"""Re-pins the oracle on REPEAT-STRUCTURED read sets (interspersed families at 0 - 5 % divergence, microsatellites, homopolymer runs:
helpers.REP_SETS) against the UNMODIFIED reference compiled into oracle/_ref — index, candidate lists function by function, and the
sorted `.can` / `.m4` output of the reference binary.  These inputs fill k-mer buckets up to and beyond the cap of 128
(lookup_table.cpp:97), run the 41st-seed rule on hits that are not self hits (pw_impl.cpp:121-159) and tie scores in the top-MAXC
list (pw_impl.cpp:442-455); the uniform genomes of the other sets do none of that ("What's weak").
Skipped where oracle/_ref is absent; tests/golden/rep.json (make_golden_rep.py) carries the same pins as hashes for the GPU box."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.skipif(not H.ref_available(), reason="oracle/_ref not built")


@pytest.fixture(scope="module", params=sorted(H.REP_SETS))
def rep(request, tmp_path_factory):
    name = request.param
    codes, lens, ont, st = H.rep_set(name)
    d = str(tmp_path_factory.mktemp(name))
    fa = os.path.join(d, name + ".fa")
    H.write_fasta(fa, codes, lens)
    wrk = os.path.join(d, "wrk")
    os.makedirs(wrk)
    R = H.ref()
    assert R.refh_split(fa.encode(), wrk.encode()) == 1
    rv = R.refh_load_volume(os.path.join(wrk, "vol0").encode())
    ridx = R.refh_build_index(rv, 1)
    ov = H.orc_pack(codes, lens)
    oidx = H.orc().orc_index_build(ov)
    return dict(name=name, codes=codes, lens=lens, ont=ont, st=st, dir=d, fa=fa, rv=rv, ridx=ridx, ov=ov, oidx=oidx)


def test_the_sets_do_what_they_are_for(rep):
    b = H.bucket_stats(rep["codes"], rep["lens"])
    assert b["dropped"] > 50 and b["near_cap"] > 5, b
    H.orc_stats_reset()
    H.orc_seed_all(rep["ov"], rep["ov"], rep["oidx"], H.orc_params(tech=rep["ont"]))
    s = H.orc_stats()
    assert s["non_self"] > 100 and s["non_self_dropped"] > 10, s


def test_index_identical_on_repeats(rep):
    R = H.ref()
    counts = np.empty(H.NK, dtype=np.int32)
    n = R.refh_index_dump(rep["ridx"], counts.ctypes.data, None)
    offs = np.empty(n, dtype=np.int32)
    R.refh_index_dump(rep["ridx"], counts.ctypes.data, offs.ctypes.data)
    oi = rep["oidx"].contents
    assert oi.num_kmers == n
    assert np.array_equal(np.ctypeslib.as_array(oi.counts, shape=(H.NK,)), counts)
    assert np.array_equal(np.ctypeslib.as_array(oi.offsets, shape=(n,)), offs)


@pytest.mark.parametrize("maxc", [100, 4])
def test_candidates_identical_on_repeats(rep, maxc):
    R = H.ref()
    tech = rep["ont"]
    p = H.orc_params(tech=tech, maxc=maxc)
    R.refh_set_params(maxc, p.min_align_size, p.min_kmer_match, tech)
    ours = H.orc_seed_all(rep["ov"], rep["ov"], rep["oidx"], p)
    out = np.zeros((maxc, 12), dtype=np.int32)
    tot = ties = 0
    for rid in range(len(rep["lens"])):
        k = R.refh_seed_read(rep["rv"], rep["rv"], rep["ridx"], rid, 0, out.ctypes.data)
        a = ours[rid]
        assert k == len(a), rid
        got = np.stack([a[n] for n in H.CAND_DTYPE.names], axis=1) if k else np.zeros((0, 12), np.int32)
        assert np.array_equal(got, out[:k]), rid
        tot += k
        ties += int(k - len(np.unique(out[:k, 6])))
    assert tot > 1000 and (maxc < 100 or ties > 100)       # tied scores inside the kept lists


def _run_ref(rep, args, name):
    out = os.path.join(rep["dir"], name)
    wrk = os.path.join(rep["dir"], "w_" + name)
    subprocess.run([H.ref_bin(), "-d", rep["fa"], "-o", out, "-w", wrk, "-t", "16"] + args, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(open(out).read().splitlines())


def test_can_and_m4_output_identical_on_repeats(rep):
    tech = rep["ont"]
    O = H.orc()
    p = H.orc_params(tech=tech)
    offs, _ = H.vol_arrays(rep["ov"])
    cands = H.orc_seed_all(rep["ov"], rep["ov"], rep["oidx"], p)
    want = _run_ref(rep, ["-j", "0", "-x", str(tech)], "ref.can")
    assert sorted(H.can_lines_from_cands(cands, offs, offs)) == want
    g = json.load(open(os.path.join(H.GOLDEN, "rep.json")))[rep["name"]]
    if tech:      # (95 s of one reference thread: its lines for the first 40 query reads are a fixture, tests/golden/make_golden_rep.py)
        want = open(os.path.join(H.GOLDEN, "rep_ont.m4_g1.q40.sorted")).read().splitlines()
    else:
        want = _run_ref(rep, ["-j", "1", "-g", "1", "-x", str(tech)], "ref.m4")
        assert H.sha256_lines(want) == g["m4_g1"]["sorted_sha256"] and len(want) == g["m4_g1"]["lines"]      # the committed pin is this output
    bk = O.orc_bk_new(rep["ov"].contents.num_bases)
    out = (H.OrcM4 * 100)()
    buf = C.create_string_buffer(512)
    lines = []
    al = O.orc_xaligner_new() if tech else O.orc_aligner_new()
    # (the X-drop restatement takes milliseconds per candidate: the nanopore set compares the first 40 query reads' lines — column 2
    # of an m4 line is the query read — and leaves the whole file to the hash of tests/golden/rep.json on the GPU side)
    nq = 40 if tech else len(rep["lens"])
    want = [ln for ln in want if int(ln.split()[1]) < nq]
    for rid in range(nq):
        if tech:
            k = O.orc_map_read_x(rep["ov"], rep["ov"], rep["oidx"], bk, None, al, rid, C.byref(p), out)
        else:
            k = O.orc_map_read(rep["ov"], rep["ov"], rep["oidx"], bk, al, rid, C.byref(p), out)
        for i in range(k):
            n = O.orc_m4_line(C.byref(out[i]), 1, buf)
            lines.append(buf.raw[:n].decode().rstrip("\n"))
    O.orc_bk_free(bk)
    (O.orc_xaligner_free if tech else O.orc_aligner_free)(al)
    assert len(want) > 500
    assert sorted(lines) == want
