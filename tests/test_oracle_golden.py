"""Oracle (oracle/liboracle.so) against the committed golden vectors generated from the unmodified reference
(tests/golden/make_golden.py).  Runs anywhere (no GPU, no /root/reference)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import helpers as H

G = json.load(open(os.path.join(H.GOLDEN, "golden.json")))
KAT = np.load(os.path.join(H.GOLDEN, "kats.npz"))


def sha(b):
    return hashlib.sha256(b).hexdigest()


_cache = {}


def dataset(name):
    if name not in _cache:
        g = G["sets"][name]["gen"]
        codes, lens = H.synth_reads(g["nreads"], g["L"], g["err"], g["genome"], g["seed"], g["ont"])
        assert sha(lens.tobytes()) == G["sets"][name]["lens_sha256"]
        v = H.orc_pack(codes, lens)
        idx = H.orc().orc_index_build(v)
        _cache[name] = (codes, lens, v, idx)
    return _cache[name]


def test_sizeof():
    assert G["sizeof"] == {"Back_List": 168, "candidate_save": 48, "M4Record": 104, "ExtensionCandidate": 52,
                           "DPathData2": 28, "offset_t": 8, "volume_t": 32}


@pytest.mark.parametrize("name", ["tiny", "tiny_ont", "config1"])
def test_volume_and_index(name, tmp_path):
    codes, lens, v, idx = dataset(name)
    p = str(tmp_path / "vol0")
    assert H.orc().orc_volume_dump(v, p.encode()) == 0
    assert sha(open(p, "rb").read()) == G["sets"][name]["vol0_sha256"]
    gi = G["sets"][name]["index"]
    ii = idx.contents
    assert ii.num_kmers == gi["num_kmers"]
    counts = np.ctypeslib.as_array(ii.counts, shape=(H.NK,))
    assert sha(counts.tobytes()) == gi["counts_sha256"]
    offs = np.ctypeslib.as_array(ii.offsets, shape=(ii.num_kmers,))
    assert sha(offs.tobytes()) == gi["offsets_sha256"]
    if "buckets" in gi:
        starts = np.ctypeslib.as_array(ii.starts, shape=(H.NK,))
        for k, want in gi["buckets"].items():
            k = int(k)
            assert list(offs[starts[k]: starts[k] + counts[k]]) == want


@pytest.mark.parametrize("name", ["tiny", "tiny_ont"])
@pytest.mark.parametrize("maxc", [100, 5])
def test_candidates(name, maxc):
    codes, lens, v, idx = dataset(name)
    tech = G["sets"][name]["gen"]["ont"]
    z = np.load(os.path.join(H.GOLDEN, name + "_cands.npz"))
    want, cnts = z["cands_maxc%d" % maxc], z["counts_maxc%d" % maxc]
    got = H.orc_seed_all(v, v, idx, H.orc_params(tech=tech, maxc=maxc))
    assert [len(a) for a in got] == list(cnts)
    flat = np.concatenate([np.stack([a[n] for n in H.CAND_DTYPE.names], axis=1) for a in got if len(a)])
    assert np.array_equal(flat, want)


@pytest.mark.parametrize("name", ["tiny", "tiny_ont", "config1"])
def test_can_lines(name):
    codes, lens, v, idx = dataset(name)
    tech = G["sets"][name]["gen"]["ont"]
    cands = H.orc_seed_all(v, v, idx, H.orc_params(tech=tech))
    offs, _ = H.vol_arrays(v)
    lines = sorted(H.can_lines_from_cands(cands, offs, offs))
    assert len(lines) == G["sets"][name]["can_lines"]
    assert sha(("\n".join(lines) + "\n").encode()) == G["sets"][name]["can_sorted_sha256"]
    if name != "config1":
        assert lines == open(os.path.join(H.GOLDEN, name + ".can.sorted")).read().splitlines()


@pytest.mark.parametrize("gapped", [0, 1])
def test_m4_lines_tiny(gapped):
    codes, lens, v, idx = dataset("tiny")
    O = H.orc()
    p = H.orc_params(tech=0)
    bk = O.orc_bk_new(v.contents.num_bases)
    al = O.orc_aligner_new()
    out = (H.OrcM4 * 100)()
    buf = C.create_string_buffer(512)
    lines = []
    for rid in range(len(lens)):
        k = O.orc_map_read(v, v, idx, bk, al, rid, C.byref(p), out)
        for i in range(k):
            n = O.orc_m4_line(C.byref(out[i]), gapped, buf)
            lines.append(buf.raw[:n].decode().rstrip("\n"))
    O.orc_bk_free(bk)
    O.orc_aligner_free(al)
    assert sorted(lines) == open(os.path.join(H.GOLDEN, "tiny.g%d.m4.sorted" % gapped)).read().splitlines()


def test_find_location_kats():
    O = H.orc()
    fin, fout = KAT["find_location_in"], KAT["find_location_out"]
    for a, b in zip(fin, fout):
        k, rl = int(a[0]), int(a[1])
        loc, seedn = a[2: 2 + k].copy(), a[82: 82 + k].copy()
        sc = np.zeros(k, np.int32)
        lo = np.zeros(4, np.int32)
        rep = C.c_int(-1)
        r = O.orc_find_location(loc.ctypes.data, seedn.ctypes.data, sc.ctypes.data, lo.ctypes.data, k, C.byref(rep), 10.0, rl, 0.25)
        assert r == b[0]
        assert list(sc) == list(b[6: 6 + k])
        if r:
            assert rep.value == b[1] and list(lo) == list(b[2:6])
    # SURVEY.md §8c known answer
    assert list(fout[0][:6]) == [1, 0, 100, 1, 170, 8] and list(fout[0][6:15]) == [7, 7, 7, 7, 7, 7, 7, 7, 0]


def test_insert_loc_kats():
    O = H.orc()
    for a, b in zip(KAT["insert_loc_in"], KAT["insert_loc_out"]):
        bl = H.OrcBackList()
        bl.score = int(a[0])
        for i in range(40):
            bl.loczhi[i] = int(a[3 + i])
            bl.seedno[i] = int(a[43 + i])
        O.orc_insert_loc(C.byref(bl), int(a[1]), int(a[2]), 10.0, 0.25)
        assert bl.score == b[0]
        assert list(bl.loczhi) == list(b[1:41]) and list(bl.seedno) == list(b[41:81])


def test_align_kats():
    O = H.orc()
    al = O.orc_aligner_new()
    qo = to = ao = 0
    for par, want in zip(KAT["align_par"], KAT["align_res"]):
        nq, nt, band, right = [int(x) for x in par]
        q = KAT["align_q"][qo: qo + nq].copy(); qo += nq
        t = KAT["align_t"][to: to + nt].copy(); to += nt
        res = np.zeros(6, np.int32)
        qa = np.zeros(4096, np.int8)
        ta = np.zeros(4096, np.int8)
        r = O.orc_align(al, q.ctypes.data, nq, t.ctypes.data, nt, band, 1, right, res.ctypes.data, qa.ctypes.data, ta.ctypes.data)
        assert [r] + list(res) == list(want)
        n = int(want[1])
        assert np.array_equal(qa[:n], KAT["align_qaln"][ao: ao + n]) and np.array_equal(ta[:n], KAT["align_taln"][ao: ao + n])
        ao += n
    # SURVEY.md §8c known answer: dist 2, 21 columns
    assert list(KAT["align_res"][0]) == [1, 21, 2, 0, 20, 0, 20]
    dec = "".join("ACGT-"[c] for c in KAT["align_qaln"][:21])
    assert dec == "ACGTACGTACGTACG-TACGT"
    O.orc_aligner_free(al)


def test_dw_go_kats():
    O = H.orc()
    al = O.orc_aligner_new()
    qo = to = 0
    for par, want in zip(KAT["dw_par"], KAT["dw_res"]):
        nq, nt, qs, ts, mn = [int(x) for x in par]
        q = KAT["dw_q"][qo: qo + nq].copy(); qo += nq
        t = KAT["dw_t"][to: to + nt].copy(); to += nt
        o = H.OrcAlnResult()
        O.orc_dw_go(al, q.ctypes.data, qs, nq, t.ctypes.data, ts, nt, mn, C.byref(o))
        assert [o.ok, o.query_start, o.query_end, o.target_start, o.target_end, o.matches, o.columns] == list(want)
    O.orc_aligner_free(al)


def test_xdrop_go_kats():
    """XdropAligner::go known answers (nanopore mode)"""
    O = H.orc()
    xa = O.orc_xaligner_new()
    qo = to = 0
    for par, want in zip(KAT["xd_par"], KAT["xd_res"]):
        nq, nt, qs, ts, mn = [int(x) for x in par]
        q = KAT["xd_q"][qo: qo + nq].copy(); qo += nq
        t = KAT["xd_t"][to: to + nt].copy(); to += nt
        o = H.OrcAlnResult()
        O.orc_xdrop_go(xa, q.ctypes.data, qs, nq, t.ctypes.data, ts, nt, mn, C.byref(o))
        assert [o.ok, o.query_start, o.query_end, o.target_start, o.target_end, o.matches, o.columns] == list(want)
    O.orc_xaligner_free(xa)


def test_m4_lines_tiny_ont():
    """-x 1 -j 1 -g 1 (X-drop aligner) against the reference's sorted output"""
    codes, lens, v, idx = dataset("tiny_ont")
    O = H.orc()
    p = H.orc_params(tech=1)
    bk = O.orc_bk_new(v.contents.num_bases)
    al, xa = O.orc_aligner_new(), O.orc_xaligner_new()
    out = (H.OrcM4 * 100)()
    buf = C.create_string_buffer(512)
    lines = []
    for rid in range(len(lens)):
        k = O.orc_map_read_x(v, v, idx, bk, al, xa, rid, C.byref(p), out)
        for i in range(k):
            n = O.orc_m4_line(C.byref(out[i]), 1, buf)
            lines.append(buf.raw[:n].decode().rstrip("\n"))
    assert sorted(lines) == open(os.path.join(H.GOLDEN, "tiny_ont.g1.m4.sorted")).read().splitlines()


def test_cns_aligner_kats():
    """N1: known answers of the mecat2cns re-aligner (ns_banded_sw::dw, GetAlignment) from the unmodified reference"""
    import hashlib
    K = np.load(os.path.join(H.GOLDEN, "cns_kats.npz"))
    O = H.orc()
    a = O.orc_cns_new()
    qo = to = 0
    for par, dw_want, ga_want, dig in zip(K["par"], K["dw_res"], K["ga_res"], K["digests"]):
        nq, nt, qs, ts, mn, er100 = [int(x) for x in par]
        q = K["q"][qo: qo + nq].copy(); qo += nq
        t = K["t"][to: to + nt].copy(); to += nt
        res = np.zeros(9, np.int32)
        s1 = np.zeros(100001, np.int8)
        s2 = np.zeros(100001, np.int8)
        ok = O.orc_cns_dw(a, q.ctypes.data, qs, nq, t.ctypes.data, ts, nt, er100 / 100.0, mn, res.ctypes.data, s1.ctypes.data, s2.ctypes.data)
        assert [ok] + list(res) == list(dw_want)
        d1 = hashlib.sha256(s1[: res[4]].tobytes() + b"|" + s2[: res[4]].tobytes()).hexdigest()
        res2 = np.zeros(5, np.int32)
        ok2 = O.orc_cns_get_alignment(a, q.ctypes.data, qs, nq, t.ctypes.data, ts, nt, er100 / 100.0, mn, res2.ctypes.data, s1.ctypes.data, s2.ctypes.data)
        assert [ok2] + list(res2) == list(ga_want)
        d2 = hashlib.sha256(s1[: res2[4]].tobytes() + b"|" + s2[: res2[4]].tobytes()).hexdigest() if ok2 else ""
        assert d1 + ":" + d2 == str(dig)
    O.orc_cns_free(a)


def test_xdrop_row_scan_formulation_equals_sequential_row():
    """the X-drop row as per-cell work + prefix scans (oracle/xdrop_rowpar.c, what a wave-parallel kernel computes) against
    the literal sequential row, state by state, on random / unrelated / low-complexity blocks"""
    O = H.orc()
    rng = np.random.default_rng(3)

    def mut(s, e):
        out = []
        for b in s:
            u = rng.random()
            if u < 0.35 * e:
                continue
            out.append(int(rng.integers(0, 4)) if u < 0.7 * e else int(b))
            if rng.random() < 0.3 * e:
                out.append(int(rng.integers(0, 4)))
        return np.array(out, dtype=np.int8)

    rows = 0
    for it in range(600):
        n = int(rng.integers(5, 740))
        g = rng.integers(0, 4, size=n).astype(np.int8)
        e = [0.0, 0.04, 0.12, 0.2, 0.35, 0.6][it % 6]
        a = mut(g, e)
        b = mut(g, e) if it % 9 else rng.integers(0, 4, size=n).astype(np.int8)
        if it % 11 == 0:
            a = np.tile(rng.integers(0, 4, size=int(rng.integers(1, 5))).astype(np.int8), 300)[:n]
            b = mut(a, 0.1)
        a, b = a[:736], b[:736]
        if len(a) < 2 or len(b) < 2:
            continue
        assert O.orc_xdrop_rowpar_selfcheck(a.ctypes.data, len(a), b.ctypes.data, len(b)) == 0, it
        rows += len(a)
    assert rows > 100000
