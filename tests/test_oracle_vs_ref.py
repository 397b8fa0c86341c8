"""Pins the oracle restatement against the UNMODIFIED reference compiled into oracle/_ref (function level, random inputs).
Skipped where oracle/_ref is absent.  The committed golden vectors (test_oracle_golden.py) cover the same ground when it is."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.skipif(not H.ref_available(), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def tiny():
    codes, lens = H.synth_reads(200, 3000, 0.15, 30000, 11)
    d = tempfile.mkdtemp(prefix="orc_ref_")
    fa = os.path.join(d, "tiny.fa")
    H.write_fasta(fa, codes, lens)
    wrk = os.path.join(d, "wrk")
    os.makedirs(wrk)
    R = H.ref()
    assert R.refh_split(fa.encode(), wrk.encode()) == 1
    rv = R.refh_load_volume(os.path.join(wrk, "vol0").encode())
    ridx = R.refh_build_index(rv, 1)
    ov = H.orc_pack(codes, lens)
    oidx = H.orc().orc_index_build(ov)
    yield dict(codes=codes, lens=lens, dir=d, fa=fa, wrk=wrk, rv=rv, ridx=ridx, ov=ov, oidx=oidx)


def test_volume_bytes(tiny):
    ov = tiny["ov"]
    p = os.path.join(tiny["dir"], "vol0.orc")
    assert H.orc().orc_volume_dump(ov, p.encode()) == 0
    assert open(p, "rb").read() == open(os.path.join(tiny["wrk"], "vol0"), "rb").read()


def test_index_identical(tiny):
    R = H.ref()
    counts = np.empty(H.NK, dtype=np.int32)
    n = R.refh_index_dump(tiny["ridx"], counts.ctypes.data, None)
    offs = np.empty(n, dtype=np.int32)
    R.refh_index_dump(tiny["ridx"], counts.ctypes.data, offs.ctypes.data)
    oi = tiny["oidx"].contents
    assert oi.num_kmers == n
    oc = np.ctypeslib.as_array(oi.counts, shape=(H.NK,))
    assert np.array_equal(oc, counts)
    assert np.array_equal(np.ctypeslib.as_array(oi.offsets, shape=(n,)), offs)


@pytest.mark.parametrize("tech,maxc", [(0, 100), (0, 5), (1, 100)])
def test_candidates_identical(tiny, tech, maxc):
    R = H.ref()
    p = H.orc_params(tech=tech, maxc=maxc)
    R.refh_set_params(maxc, p.min_align_size, p.min_kmer_match, tech)
    ours = H.orc_seed_all(tiny["ov"], tiny["ov"], tiny["oidx"], p)
    out = np.zeros((maxc, 12), dtype=np.int32)
    tot = 0
    for rid in range(len(tiny["lens"])):
        k = R.refh_seed_read(tiny["rv"], tiny["rv"], tiny["ridx"], rid, 0, out.ctypes.data)
        a = ours[rid]
        assert k == len(a), rid
        got = np.stack([a[n] for n in H.CAND_DTYPE.names], axis=1) if k else np.zeros((0, 12), np.int32)
        assert np.array_equal(got, out[:k]), rid
        tot += k
    assert tot > 100


def test_insert_loc_random():
    R, O = H.ref(), H.orc()
    rng = np.random.default_rng(5)
    for it in range(300):
        bl = H.OrcBackList()
        mode = it % 3
        base = int(rng.integers(0, 1500))
        for i in range(40):
            if mode == 0:      # perfect diagonal (self hit)
                bl.loczhi[i] = (base + 10 * i) % 2000
                bl.seedno[i] = 1 + i
            elif mode == 1:    # noisy diagonal
                bl.loczhi[i] = int(np.clip(base + 10 * i + rng.integers(-8, 9), 0, 1999))
                bl.seedno[i] = 1 + i + int(rng.integers(0, 2))
            else:
                bl.loczhi[i] = int(rng.integers(0, 2000))
                bl.seedno[i] = int(rng.integers(1, 1500))
        bl.score = 41
        for step in range(12):
            loc = int(rng.integers(0, 2000)) if mode == 2 else int(np.clip(base + 10 * (40 + step) + rng.integers(-3, 4), 0, 1999))
            seedn = int(rng.integers(1, 1500)) if mode == 2 else 41 + step
            sc = np.array([bl.score], dtype=np.int16)
            lz = np.array(list(bl.loczhi), dtype=np.int16)
            sn = np.array(list(bl.seedno), dtype=np.int16)
            R.refh_insert_loc(sc.ctypes.data, lz.ctypes.data, sn.ctypes.data, loc, seedn, 10.0)
            O.orc_insert_loc(C.byref(bl), loc, seedn, 10.0, 0.25)
            assert bl.score == sc[0]
            assert list(bl.loczhi) == list(lz) and list(bl.seedno) == list(sn)
            bl.score += 1


def test_find_location_random():
    R, O = H.ref(), H.orc()
    rng = np.random.default_rng(7)
    hits = 0
    for it in range(2000):
        k = int(rng.integers(1, 81))
        if it % 2:
            seedn = np.sort(rng.integers(1, 400, size=k)).astype(np.int32)
            loc = (seedn * 10 + rng.integers(-30, 31, size=k) + 200).astype(np.int32)
            if it % 4 == 1:
                rng.shuffle(loc[: k // 2])
        else:
            seedn = rng.integers(1, 200, size=k).astype(np.int32)
            loc = rng.integers(0, 4000, size=k).astype(np.int32)
        read_len = int(rng.integers(500, 20000))
        res = []
        for lib, extra in ((R.refh_find_location, ()), (O.orc_find_location, (0.25,))):
            sc = np.zeros(k, dtype=np.int32)
            lo = np.zeros(4, dtype=np.int32)
            rep = C.c_int(-1)
            l2, s2 = loc.copy(), seedn.copy()
            r = lib(l2.ctypes.data, s2.ctypes.data, sc.ctypes.data, lo.ctypes.data, k, C.byref(rep), 10.0, read_len, *extra)
            res.append((r, tuple(sc), tuple(lo) if r else None, rep.value if r else None))
        assert res[0] == res[1]
        hits += res[0][0]
    assert hits > 100


def _mutate(rng, s, e):
    out = []
    for b in s:
        u = rng.random()
        if u < 0.25 * e:
            continue
        out.append(int(rng.integers(0, 4)) if u < 0.4 * e else int(b))
        if rng.random() < 0.6 * e:
            out.append(int(rng.integers(0, 4)))
    return np.array(out, dtype=np.int8)


def test_align_random():
    R, O = H.ref(), H.orc()
    al = O.orc_aligner_new()
    rng = np.random.default_rng(9)
    for it in range(300):
        n = int(rng.integers(20, 600))
        q = rng.integers(0, 4, size=n).astype(np.int8)
        e = [0.0, 0.05, 0.15, 0.3, 0.6][it % 5]
        t = _mutate(rng, q, e)
        if len(t) < 5:
            continue
        band = int(0.3 * max(len(q), len(t))) if it % 7 else int(rng.integers(2, 40))
        right = it % 2
        outs = []
        for fn, pre in ((R.refh_align, ()), (O.orc_align, (al,))):
            res = np.zeros(6, dtype=np.int32)
            qa = np.zeros(4096, dtype=np.int8)
            ta = np.zeros(4096, dtype=np.int8)
            r = fn(*pre, q.ctypes.data, len(q), t.ctypes.data, len(t), band, 1, right, res.ctypes.data, qa.ctypes.data, ta.ctypes.data)
            outs.append((r, tuple(res), qa[: res[0]].tobytes(), ta[: res[0]].tobytes()))
        assert outs[0] == outs[1], it
    O.orc_aligner_free(al)


def test_dw_go_random():
    R, O = H.ref(), H.orc()
    al = O.orc_aligner_new()
    rng = np.random.default_rng(13)
    oks = 0
    for it in range(60):
        n = int(rng.integers(800, 6000))
        g = rng.integers(0, 4, size=n + 2000).astype(np.int8)
        a0, b0 = int(rng.integers(0, 1000)), int(rng.integers(0, 1000))
        q = _mutate(rng, g[a0: a0 + n], 0.15)
        t = _mutate(rng, g[b0: b0 + n], 0.15)
        # seed point roughly on the shared diagonal (or off it for some cases)
        mid = max(a0, b0) + n // 3
        qs = int((mid - a0) * 1.05) if it % 5 else int(rng.integers(0, len(q)))
        ts = int((mid - b0) * 1.05) if it % 5 else int(rng.integers(0, len(t)))
        qs = min(max(qs, 0), len(q) - 1)
        ts = min(max(ts, 0), len(t) - 1)
        if it % 11 == 0:
            qs = 0
        res_r = np.zeros(7, dtype=np.int32)
        ident = C.c_double()
        R.refh_dw_go(q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), 500, res_r.ctypes.data, C.byref(ident))
        o = H.OrcAlnResult()
        O.orc_dw_go(al, q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), 500, C.byref(o))
        got = (o.ok, o.query_start, o.query_end, o.target_start, o.target_end, o.matches, o.columns)
        assert got == tuple(res_r), it
        if o.columns:
            assert 100.0 * o.matches / o.columns == ident.value
        oks += o.ok
    assert oks > 10
    O.orc_aligner_free(al)


def _run_ref(tiny, args, name):
    out = os.path.join(tiny["dir"], name)
    wrk = os.path.join(tiny["dir"], "w_" + name)
    subprocess.run([H.ref_bin(), "-d", tiny["fa"], "-o", out, "-w", wrk, "-t", "2"] + args, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(open(out).read().splitlines())


@pytest.mark.parametrize("tech", [0, 1])
def test_can_output_identical(tiny, tech):
    want = _run_ref(tiny, ["-j", "0", "-x", str(tech)], "t%d.can" % tech)
    p = H.orc_params(tech=tech)
    cands = H.orc_seed_all(tiny["ov"], tiny["ov"], tiny["oidx"], p)
    offs, _ = H.vol_arrays(tiny["ov"])
    O = H.orc()
    lines = []
    buf = C.create_string_buffer(256)
    ec = H.OrcExtCandidate()
    out = (H.OrcCandidate * 100)()
    bk = O.orc_bk_new(tiny["ov"].contents.num_bases)
    for rid in range(len(tiny["lens"])):
        k = O.orc_seed_read(tiny["ov"], tiny["ov"], tiny["oidx"], bk, rid, 0, C.byref(p), out)
        for i in range(k):
            O.orc_can_record(C.byref(out[i]), rid, int(offs[rid, 1]), int(offs[out[i].readno, 1]), C.byref(ec))
            n = O.orc_can_line(C.byref(ec), buf)
            lines.append(buf.raw[:n].decode().rstrip("\n"))
    O.orc_bk_free(bk)
    assert sorted(lines) == want
    assert sorted(H.can_lines_from_cands(cands, offs, offs)) == want


@pytest.mark.parametrize("gapped", [0, 1])
def test_m4_output_identical(tiny, gapped):
    want = _run_ref(tiny, ["-j", "1", "-g", str(gapped)], "t%d.m4" % gapped)
    O = H.orc()
    p = H.orc_params(tech=0)
    bk = O.orc_bk_new(tiny["ov"].contents.num_bases)
    al = O.orc_aligner_new()
    out = (H.OrcM4 * 100)()
    buf = C.create_string_buffer(512)
    lines = []
    for rid in range(len(tiny["lens"])):
        k = O.orc_map_read(tiny["ov"], tiny["ov"], tiny["oidx"], bk, al, rid, C.byref(p), out)
        for i in range(k):
            n = O.orc_m4_line(C.byref(out[i]), gapped, buf)
            lines.append(buf.raw[:n].decode().rstrip("\n"))
    O.orc_bk_free(bk)
    O.orc_aligner_free(al)
    assert len(want) > 50
    assert sorted(lines) == want


def _mutate_ont(rng, s, e):
    out = []
    for b in s:
        u = rng.random()
        if u < 0.35 * e:
            continue
        out.append(int(rng.integers(0, 4)) if u < 0.70 * e else int(b))
        if rng.random() < 0.30 * e:
            out.append(int(rng.integers(0, 4)))
    return np.array(out, dtype=np.int8)


def test_xdrop_align_random():
    R, O = H.ref(), H.orc()
    xa = O.orc_xaligner_new()
    rng = np.random.default_rng(21)
    for it in range(200):
        n = int(rng.integers(1, 700))
        q = rng.integers(0, 4, size=n).astype(np.int8)
        e = [0.0, 0.05, 0.12, 0.3, 0.7][it % 5]
        t = _mutate_ont(rng, q, e) if it % 9 else rng.integers(0, 4, size=int(rng.integers(1, 700))).astype(np.int8)
        if len(t) == 0:
            continue
        fwd = it % 2
        outs = []
        for fn, pre in ((R.refh_xdrop_align, ()), (O.orc_xdrop_align, (xa,))):
            res = np.zeros(3, np.int32)
            ops = np.zeros(2 * 4096, np.int32)
            sc = fn(*pre, q.ctypes.data, len(q), t.ctypes.data, len(t), fwd, res.ctypes.data, ops.ctypes.data)
            outs.append((sc, tuple(res), tuple(ops[: 2 * res[2]])))
        assert outs[0] == outs[1], it
    O.orc_xaligner_free(xa)


def test_xdrop_go_random():
    R, O = H.ref(), H.orc()
    xa = O.orc_xaligner_new()
    rng = np.random.default_rng(23)
    oks = 0
    for it in range(60):
        n = int(rng.integers(600, 6000))
        g = rng.integers(0, 4, size=n + 2000).astype(np.int8)
        a0, b0 = int(rng.integers(0, 1000)), int(rng.integers(0, 1000))
        q = _mutate_ont(rng, g[a0: a0 + n], 0.12)
        t = _mutate_ont(rng, g[b0: b0 + n], 0.12)
        mid = max(a0, b0) + n // 3
        qs = int((mid - a0) * 0.995) if it % 5 else int(rng.integers(0, len(q)))
        ts = int((mid - b0) * 0.995) if it % 5 else int(rng.integers(0, len(t)))
        qs = min(max(qs, 0), len(q) - 1)
        ts = min(max(ts, 0), len(t) - 1)
        if it % 11 == 0:
            qs = 0
        res_r = np.zeros(7, dtype=np.int32)
        ident = C.c_double()
        R.refh_xdrop_go(q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), 500, res_r.ctypes.data, C.byref(ident))
        o = H.OrcAlnResult()
        O.orc_xdrop_go(xa, q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), 500, C.byref(o))
        got = (o.ok, o.query_start, o.query_end, o.target_start, o.target_end, o.matches, o.columns)
        assert got == tuple(res_r), it
        oks += o.ok
    assert oks > 10
    O.orc_xaligner_free(xa)


def test_m4_nanopore_output_identical(tmp_path):
    """-x 1 -j 1 -g 1: XdropAligner end to end"""
    codes, lens = H.synth_reads(150, 4000, 0.12, 40000, 12, 1)
    fa = str(tmp_path / "ont.fa")
    H.write_fasta(fa, codes, lens)
    out = str(tmp_path / "ont.m4")
    subprocess.run([H.ref_bin(), "-j", "1", "-x", "1", "-g", "1", "-d", fa, "-o", out, "-w", str(tmp_path / "w"), "-t", "2"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    want = sorted(open(out).read().splitlines())
    O = H.orc()
    ov = H.orc_pack(codes, lens)
    oidx = O.orc_index_build(ov)
    p = H.orc_params(tech=1)
    bk = O.orc_bk_new(ov.contents.num_bases)
    al, xa = O.orc_aligner_new(), O.orc_xaligner_new()
    outm = (H.OrcM4 * 100)()
    buf = C.create_string_buffer(512)
    lines = []
    for rid in range(len(lens)):
        k = O.orc_map_read_x(ov, ov, oidx, bk, al, xa, rid, C.byref(p), outm)
        for i in range(k):
            n = O.orc_m4_line(C.byref(outm[i]), 1, buf)
            lines.append(buf.raw[:n].decode().rstrip("\n"))
    assert len(want) > 100
    assert sorted(lines) == want


@pytest.mark.skipif(not H.ref_cns_available(), reason="mecat2cns reference harness not built (oracle/_ref)")
@pytest.mark.parametrize("error_rate", [0.15, 0.20])
def test_cns_aligner_random(error_rate):
    """N1: ns_banded_sw::dw and GetAlignment of mecat2cns (reference) vs orc_cns_dw / orc_cns_get_alignment"""
    R, O = H.ref_cns(), H.orc()
    a = O.orc_cns_new()
    rng = np.random.default_rng(17)
    oks = 0
    for it in range(80):
        n = int(rng.integers(300, 7000))
        q, t, qs, ts = H.cns_pair(rng, n, [0.0, 0.05, 0.12, 0.15, 0.3][it % 5], it)
        min_aln = 500 if it % 3 else 50
        outs = []
        for fn, pre in ((R.refc_dw, ()), (O.orc_cns_dw, (a,))):
            res = np.zeros(9, dtype=np.int32)
            s1 = np.zeros(100001, dtype=np.int8)
            s2 = np.zeros(100001, dtype=np.int8)
            r = fn(*pre, q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), error_rate, min_aln, res.ctypes.data, s1.ctypes.data, s2.ctypes.data)
            outs.append((r, tuple(res), s1[: res[4]].tobytes(), s2[: res[4]].tobytes()))
        assert outs[0] == outs[1], it
        outs = []
        for fn, pre in ((R.refc_get_alignment, ()), (O.orc_cns_get_alignment, (a,))):
            res = np.zeros(5, dtype=np.int32)
            s1 = np.zeros(100001, dtype=np.int8)
            s2 = np.zeros(100001, dtype=np.int8)
            r = fn(*pre, q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), error_rate, min_aln, res.ctypes.data, s1.ctypes.data, s2.ctypes.data)
            outs.append((r, tuple(res), s1[: res[4]].tobytes(), s2[: res[4]].tobytes()))
        assert outs[0] == outs[1], it
        oks += outs[0][0]
    assert oks > 20
    O.orc_cns_free(a)


@pytest.mark.parametrize("tech", [0, 1])
def test_candidates_identical_beyond_the_short_seed_numbers(tech):
    """reads of 420 kb (the reference's `short` seed numbers wrap from read position 327 670 on): the oracle keeps them in int16 like the
    reference and gives the reference's candidates there too — the pin behind tests/test_gpu_parity.py::test_reads_beyond_the_short_seed_numbers"""
    c1, l1 = H.synth_reads(6, 400000, 0.15, 450000, 41)
    c2, l2 = H.synth_reads(40, 8000, 0.15, 450000, 42)
    codes, lens = np.concatenate([c1, c2]), np.concatenate([l1, l2])
    d = tempfile.mkdtemp(prefix="orc_ref_long_")
    fa = os.path.join(d, "x.fa")
    H.write_fasta(fa, codes, lens)
    wrk = os.path.join(d, "wrk")
    os.makedirs(wrk)
    R = H.ref()
    assert R.refh_split(fa.encode(), wrk.encode()) == 1
    rv = R.refh_load_volume(os.path.join(wrk, "vol0").encode())
    ridx = R.refh_build_index(rv, 1)
    ov = H.orc_pack(codes, lens)
    oidx = H.orc().orc_index_build(ov)
    p = H.orc_params(tech=tech, maxc=100)
    R.refh_set_params(100, p.min_align_size, p.min_kmer_match, tech)
    ours = H.orc_seed_all(ov, ov, oidx, p)
    out = np.zeros((100, 12), dtype=np.int32)
    tot = 0
    for rid in range(len(lens)):
        k = R.refh_seed_read(rv, rv, ridx, rid, 0, out.ctypes.data)
        a = ours[rid]
        assert k == len(a), rid
        got = np.stack([a[n] for n in H.CAND_DTYPE.names], axis=1) if k else np.zeros((0, 12), np.int32)
        assert np.array_equal(got, out[:k]), rid
        tot += k
    assert tot >= 30


@pytest.mark.parametrize("n", [60, 999, 1000, 1500, 4000])
def test_m4_sort_and_containment_filter_tie_order(n):
    """append_m4v (pw_impl.cpp:576-610) on lists with MANY ties, through the 1 000-element threshold of libstdc++'s parallel-mode
    std::sort (the repository takes -n up to 2^20, the reference's per-read sort is std::sort under -D_GLIBCXX_PARALLEL).
    oracle/_ref is compiled the way the reference's release build is (no -fopenmp on compile lines): the parallel-mode sort then runs as
    one piece and keeps the sequential introsort's order of equal keys — which is what the restatement (plain std::sort, m4sort.cpp) and
    the product's driver (plain std::sort) produce.  Records carry their input position in vscore, so the comparison sees the order."""
    R, O = H.ref(), H.orc()
    R.refh_append_m4v.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    R.refh_append_m4v.restype = C.c_int
    rng = np.random.default_rng(1000 + n)
    for it in range(6):
        recs = (H.OrcM4 * n)()
        nq = max(1, n // (3 + 4 * it))                       # few subjects: long runs of equal qid
        for i in range(n):
            m = recs[i]
            m.qid = int(rng.integers(0, nq)); m.sid = 7; m.ident = 90.0; m.vscore = i; m.qdir = 0
            size = int(rng.choice([2000, 2500, 3000, 3000, 4000]))      # few sizes: many ties inside a qid
            m.qoff = int(rng.integers(0, 300)) * (it % 2); m.qend = m.qoff + size; m.qsize = 20000
            m.sdir = int(rng.integers(0, 2))
            m.soff = int(rng.integers(0, 300)) * (it % 3 == 0); m.send = m.soff + size + int(rng.integers(0, 50)); m.ssize = 20000
            m.qext = m.qoff + 10; m.sext = m.soff + 10
        a = (H.OrcM4 * n)(); b = (H.OrcM4 * n)(); work = (H.OrcM4 * n)()
        C.memmove(work, recs, C.sizeof(recs))
        ka = R.refh_append_m4v(C.byref(recs), n, C.byref(a))
        kb = O.orc_m4_postfilter(C.byref(work), n, C.byref(b))
        assert ka == kb, (n, it, ka, kb)
        assert [a[i].vscore for i in range(ka)] == [b[i].vscore for i in range(kb)], (n, it)
        assert bytes(a)[: ka * C.sizeof(H.OrcM4)] == bytes(b)[: kb * C.sizeof(H.OrcM4)]
