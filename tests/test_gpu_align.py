"""GPU parity of the dw extension kernel (mhip_align_candidates) against the oracle's DiffAligner restatement and the
golden .m4 outputs of the unmodified reference.  Integer coordinates and match/column counts: bit-exact."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

G = json.load(open(os.path.join(H.GOLDEN, "golden.json")))
KAT = np.load(os.path.join(H.GOLDEN, "kats.npz"))


@pytest.fixture(scope="module")
def hip():
    import mecat_amd.hip as M
    return M


@pytest.fixture(scope="module")
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def _orc_go(al, q, qs, t, ts, mn):
    o = H.OrcAlnResult()
    H.orc().orc_dw_go(al, q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), mn, C.byref(o))
    return (o.ok, o.query_start, o.query_end, o.target_start, o.target_end, o.matches, o.columns)


def _gpu_tuple(r):
    return tuple(int(r[f]) for f in ("ok", "query_start", "query_end", "target_start", "target_end", "matches", "columns"))


def _pairs_volume(hip, ctx, seqs):
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    codes = np.concatenate(seqs).astype(np.uint8)
    ov = H.orc_pack(codes, lens)
    offs, pac = H.vol_arrays(ov)
    return hip.Volume(ctx, pac, offs, ov.contents.num_bases, 0)


def test_dw_golden_kats(hip, ctx):
    """the DiffAligner::go known answers generated from the reference (tests/golden/kats.npz)"""
    seqs, jobs, want = [], [], []
    qo = to = 0
    for i, (par, res) in enumerate(zip(KAT["dw_par"], KAT["dw_res"])):
        nq, nt, qs, ts, mn = [int(x) for x in par]
        seqs.append(KAT["dw_q"][qo: qo + nq]); qo += nq
        seqs.append(KAT["dw_t"][to: to + nt]); to += nt
        jobs.append((2 * i, 2 * i + 1, 0, qs, ts))
        want.append(tuple(int(x) for x in res))
    gv = _pairs_volume(hip, ctx, seqs)
    out = hip.align_candidates(ctx, gv, gv, np.array(jobs, dtype=hip.JOB_DTYPE), 500)
    bad = [(i, _gpu_tuple(out[i]), want[i]) for i in range(len(jobs)) if _gpu_tuple(out[i]) != want[i]]
    assert not bad, bad[:5]
    gv.free()


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


def test_dw_random_pairs_both_strands(hip, ctx):
    """random overlapping pairs incl. reverse-complemented queries, seed points on/off the diagonal, at the read ends,
    identical sequences (long snakes) and unrelated sequences (early stop)"""
    rng = np.random.default_rng(77)
    al = H.orc().orc_aligner_new()
    seqs, jobs, want = [], [], []
    for it in range(160):
        n = int(rng.integers(300, 7000))
        g = rng.integers(0, 4, size=n + 2000).astype(np.int8)
        a0, b0 = int(rng.integers(0, 1000)), int(rng.integers(0, 1000))
        e = [0.15, 0.15, 0.05, 0.0, 0.3][it % 5]
        q = _mutate(rng, g[a0: a0 + n], e)
        t = _mutate(rng, g[b0: b0 + n], e) if it % 13 else rng.integers(0, 4, size=n).astype(np.int8)
        if len(q) < 20 or len(t) < 20:
            continue
        mid = max(a0, b0) + n // 3
        qs = int((mid - a0) * (1 + 0.35 * e)) if it % 7 else int(rng.integers(0, len(q)))
        ts = int((mid - b0) * (1 + 0.35 * e)) if it % 7 else int(rng.integers(0, len(t)))
        qs = min(max(qs, 0), len(q) - 1)
        ts = min(max(ts, 0), len(t) - 1)
        if it % 17 == 0:
            qs = 0
        if it % 19 == 0:
            ts = len(t) - 1
        chain = it % 2
        # the aligner sees the strand-specific query; the volume holds the read as sequenced
        stored_q = (3 - q[::-1]).astype(np.int8) if chain else q
        i = len(seqs)
        seqs.append(stored_q)
        seqs.append(t)
        jobs.append((i, i + 1, chain, qs, ts))
        want.append(_orc_go(al, np.ascontiguousarray(q), qs, np.ascontiguousarray(t), ts, 500))
    gv = _pairs_volume(hip, ctx, seqs)
    out = hip.align_candidates(ctx, gv, gv, np.array(jobs, dtype=hip.JOB_DTYPE), 500)
    bad = [(i, jobs[i], _gpu_tuple(out[i]), want[i]) for i in range(len(jobs)) if _gpu_tuple(out[i]) != want[i]]
    assert not bad, "%d/%d differ: %s" % (len(bad), len(jobs), bad[:4])
    assert sum(w[0] for w in want) > 40
    H.orc().orc_aligner_free(al)
    gv.free()


def _m4_lines(hip, ctx, name, gapped, maxc=100):
    """full -j 1 body on the GPU (seed -> dw), records/post-filter/formatting by the oracle's A14 restatement"""
    g = G["sets"][name]["gen"]
    codes, lens = H.synth_reads(g["nreads"], g["L"], g["err"], g["genome"], g["seed"], g["ont"])
    ov = H.orc_pack(codes, lens)
    offs, pac = H.vol_arrays(ov)
    gv = hip.Volume(ctx, pac, offs, ov.contents.num_bases, 0)
    gi = hip.Index(ctx, gv)
    p = hip.default_params(0, maxc=maxc)
    cands, cnt = hip.seed_reads(ctx, gi, gv, gv, 0, len(lens), p)
    jobs, meta = [], []
    for rid in range(len(lens)):
        for c in cands[rid][: cnt[rid]]:
            qstart, sstart = int(c["loc2"]), int(c["loc1"])
            if qstart and sstart:
                qstart += 6
                sstart += 6
            jobs.append((rid, int(c["readno"]), int(c["chain"]), qstart, sstart))
            meta.append((rid, int(c["readno"]), int(c["chain"]), qstart, sstart, int(c["score"])))
    res = hip.align_candidates(ctx, gv, gv, np.array(jobs, dtype=hip.JOB_DTYPE), p.min_align_size)
    O = H.orc()
    lines = []
    buf = C.create_string_buffer(512)
    i = 0
    for rid in range(len(lens)):
        m4v = (H.OrcM4 * max(1, int(cnt[rid])))()
        k = 0
        for _ in range(int(cnt[rid])):
            r, (q, s, chain, qstart, sstart, score) = res[i], meta[i]
            i += 1
            if not r["ok"]:
                continue
            ar = H.OrcAlnResult(int(r["ok"]), int(r["query_start"]), int(r["query_end"]), int(r["target_start"]),
                                int(r["target_end"]), int(r["matches"]), int(r["columns"]))
            O.orc_m4_fill(C.byref(ar), q, s, b"R" if chain else b"F", int(offs[q, 1]), int(offs[s, 1]), qstart, sstart, score,
                          C.byref(m4v[k]))
            k += 1
        out = (H.OrcM4 * max(1, k))()
        kept = O.orc_m4_postfilter(m4v, k, out)
        for j in range(kept):
            n = O.orc_m4_line(C.byref(out[j]), gapped, buf)
            lines.append(buf.raw[:n].decode().rstrip("\n"))
    gi.free()
    gv.free()
    return sorted(lines)


@pytest.mark.parametrize("gapped", [0, 1])
def test_m4_tiny_matches_golden(gapped, hip, ctx):
    lines = _m4_lines(hip, ctx, "tiny", gapped)
    assert lines == open(os.path.join(H.GOLDEN, "tiny.g%d.m4.sorted" % gapped)).read().splitlines()


def test_m4_config1_matches_golden(hip, ctx):
    """config 1 of BASELINE.json (1 000 x 10 kb, 15 % error): whole -j 1 -g 1 output vs the reference's sha256"""
    lines = _m4_lines(hip, ctx, "config1", 1)
    gs = G["sets"]["config1"]
    assert len(lines) == gs["m4_g1_lines"]
    assert hashlib.sha256(("\n".join(lines) + "\n").encode()).hexdigest() == gs["m4_g1_sorted_sha256"]
    assert sum(int(x.split("\t")[6]) - int(x.split("\t")[5]) for x in lines) == gs["m4_aligned_bases"]
