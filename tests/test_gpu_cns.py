"""GPU parity of the mecat2cns re-aligner (mhip_cns_align_candidates) against the known answers of
the unmodified reference (tests/golden/cns_kats.npz) and against the oracle restatement on random pairs, both strands.
Coordinates, counts and the aligned strings (rebuilt from the 2-bit columns) are bit-exact."""
import hashlib
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
CAP = 32768


@pytest.fixture(scope="module")
def hip():
    import mecat_amd.hip as M
    return M


@pytest.fixture(scope="module")
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def _volume(hip, ctx, seqs):
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    codes = np.concatenate(seqs).astype(np.uint8)
    ov = H.orc_pack(codes, lens)
    offs, pac = H.vol_arrays(ov)
    return hip.Volume(ctx, pac, offs, ov.contents.num_bases, 0)


def _orc(a, q, qs, t, ts, er, mn):
    O = H.orc()
    res = np.zeros(9, np.int32)
    s1 = np.zeros(100001, np.int8)
    s2 = np.zeros(100001, np.int8)
    ok = O.orc_cns_dw(a, q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), er, mn, res.ctypes.data, s1.ctypes.data, s2.ctypes.data)
    res2 = np.zeros(5, np.int32)
    ok2 = O.orc_cns_get_alignment(a, q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), er, mn, res2.ctypes.data, s1.ctypes.data, s2.ctypes.data)
    return ok, res, ok2, res2, s1[: res2[4]].tobytes(), s2[: res2[4]].tobytes()


def _check(hip, r, ops_row, q, t, want):
    ok, res, ok2, res2, qaln, saln = want
    # dw level: coordinates, column count, indel counts of the untrimmed string (the reference only fills the counts on success)
    assert (int(r["query_start"]), int(r["query_end"]), int(r["target_start"]), int(r["target_end"])) == tuple(int(x) for x in res[:4])
    assert int(r["left_cols"]) + int(r["right_cols"]) == int(res[4])
    if ok:
        assert (int(r["mat"]), int(r["ins"]), int(r["dele"])) == (int(res[5]), int(res[7]), int(res[8]))
    # GetAlignment level
    assert int(r["ok"]) == int(ok2)
    if ok2:
        assert (int(r["qoff"]), int(r["qend"]), int(r["soff"]), int(r["send"])) == tuple(int(x) for x in res2[:4])
        assert int(r["last_col"]) - int(r["first_col"]) == int(res2[4])
        gq, gs = hip.cns_expand(r, ops_row, q, t)
        assert gq == qaln and gs == saln


def test_cns_golden_kats(hip, ctx):
    K = np.load(os.path.join(H.GOLDEN, "cns_kats.npz"))
    seqs, cases = [], []
    qo = to = 0
    for par in K["par"]:
        nq, nt, qs, ts, mn, er100 = [int(x) for x in par]
        seqs += [K["q"][qo: qo + nq], K["t"][to: to + nt]]
        qo += nq; to += nt
        cases.append((qs, ts, mn, er100))
    vol = _volume(hip, ctx, seqs)
    for er100 in (15, 20):
        for mn in (50, 500):
            sel = [i for i, c in enumerate(cases) if c[3] == er100 and c[2] == mn]
            jobs = np.zeros(len(sel), dtype=hip.JOB_DTYPE)
            for j, i in enumerate(sel):
                jobs[j] = (2 * i, 2 * i + 1, 0, cases[i][0], cases[i][1])
            res, ops = hip.cns_align_candidates(ctx, vol, vol, jobs, er100 / 100.0, mn, CAP)
            for j, i in enumerate(sel):
                want_dw, want_ga = K["dw_res"][i], K["ga_res"][i]
                r = res[j]
                assert [int(r["query_start"]), int(r["query_end"]), int(r["target_start"]), int(r["target_end"]),
                        int(r["left_cols"]) + int(r["right_cols"])] == [int(x) for x in want_dw[1:6]], i
                assert int(r["ok"]) == int(want_ga[0]), i
                if want_ga[0]:
                    assert [int(r["qoff"]), int(r["qend"]), int(r["soff"]), int(r["send"]), int(r["last_col"]) - int(r["first_col"])] == \
                        [int(x) for x in want_ga[1:6]], i
                    gq, gs = hip.cns_expand(r, ops[j], seqs[2 * i], seqs[2 * i + 1])
                    # digests: "<untrimmed>:<trimmed>" of "qaln|saln"
                    assert hashlib.sha256(gq + b"|" + gs).hexdigest() == str(K["digests"][i]).split(":")[1], i
    vol.free()


@pytest.mark.parametrize("error_rate", [0.15, 0.20])
def test_cns_random_pairs_both_strands(hip, ctx, error_rate):
    rng = np.random.default_rng(23)
    O = H.orc()
    a = O.orc_cns_new()
    seqs, metas = [], []
    for it in range(120):
        n = int(rng.integers(200, 12000))
        q, t, qs, ts = H.cns_pair(rng, n, [0.0, 0.05, 0.12, 0.15, 0.18, 0.3][it % 6], it)
        chain = it % 2
        # the volume holds the read as sequenced; chain 1 means the aligner sees its reverse complement
        stored_q = (3 - q)[::-1].copy() if chain else q
        seqs += [stored_q.astype(np.int8), t]
        metas.append((q, t, qs, ts, chain))
    vol = _volume(hip, ctx, seqs)
    for mn in (50, 500):
        jobs = np.zeros(len(metas), dtype=hip.JOB_DTYPE)
        for i, (q, t, qs, ts, chain) in enumerate(metas):
            jobs[i] = (2 * i, 2 * i + 1, chain, qs, ts)
        res, ops = hip.cns_align_candidates(ctx, vol, vol, jobs, error_rate, mn, CAP)
        oks = 0
        for i, (q, t, qs, ts, chain) in enumerate(metas):
            want = _orc(a, q, qs, t, ts, error_rate, mn)
            try:
                _check(hip, res[i], ops[i], q, t, want)
            except AssertionError:
                print("case", i, "n", len(q), len(t), "qs", qs, "ts", ts, "chain", chain, "gpu", res[i], "orc", want[:4])
                raise
            oks += int(want[2])
        assert oks > 30
    O.orc_cns_free(a)
    vol.free()


def test_cns_argument_checks_and_tiny_reads(hip, ctx):
    rng = np.random.default_rng(5)
    base = rng.integers(0, 4, size=3000).astype(np.int8)
    tiny = [rng.integers(0, 4, size=n).astype(np.int8) for n in (1, 2, 3, 5, 9)]
    seqs = [base, base.copy()] + tiny
    vol = _volume(hip, ctx, seqs)
    jobs = np.zeros(1, dtype=hip.JOB_DTYPE)
    jobs[0] = (0, 1, 0, 1500, 1500)
    with pytest.raises(hip.MhipError, match="columns"):
        hip.cns_align_candidates(ctx, vol, vol, jobs, 0.15, 500, 16)          # 3000 identical columns do not fit 16
    with pytest.raises(hip.MhipError, match="error_rate"):
        hip.cns_align_candidates(ctx, vol, vol, jobs, 0.25, 500, CAP)
    with pytest.raises(hip.MhipError, match="multiple of 16"):
        hip.cns_align_candidates(ctx, vol, vol, jobs, 0.15, 500, 1000)
    res, ops = hip.cns_align_candidates(ctx, vol, vol, jobs, 0.15, 500, CAP)
    assert int(res[0]["ok"]) == 1 and (int(res[0]["qoff"]), int(res[0]["qend"])) == (0, 3000) and int(res[0]["mat"]) == 3000
    # reads shorter than the four-match anchor never align; every start position is legal
    O = H.orc()
    a = O.orc_cns_new()
    js, metas = [], []
    for i, s in enumerate(tiny):
        for qs in range(len(s)):
            js.append((2 + i, 0, 0, qs, 7))
            metas.append((s, base, qs, 7))
            js.append((0, 2 + i, 0, 11, qs))
            metas.append((base, s, 11, qs))
    jobs = np.array(js, dtype=hip.JOB_DTYPE)
    res, ops = hip.cns_align_candidates(ctx, vol, vol, jobs, 0.15, 1, CAP)
    for r, o, (q, t, qs, ts) in zip(res, ops, metas):
        _check(hip, r, o, q, t, _orc(a, q, qs, t, ts, 0.15, 1))
    O.orc_cns_free(a)
    vol.free()


@pytest.mark.parametrize("error_rate,ont", [(0.15, 0), (0.20, 1)])
def test_two_kernel_realigner_equals_one_unit_per_wave_kernel_at_scale(hip, ctx, error_rate, ont, monkeypatch):
    """round 5: the re-aligner's default path (forward rows = dw_extend2 under mecat2cns' block rules, one 16-byte record per row;
    paths by cns_trace, one lane per block; cns_extend for the units the forward pass hands over) against MECAT_CNS_KERNEL=1 (every unit
    through cns_extend, which keeps every row and is the kernel rounds 1-4 pinned to the reference): results and every column word of
    ~100 k candidates of a 30x read set, also with the row log cut into slices of 1 GB."""
    from mecat_amd import workload as W
    n = 5000
    codes, lens = W.synth_reads(n, 9000, 0.15 if not ont else 0.12, n * 9000 // 30, 31 + ont, ont)
    pac, offs, nb = W.pack_volume(codes, lens)
    vol = hip.Volume(ctx, pac, offs, nb, 0)
    idx = hip.Index(ctx, vol)
    p = hip.default_params(ont)
    cands, cnt = hip.seed_reads(ctx, idx, vol, vol, 0, n, p)
    jobs = W.jobs_from_candidates(cands, cnt, 0)
    assert len(jobs) > 50000
    cap = 16384
    monkeypatch.setenv("MECAT_CNS_KERNEL", "1")
    want_r, want_o = hip.cns_align_candidates(ctx, vol, vol, jobs, error_rate, 500, cap)
    monkeypatch.delenv("MECAT_CNS_KERNEL")
    for gb in (None, "1"):
        if gb:
            monkeypatch.setenv("MECAT_CNS_LOG_GB", gb)
        got_r, got_o = hip.cns_align_candidates(ctx, vol, vol, jobs, error_rate, 500, cap)
        assert int((want_r["ok"] != 0).sum()) > 0.8 * len(jobs)
        assert np.array_equal(got_r, want_r)
        assert np.array_equal(got_o, want_o)
    # units that run out of block records (a share of reach / 4096 + 2 instead of reach / 256 + 4) go to cns_extend like the units that
    # outgrow their share of the row log: same results (this used to fail the whole batch with "internal error 2")
    monkeypatch.delenv("MECAT_CNS_LOG_GB")
    monkeypatch.setenv("MECAT_CNS_BLOCK_SHARE", "12,2")
    got_r, got_o = hip.cns_align_candidates(ctx, vol, vol, jobs[:20000], error_rate, 500, cap)
    assert np.array_equal(got_r, want_r[:20000])
    assert np.array_equal(got_o, want_o[:20000])
    monkeypatch.delenv("MECAT_CNS_BLOCK_SHARE")
    idx.free()
    vol.free()
