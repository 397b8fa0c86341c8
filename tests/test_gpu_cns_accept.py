"""mecat2cns' candidate accept loop on the device re-aligner (mhip_cns_accept_templates; SURVEY.md §8f row N1, BASELINE config 4)
against the UNMODIFIED reference: tests/golden/cns_accept.npz holds, for 240 PacBio-style and 160 ONT-style templates at ~100x
coverage, the normalised candidate records that were fed to consensus_one_read_can_pacbio / _nanopore (compiled from
/root/reference/src/mecat2cns) and what each call left in CnsAlns: (soff, send, aln_size) of every accepted alignment in order
and a SHA-256 over the gap-normalised strings.  The coverage cap, the used-read set, the mapping-range check and the 200-candidate
window are all active on these sets.  Everything must match exactly."""
import hashlib
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(H.GOLDEN, "cns_accept.npz"))


@pytest.mark.parametrize("name", ["pacbio", "nanopore"])
def test_accept_loop_equals_reference(name):
    import mecat_amd.hip as M
    from mecat_amd import workload as W
    n, L, Gn, seed, ont, tech, mas = (int(x) for x in G[name + "_par"])
    err, ratio = (float(x) for x in G[name + "_ratio"])
    codes, lens = W.synth_reads(n, L, err, Gn, seed, ont)
    pac, offs, nb = W.pack_volume(codes, lens)
    ctx = M.Context(0)
    vol = M.Volume(ctx, pac, offs, nb, 0)
    cands = G[name + "_cands"].copy()
    tb = G[name + "_tmpl_begin"]
    acc, strings, njobs = M.cns_accept_templates(ctx, vol, pac, cands, tb, tech, mas, ratio, threads=16)
    assert njobs == int(np.minimum(np.diff(tb), 200).sum())
    want_n = G[name + "_nacc"]
    got_n = np.bincount(acc["template_index"], minlength=n)
    assert np.array_equal(got_n, want_n), np.nonzero(got_n != want_n)[0][:10]
    assert np.array_equal(np.stack([acc["soff"], acc["send"], acc["aln_size"]], axis=1), G[name + "_meta"])
    assert np.all(np.diff(acc["template_index"]) >= 0)
    first = np.concatenate([[0], np.cumsum(want_n)])
    sha = G[name + "_sha"]
    for t in range(n):
        if want_n[t] == 0:
            continue
        a = acc[first[t]: first[t + 1]]
        lo, hi = int(a["str_offset"][0]), int(a["str_offset"][-1]) + 2 * (int(a["aln_size"][-1]) + 1)
        assert hashlib.sha256(strings[lo:hi]).hexdigest() == str(sha[t]), t
        # the accepted record points at the candidate it came from (after the in-place sort) and repeats its ids
        assert np.all(cands[a["cand_index"], 1] == a["qid"]) and np.all(cands[a["cand_index"], 7] == t)
        assert len(set(a["qid"].tolist())) == len(a)           # a query read is used at most once per template
    assert int(want_n.sum()) > 4000 and want_n.max() > 30
    vol.free()
    ctx.close()


def test_accept_loop_on_config2_overlaps_equals_reference():
    """BASELINE config 4 proper: the templates mecat2cns would build from config 2's own overlaps (the device's candidate table of the whole
    100 000-read set, = the reference mecat2pw's `-j 0` file, test_gpu_bigconfigs.py::test_config2_cli_equals_reference), the first 2 000 of
    them against what the UNMODIFIED consensus_one_read_can_pacbio accepted on the same records (tests/golden/cns_config2.npz, written by
    tests/golden/make_golden_cns_config2.py from the reference mecat2pw's candidate file)."""
    import mecat_amd.hip as M
    from mecat_amd import workload as W
    g = np.load(os.path.join(H.GOLDEN, "cns_config2.npz"))
    mas, T = (int(x) for x in g["par"])
    ratio = float(g["ratio"][0])
    n, L, err, Gn, seed, ont = W.CONFIGS["config2"]
    codes, lens = W.synth_reads(n, L, err, Gn, seed, ont)
    pac, offs, nb = W.pack_volume(codes, lens)
    del codes
    ctx = M.Context(0)
    vol = M.Volume(ctx, pac, offs, nb, 0)
    idx = M.Index(ctx, vol)
    params = M.default_params(0)
    cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, n, params)
    idx.free()
    ec = W.ext_candidates_from_table(cands, cnt, lens)
    rec, tb, ids = W.cns_templates(ec, n)
    assert np.array_equal(ids[:T], g["ids"]) and np.array_equal(tb[: T + 1], g["tmpl_begin"])
    rec = np.ascontiguousarray(rec[: tb[T]]).copy()
    assert hashlib.sha256(rec.tobytes()).hexdigest() == str(g["rec_sha"])          # the records the reference was given
    acc, strings, njobs = M.cns_accept_templates(ctx, vol, pac, rec, tb[: T + 1], 0, mas, ratio, threads=32)
    assert njobs == int(np.minimum(np.diff(tb[: T + 1]), 200).sum())
    want_n = g["nacc"]
    got_n = np.bincount(acc["template_index"], minlength=T)
    assert np.array_equal(got_n, want_n), np.nonzero(got_n != want_n)[0][:10]
    assert np.array_equal(np.stack([acc["soff"], acc["send"], acc["aln_size"]], axis=1), g["meta"])
    first = np.concatenate([[0], np.cumsum(want_n)])
    for t in range(T):
        if want_n[t] == 0:
            continue
        a = acc[first[t]: first[t + 1]]
        lo, hi = int(a["str_offset"][0]), int(a["str_offset"][-1]) + 2 * (int(a["aln_size"][-1]) + 1)
        assert hashlib.sha256(strings[lo:hi]).hexdigest() == str(g["sha"][t]), t
    assert int(want_n.sum()) > 10000
    vol.free()
    ctx.close()
