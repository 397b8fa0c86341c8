"""Parity on REPEAT-STRUCTURED genomes: interspersed repeat families at 0 - 5 % divergence on either strand,
microsatellites and homopolymer runs (synth_genome_repeats in mecat_amd/tools/synth_reads.c).  Such inputs fill k-mer buckets up to
and beyond the index's cap of 128 (lookup_table.cpp:97), run the 41st-seed replacement rule on hits that are NOT self hits
(pw_impl.cpp:121-159), tie scores in the top-MAXC list (pw_impl.cpp:442-455) and hand the aligners low-complexity sequence.

  * the fixed sets of helpers.REP_SETS: index, candidates (MAXC 100 and 4) and extension against the oracle — which is itself pinned
    on the same sets against the unmodified reference (tests/test_oracle_rep_cpu.py, tests/golden/rep.json)
  * random repeat-structured sets (short version of `REP=1 tools/dev/parity_sweep.py` / `align_sweep.py`)
  * the CLI at config-1 size on a repeat-rich set: sorted `.can` / `.m4` == the reference binary's (hashes in tests/golden/rep.json,
    and side by side where oracle/_ref/mecat2pw travelled)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(H.GOLDEN, "rep.json")))
BIN = os.path.join(H.ROOT, "mecat_amd", "bin", "mecat2pw")
REF = os.path.join(H.ROOT, "oracle", "_ref", "mecat2pw")
FIELDS = ("ok", "query_start", "query_end", "target_start", "target_end", "matches", "columns")


@pytest.fixture(scope="module")
def ctx():
    import mecat_amd.hip as M
    c = M.Context(0)
    yield c
    c.close()


def _cmp_cands(got, cnt, want, lens):
    return [r for r, w in enumerate(want) if not (4 <= lens[r] < 13)
            and not (cnt[r] == len(w) and all(np.array_equal(got[r][: cnt[r]][f], w[f]) for f in H.CAND_DTYPE.names))]


def _extend_and_compare(M, ctx, gv, codes, lens, cands, cnt, ont, p, rng, max_jobs):
    O = H.orc()
    jobs = []
    for rid in range(len(lens)):
        for c in cands[rid][: cnt[rid]]:
            qs, ss = int(c["loc2"]), int(c["loc1"])
            if qs and ss:
                qs, ss = qs + 6, ss + 6
            jobs.append((rid, int(c["readno"]), int(c["chain"]), qs, ss))
    if len(jobs) > max_jobs:
        jobs = [jobs[i] for i in sorted(rng.choice(len(jobs), max_jobs, replace=False))]
    res = M.align_candidates(ctx, gv, gv, np.array(jobs, dtype=M.JOB_DTYPE), p.min_align_size, tech=ont)
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    al = O.orc_xaligner_new() if ont else O.orc_aligner_new()
    go = O.orc_xdrop_go if ont else O.orc_dw_go
    bad, aligned = [], 0
    for j, (q, s, chain, qs, ss) in enumerate(jobs):
        qq = codes[starts[q]: starts[q + 1]].astype(np.int8)
        qq = np.ascontiguousarray((3 - qq[::-1]).astype(np.int8) if chain else qq)
        tt = np.ascontiguousarray(codes[starts[s]: starts[s + 1]].astype(np.int8))
        o = H.OrcAlnResult()
        go(al, qq.ctypes.data, qs, len(qq), tt.ctypes.data, ss, len(tt), p.min_align_size, C.byref(o))
        want = (o.ok, o.query_start, o.query_end, o.target_start, o.target_end, o.matches, o.columns)
        got = tuple(int(res[j][f]) for f in FIELDS)
        if got != want:
            bad.append((jobs[j], got, want))
        aligned += want[0]
    (O.orc_xaligner_free if ont else O.orc_aligner_free)(al)
    return bad, aligned, len(jobs)


@pytest.mark.parametrize("name", sorted(H.REP_SETS))
def test_fixed_repeat_sets_match_the_oracle(name, ctx):
    import mecat_amd.hip as M
    codes, lens, ont, st = H.rep_set(name)
    g = G[name]
    assert g["genome_repeats"] == st and g["bases"] == int(lens.sum())
    assert g["buckets"]["dropped"] > 50 and g["insert_loc"]["non_self_dropped"] > 100      # what the set is for (recorded by make_golden_rep.py)
    ov = H.orc_pack(codes, lens)
    oidx = H.orc().orc_index_build(ov)
    oi = oidx.contents
    offs, pac = H.vol_arrays(ov)
    gv = M.Volume(ctx, pac, offs, ov.contents.num_bases, 0)
    gi = M.Index(ctx, gv)
    counts, offsets = gi.download()
    assert gi.num_kmers == oi.num_kmers
    assert np.array_equal(counts, np.ctypeslib.as_array(oi.counts, shape=(H.NK,)))
    assert np.array_equal(offsets, np.ctypeslib.as_array(oi.offsets, shape=(oi.num_kmers,)))
    rng = np.random.default_rng(5)
    for maxc in (100, 4):
        p = M.default_params(ont, maxc=maxc)
        ctx.reset_stats()
        got, cnt = M.seed_reads(ctx, gi, gv, gv, 0, len(lens), p)
        want = H.orc_seed_all(ov, ov, oidx, H.orc_params(tech=ont, maxc=maxc))
        bad = _cmp_cands(got, cnt, want, lens)
        assert not bad, (name, maxc, bad[:5])
        if maxc == 100:
            assert int(cnt.sum()) == g["candidates"]
            lines = H.can_lines_from_cands([got[r][: cnt[r]] for r in range(len(lens))], offs, offs)
            assert H.sha256_lines(lines) == g["can"]["sorted_sha256"]           # == the reference binary's .can
            bad, aligned, njobs = _extend_and_compare(M, ctx, gv, codes, lens, got, cnt, ont, p, rng, 2500)
            assert not bad, (name, len(bad), bad[:3])
            assert aligned > 1000
    gi.free()
    gv.free()


@pytest.mark.parametrize("sweep_seed", [31, 32])
def test_random_repeat_structured_sets(sweep_seed, ctx):
    import mecat_amd.hip as M
    rng = np.random.default_rng(sweep_seed)
    non_self = dropped = 0
    for it in range(4):
        ont = int(rng.integers(0, 2))
        nreads = int(rng.integers(200, 700))
        L = int(rng.integers(2500, 9000))
        err = float(rng.choice([0.06, 0.10, 0.15]))
        cov = float(rng.choice([10, 25, 40]))
        genome = max(30000, int(nreads * L / cov))
        nfam, mc, nsat = int(rng.integers(2, 9)), int(rng.choice([8, 30, 100])), int(rng.integers(0, 40))
        seed = int(rng.integers(1, 1 << 30))
        maxc = int(rng.choice([100, 100, 10, 3]))
        codes, lens, st = H.synth_reads_rep(nreads, L, err, genome, seed, ont, nfam, mc, nsat)
        ov = H.orc_pack(codes, lens)
        oidx = H.orc().orc_index_build(ov)
        offs, pac = H.vol_arrays(ov)
        gv = M.Volume(ctx, pac, offs, ov.contents.num_bases, 0)
        gi = M.Index(ctx, gv)
        counts, offsets = gi.download()
        oi = oidx.contents
        assert np.array_equal(counts, np.ctypeslib.as_array(oi.counts, shape=(H.NK,)))
        assert np.array_equal(offsets, np.ctypeslib.as_array(oi.offsets, shape=(oi.num_kmers,)))
        p = M.default_params(ont, maxc=maxc)
        got, cnt = M.seed_reads(ctx, gi, gv, gv, 0, len(lens), p)
        H.orc_stats_reset()
        want = H.orc_seed_all(ov, ov, oidx, H.orc_params(tech=ont, maxc=maxc))
        s = H.orc_stats()
        non_self += s["non_self"]
        dropped += H.bucket_stats(codes, lens)["dropped"]
        bad = _cmp_cands(got, cnt, want, lens)
        assert not bad, (it, ont, nreads, L, err, cov, nfam, mc, nsat, seed, maxc, bad[:5])
        bad, aligned, njobs = _extend_and_compare(M, ctx, gv, codes, lens, got, cnt, ont, p, rng, 500)
        assert not bad, (it, ont, seed, len(bad), bad[:3])
        gi.free()
        gv.free()
    assert non_self > 1000 and dropped > 100


def _lines(binary, tmp_path, tag, fa, args):
    out = str(tmp_path / (tag + ".out"))
    wrk = tmp_path / ("w_" + tag)
    wrk.mkdir()
    r = subprocess.run([binary, "-d", fa, "-o", out, "-w", str(wrk), "-t", "16"] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return sorted(open(out).read().splitlines())


def test_cli_on_a_repeat_rich_set_at_config1_size(tmp_path):
    """the drop-in binary on helpers.REP_CLI (1 000 reads x 10 kb @ 15 %, 8 repeat families of up to 120 copies): all three tasks"""
    g = G["rep_cli"]
    codes, lens, ont, st = H.rep_set("rep_cli")
    fa = str(tmp_path / "rep.fa")
    H.write_fasta(fa, codes, lens)
    assert H.sha256_lines(open(fa).read().splitlines()) == g["fasta_sha256"]
    for task in ("can", "m4_g0", "m4_g1"):
        got = _lines(BIN, tmp_path, "hip_" + task, fa, g[task]["args"])
        assert len(got) == g[task]["lines"], task
        assert H.sha256_lines(got) == g[task]["sorted_sha256"], task
    if os.path.exists(REF):
        assert _lines(REF, tmp_path, "ref_m4", fa, g["m4_g1"]["args"]) == _lines(BIN, tmp_path, "hip_m4b", fa, g["m4_g1"]["args"])
