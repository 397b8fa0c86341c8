"""Randomized candidate parity: the device path (seed_strand and the kernel chain mixed strand by strand, bucket cuts, early drop of
higher-id subjects) against the CPU oracle on seeded random read sets of varying size, read length, error, coverage, technology
and MAXC, every third one with ragged read lengths (down to a single base).  tools/dev/parity_sweep.py is the long version
(125 sets run clean when this was written)."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sweep_seed", [11, 12])
def test_random_read_sets_match_the_oracle(sweep_seed):
    import mecat_amd.hip as M
    rng = np.random.default_rng(sweep_seed)
    ctx = M.Context(0)
    fused = chain = 0
    for it in range(6):
        ont = int(rng.integers(0, 2))
        nreads = int(rng.integers(150, 1000))
        L = int(rng.integers(2500, 11000))
        err = float(rng.choice([0.08, 0.12, 0.15, 0.18]))
        cov = float(rng.choice([4, 10, 25, 60]))
        genome = max(20000, int(nreads * L / cov))
        seed = int(rng.integers(1, 1 << 30))
        maxc = int(rng.choice([100, 100, 10, 3]))
        codes, lens = H.synth_reads(nreads, L, err, genome, seed, ont)
        if it % 3 == 0:
            starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
            parts, nl = [], []
            for i in range(len(lens)):
                n = int(lens[i]) if rng.random() > 0.2 else int(rng.integers(1, max(2, int(lens[i]))))
                parts.append(codes[starts[i]: starts[i] + n])
                nl.append(n)
            codes, lens = np.concatenate(parts), np.array(nl, dtype=np.int32)
        ov = H.orc_pack(codes, lens)
        oidx = H.orc().orc_index_build(ov)
        offs, pac = H.vol_arrays(ov)
        gv = M.Volume(ctx, pac, offs, ov.contents.num_bases, 0)
        gi = M.Index(ctx, gv)
        p = M.default_params(ont, maxc=maxc)
        ctx.reset_stats()
        got, cnt = M.seed_reads(ctx, gi, gv, gv, 0, len(lens), p)
        fused += ctx.debug_counter(13)
        chain += ctx.debug_counter(14)
        want = H.orc_seed_all(ov, ov, oidx, H.orc_params(tech=ont, maxc=maxc))
        # reads of 4 .. 12 bases: one k-mer that runs past the read in the reference (undefined there, no k-mers here) — left out
        bad = [r for r, w in enumerate(want) if not (4 <= lens[r] < 13)
               and not (cnt[r] == len(w) and all(np.array_equal(got[r][: cnt[r]][f], w[f]) for f in H.CAND_DTYPE.names))]
        assert not bad, (it, ont, nreads, L, err, cov, maxc, bad[:5])
        gi.free()
        gv.free()
    ctx.close()
    assert fused > 0


@pytest.mark.parametrize("sweep_seed", [21, 22])
def test_random_read_sets_extend_as_the_oracle_does(sweep_seed):
    """Extension parity on the device's own candidates: every field of every job (ok, both intervals, matches, columns) against the
    oracle's aligner restatements — dw (PacBio gates) and X-drop (nanopore gates) alternate.  tools/dev/align_sweep.py is the long
    version (profiles/r04_parity_sweeps.md)."""
    import ctypes as C
    import mecat_amd.hip as M
    rng = np.random.default_rng(sweep_seed)
    ctx = M.Context(0)
    O = H.orc()
    fields = ("ok", "query_start", "query_end", "target_start", "target_end", "matches", "columns")
    aligned = 0
    for it in range(4):
        ont = it & 1
        nreads = int(rng.integers(100, 500))
        L = int(rng.integers(2500, 12000))
        err = float(rng.choice([0.06, 0.12, 0.15, 0.18]))
        cov = float(rng.choice([4, 10, 25]))
        codes, lens = H.synth_reads(nreads, L, err, max(20000, int(nreads * L / cov)), int(rng.integers(1, 1 << 30)), ont)
        ov = H.orc_pack(codes, lens)
        offs, pac = H.vol_arrays(ov)
        gv = M.Volume(ctx, pac, offs, ov.contents.num_bases, 0)
        gi = M.Index(ctx, gv)
        p = M.default_params(ont)
        cands, cnt = M.seed_reads(ctx, gi, gv, gv, 0, len(lens), p)
        jobs = []
        for rid in range(len(lens)):
            for c in cands[rid][: cnt[rid]]:
                qs, ss = int(c["loc2"]), int(c["loc1"])
                if qs and ss:
                    qs, ss = qs + 6, ss + 6
                jobs.append((rid, int(c["readno"]), int(c["chain"]), qs, ss))
        if len(jobs) > 800:
            jobs = [jobs[i] for i in sorted(rng.choice(len(jobs), 800, replace=False))]
        assert jobs
        res = M.align_candidates(ctx, gv, gv, np.array(jobs, dtype=M.JOB_DTYPE), p.min_align_size, tech=ont)
        starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
        al = O.orc_xaligner_new() if ont else O.orc_aligner_new()
        go = O.orc_xdrop_go if ont else O.orc_dw_go
        bad = []
        for j, (q, s, chain, qs, ss) in enumerate(jobs):
            qq = codes[starts[q]: starts[q + 1]].astype(np.int8)
            qq = np.ascontiguousarray((3 - qq[::-1]).astype(np.int8) if chain else qq)
            tt = np.ascontiguousarray(codes[starts[s]: starts[s + 1]].astype(np.int8))
            o = H.OrcAlnResult()
            go(al, qq.ctypes.data, qs, len(qq), tt.ctypes.data, ss, len(tt), p.min_align_size, C.byref(o))
            want = (o.ok, o.query_start, o.query_end, o.target_start, o.target_end, o.matches, o.columns)
            got = tuple(int(res[j][f]) for f in fields)
            if got != want:
                bad.append((jobs[j], got, want))
            aligned += want[0]
        (O.orc_xaligner_free if ont else O.orc_aligner_free)(al)
        assert not bad, (it, ont, nreads, L, err, cov, len(bad), bad[:3])
        gi.free()
        gv.free()
    ctx.close()
    assert aligned > 500
