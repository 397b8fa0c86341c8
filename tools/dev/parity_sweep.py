"""development: randomized candidate parity sweep — GPU (seed_strand / kernel chain, cuts, pre-drop) against the CPU oracle on a set of
random synthetic read sets of varying size, length, error, coverage and technology, plus ragged read lengths."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))      # tests/helpers.py
import helpers as H
import mecat_amd.hip as M

rng = np.random.default_rng(int(os.environ.get("SEED", "2024")))
REP = int(os.environ.get("REP", "0"))
tot_stats = dict(at_cap=0, dropped=0, insert_loc=0, non_self=0, non_self_dropped=0)
ctx = M.Context(0)
bad_total = 0
for it in range(int(os.environ.get("N", "14"))):
    ont = int(rng.integers(0, 2))
    nreads = int(rng.integers(150, 1400))
    L = int(rng.integers(2500, 12000))
    err = float(rng.choice([0.08, 0.12, 0.15, 0.18]))
    cov = float(rng.choice([4, 10, 25, 60]))
    genome = max(20000, int(nreads * L / cov))
    seed = int(rng.integers(1, 1 << 30))
    maxc = int(rng.choice([100, 100, 10, 3]))
    rep = ""
    if REP:      # repeat-structured genome (synth_genome_repeats): families at 0 - 5 % divergence, microsatellites, homopolymer runs
        nfam, mc, nsat = int(rng.integers(2, 9)), int(rng.choice([8, 30, 100])), int(rng.integers(0, 40))
        codes, lens, st = H.synth_reads_rep(nreads, L, err, genome, seed, ont, nfam, mc, nsat)
        rep = " rep(fam %d copies<=%d sat %d: %d copies)" % (nfam, mc, nsat, st["copies"])
    else:
        codes, lens = H.synth_reads(nreads, L, err, genome, seed, ont)
    if it % 3 == 0:        # ragged: cut some reads short (down to below k)
        starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
        parts, nl = [], []
        for i in range(len(lens)):
            n = int(lens[i]) if rng.random() > 0.2 else int(rng.integers(1, max(2, int(lens[i]))))
            parts.append(codes[starts[i]: starts[i] + n]); nl.append(n)
        codes, lens = np.concatenate(parts), np.array(nl, dtype=np.int32)
    t0 = time.time()
    ov = H.orc_pack(codes, lens)
    oidx = H.orc().orc_index_build(ov)
    offs, pac = H.vol_arrays(ov)
    gv = M.Volume(ctx, pac, offs, ov.contents.num_bases, 0)
    gi = M.Index(ctx, gv)
    p = M.default_params(ont, maxc=maxc)
    ctx.reset_stats()
    got, cnt = M.seed_reads(ctx, gi, gv, gv, 0, len(lens), p)
    took, left = ctx.debug_counter(13), ctx.debug_counter(14)
    H.orc_stats_reset()
    want = H.orc_seed_all(ov, ov, oidx, H.orc_params(tech=ont, maxc=maxc))
    os_, bs_ = H.orc_stats(), H.bucket_stats(codes, lens)
    for k in tot_stats:
        tot_stats[k] += os_.get(k, 0) + bs_.get(k, 0)
    # (reads of 4 .. 12 bases: one k-mer running past the read in the reference — undefined there, no k-mers here — are left out)
    bad = [r for r, w in enumerate(want) if not (4 <= lens[r] < 13) and not (cnt[r] == len(w) and all(np.array_equal(got[r][: cnt[r]][f], w[f]) for f in H.CAND_DTYPE.names))]
    bad_total += len(bad)
    print("set %2d: ont %d reads %4d L %5d err %.2f cov %4.0f maxc %3d ragged %d: %6d candidates, strands fused %d chain %d, differing reads %d  (%.1f s)%s  buckets at cap %d dropped %d, 41st-seed rule %d (non-self %d, dropped one %d)"
          % (it, ont, len(lens), L, err, cov, maxc, it % 3 == 0, int(cnt.sum()), took, left, len(bad), time.time() - t0, rep, bs_["at_cap"], bs_["dropped"], os_["insert_loc"], os_["non_self"], os_["non_self_dropped"]), flush=True)
    gi.free(); gv.free()
print("TOTAL buckets at the cap of 128: %(at_cap)d, dropped beyond it: %(dropped)d; 41st-seed rule ran %(insert_loc)d times, %(non_self)d on non-self hits (%(non_self_dropped)d of those dropped an entry)" % tot_stats)
print("TOTAL differing reads:", bad_total)
sys.exit(1 if bad_total else 0)
