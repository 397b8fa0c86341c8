"""development: randomized extension parity sweep — every candidate of N random synthetic read sets (PacBio-style: dw_extend / dw_extend2,
nanopore-style: xd_extend_w and its wide launch) extended on the GPU and by the CPU oracle's aligner restatements, compared field by field
(ok, both intervals, matches, columns).  Jobs are built from the device's own candidates the way the host builds them (start points + 6
when both are non-zero); JOBS caps the jobs checked per set (the oracle's X-drop takes milliseconds per job)."""
import ctypes as C
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))      # tests/helpers.py
import helpers as H
import mecat_amd.hip as M

rng = np.random.default_rng(int(os.environ.get("SEED", "4242")))
JOBS = int(os.environ.get("JOBS", "1500"))
ctx = M.Context(0)
O = H.orc()
FIELDS = ("ok", "query_start", "query_end", "target_start", "target_end", "matches", "columns")
bad_total = jobs_total = ok_total = 0
for it in range(int(os.environ.get("N", "20"))):
    ont = int(rng.integers(0, 2))
    nreads = int(rng.integers(100, 700))
    L = int(rng.integers(2500, 14000))
    err = float(rng.choice([0.06, 0.12, 0.15, 0.18]))
    cov = float(rng.choice([4, 10, 25]))
    genome = max(20000, int(nreads * L / cov))
    seed = int(rng.integers(1, 1 << 30))
    if int(os.environ.get("REP", "0")):      # repeat-structured genome: low-complexity and multi-copy sequence under the aligners
        codes, lens, _st = H.synth_reads_rep(nreads, L, err, genome, seed, ont, int(rng.integers(2, 9)), int(rng.choice([8, 30, 100])), int(rng.integers(0, 40)))
    else:
        codes, lens = H.synth_reads(nreads, L, err, genome, seed, ont)
    t0 = time.time()
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
                qs += 6; ss += 6
            jobs.append((rid, int(c["readno"]), int(c["chain"]), qs, ss))
    if len(jobs) > JOBS:
        jobs = [jobs[i] for i in sorted(rng.choice(len(jobs), JOBS, replace=False))]
    if not jobs:
        print("set %2d: no candidates" % it, flush=True); gi.free(); gv.free(); continue
    res = M.align_candidates(ctx, gv, gv, np.array(jobs, dtype=M.JOB_DTYPE), p.min_align_size, tech=ont)
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    al = O.orc_xaligner_new() if ont else O.orc_aligner_new()
    go = O.orc_xdrop_go if ont else O.orc_dw_go
    bad = []
    for j, (q, s, chain, qs, ss) in enumerate(jobs):
        qq = codes[starts[q]: starts[q + 1]].astype(np.int8)
        if chain:
            qq = (3 - qq[::-1]).astype(np.int8)
        qq = np.ascontiguousarray(qq)
        tt = np.ascontiguousarray(codes[starts[s]: starts[s + 1]].astype(np.int8))
        o = H.OrcAlnResult()
        go(al, qq.ctypes.data, qs, len(qq), tt.ctypes.data, ss, len(tt), p.min_align_size, C.byref(o))
        want = (o.ok, o.query_start, o.query_end, o.target_start, o.target_end, o.matches, o.columns)
        got = tuple(int(res[j][f]) for f in FIELDS)
        if got != want:
            bad.append((jobs[j], got, want))
        ok_total += want[0]
    (O.orc_xaligner_free if ont else O.orc_aligner_free)(al)
    bad_total += len(bad); jobs_total += len(jobs)
    print("set %2d: ont %d reads %4d L %5d err %.2f cov %3.0f: %5d jobs, %5d aligned, differing %d  (%.1f s)%s"
          % (it, ont, len(lens), L, err, cov, len(jobs), int(sum(int(r["ok"]) for r in res)), len(bad), time.time() - t0, ("  " + str(bad[:2])) if bad else ""), flush=True)
    gi.free(); gv.free()
print("TOTAL jobs %d, aligned %d, differing %d" % (jobs_total, ok_total, bad_total))
sys.exit(1 if bad_total else 0)
