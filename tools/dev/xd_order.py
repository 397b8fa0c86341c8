"""dev: does the order of the units matter to xd_extend_w?  The ONT-style candidates of xd_time.py extended in candidate order, in
descending order of the longer side's reach (what a longest-first deal would do), and in ascending order (the worst case: the longest
units start last) — kernel times of the three."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mecat_amd import hip as M, workload as W
n = int(os.environ.get("N", "5000"))
codes, lens = W.synth_reads(n, 10000, 0.12, int(1_700_000 * n / 5000), 7, 1)
pac, offs, nb = W.pack_volume(codes, lens)
ctx = M.Context(0); vol = M.Volume(ctx, pac, offs, nb, 0); idx = M.Index(ctx, vol)
p = M.default_params(1)
cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, len(lens), p)
jobs = W.jobs_from_candidates(cands, cnt, 0)
L = lens.astype(np.int64)
names = jobs.dtype.names
q, s, qs, ss = (jobs[names[0]].astype(np.int64), jobs[names[1]].astype(np.int64), jobs[names[3]].astype(np.int64), jobs[names[4]].astype(np.int64))
left = np.minimum(qs, ss); right = np.minimum(L[q] - qs, L[s] - ss)
reach = np.maximum(left, right)
for label, order in (("candidate order", np.arange(len(jobs))), ("longest first", np.argsort(-reach, kind="stable")), ("shortest first", np.argsort(reach, kind="stable"))):
    jj = np.ascontiguousarray(jobs[order])
    best = None
    for it in range(3):
        ctx.set_profiling(True); ctx.reset_stats()
        res = M.align_candidates(ctx, vol, vol, jj, p.min_align_size, tech=1)
        ks = ctx.kernel_stats()
        t = sum(v[1] for k, v in ks.items() if k.startswith("xd_extend"))
        best = t if best is None else min(best, t)
    print("%-16s %d jobs: xd_extend kernels %.1f ms (%s)" % (label, len(jobs), best, ", ".join("%s %.1f" % (k, v[1]) for k, v in ks.items() if k.startswith("xd"))), flush=True)
