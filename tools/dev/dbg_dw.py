import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mecat_amd.hip as M
from mecat_amd import workload as W
n = int(os.environ.get("N", "20000"))
codes, lens = W.synth_reads(n, 15000, 0.15, int(5e7 * n / 1e5), 2, 0)
pac, offs, nb = W.pack_volume(codes, lens)
ctx = M.Context(0); vol = M.Volume(ctx, pac, offs, nb, 0); idx = M.Index(ctx, vol); p = M.default_params(0)
cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, n, p)
jobs = W.jobs_from_candidates(cands, cnt, 0)
res = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size).copy()
os.environ["MECAT_DW_KERNEL"] = "1"
res1 = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size).copy()
bad = np.nonzero(res.view(np.int32).reshape(len(jobs), -1) != res1.view(np.int32).reshape(len(jobs), -1))[0]
bad = np.unique(bad)
print("jobs", len(jobs), "differ", len(bad))
for i in bad[:8]:
    print(i, jobs[i], "\n  got ", res[i], "\n  want", res1[i])
