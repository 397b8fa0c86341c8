import os, sys, time, numpy as np
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mecat_amd.hip as M
from mecat_amd import workload as W
n = int(os.environ.get("N", "20000"))
codes, lens = W.synth_reads(n, 15000, 0.15, int(5e7 * n / 1e5), 2, 0)
pac, offs, nb = W.pack_volume(codes, lens)
ctx = M.Context(0); vol = M.Volume(ctx, pac, offs, nb, 0); idx = M.Index(ctx, vol); p = M.default_params(0)
cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, n, p)
jobs = W.jobs_from_candidates(cands, cnt, 0)
ctx.reset_stats()
res = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size)
c = ctx.counters()
rows, idle, wide = (ctx.debug_counter(i) for i in (9, 10, 11))
print("jobs", len(jobs), "cells", c["dw_cells"], "blocks", c["dw_blocks"])
print("dual rows %d, with one idle half %.3f, with a second pass %.3f, cells per dual row %.1f" % (rows, idle / rows, wide / rows, c["dw_cells"] / rows))
