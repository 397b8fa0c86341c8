"""dev: time of the X-drop extension on the ONT-style candidates of tools/dev/bench_xdrop.py (5000 x 10 kb), per library variant / waves"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mecat_amd import hip as M, workload as W
n = int(os.environ.get("N", "5000"))
codes, lens = W.synth_reads(n, 10000, 0.12, int(1_700_000 * n / 5000), 7, 1)
pac, offs, nb = W.pack_volume(codes, lens)
ctx = M.Context(0); vol = M.Volume(ctx, pac, offs, nb, 0); idx = M.Index(ctx, vol)
p = M.default_params(1)
cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, len(lens), p)
jobs = W.jobs_from_candidates(cands, cnt, 0)
for it in range(3):
    ctx.set_profiling(True); ctx.reset_stats()
    t0 = time.time(); res = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size, tech=1); dt = time.time() - t0
    ks = ctx.kernel_stats()
ok = res["ok"] != 0
print("%s waves=%s: %d jobs %.1f ms (%s)  ok %d  %.0f k/s  sum %d" % (os.path.basename(M.lib_path()), os.environ.get("MECAT_XW_WAVES", "24"), len(jobs), dt * 1e3,
      ", ".join("%s %.1f" % (k, v[1]) for k, v in ks.items() if k.startswith("xd")), ok.sum(), len(jobs) / dt / 1e3,
      int(res["query_end"].astype(np.int64).sum() + res["target_end"].sum() + res["matches"].sum())))
