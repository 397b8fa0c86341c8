import os, sys, time, numpy as np
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mecat_amd.hip as M
from mecat_amd import workload as W
n = int(os.environ.get("N", "100000"))
codes, lens = W.synth_reads(n, 15000, 0.15, int(5e7 * n / 1e5), 2, 0)
pac, offs, nb = W.pack_volume(codes, lens)
ctx = M.Context(0); vol = M.Volume(ctx, pac, offs, nb, 0); idx = M.Index(ctx, vol); p = M.default_params(0)
import torch
dc = torch.zeros((n, p.maxc, 12), dtype=torch.int32, device="cuda"); dn = torch.zeros((n,), dtype=torch.int32, device="cuda")
for it in range(2):
    ctx.set_profiling(it == 1); ctx.reset_stats()
    t0 = time.time()
    M.seed_reads_strided_dev(ctx, idx, vol, vol, 0, 1, n, p, dc.data_ptr(), dn.data_ptr()); ctx.sync()
    print("seed %.1f ms" % ((time.time() - t0) * 1e3), "cands", int(dn.sum()))
for k, (c, ms) in sorted(ctx.kernel_stats().items(), key=lambda kv: -kv[1][1]):
    print("  %-16s %3d launches %8.2f ms" % (k, c, ms))
print("fused", ctx.debug_counter(13), "chain", ctx.debug_counter(14))
if os.environ.get("CAND_STATS"):
    print("cand: gated iterations %d, reached the vote %d, maxval < 5: %d, entries voted on %d" % tuple(ctx.debug_counter(i) for i in (9, 10, 11, 12)))
