import os, sys, time, numpy as np
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mecat_amd.hip as M
from mecat_amd import workload as W
n = int(os.environ.get("N", "100000"))
if os.environ.get("ONT"):      # a nanopore-mode volume at the size limit (config 5: 107 k reads x 20 kb = 2.14 Gbases)
    codes, lens = W.synth_reads(n, 20000, 0.12, int(1.3e9), 5, 1)
else:
    codes, lens = W.synth_reads(n, 15000, 0.15, int(5e7), 2, 0)
pac, offs, nb = W.pack_volume(codes, lens)
ctx = M.Context(0); vol = M.Volume(ctx, pac, offs, nb, 0)
for it in range(2):
    ctx.set_profiling(it == 1); ctx.reset_stats()
    try:
        idx = M.Index(ctx, vol); idx.free()
    except Exception as e:
        print("build failed (expected in the timing experiment):", str(e)[:80])
tot = sum(ms for k, (c, ms) in ctx.kernel_stats().items() if k.startswith("idx") or k.startswith("ix_")); print("index kernels total %.2f ms" % tot)
for k, (c, ms) in sorted(ctx.kernel_stats().items(), key=lambda kv: -kv[1][1]):
    if k.startswith("idx") or k.startswith("ix_"): print("  %-16s %3d launches %8.2f ms" % (k, c, ms))
