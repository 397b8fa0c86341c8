"""Dev helper (GPU box): time the dw extension on config 2's candidates (first NQ query reads) and print a checksum of the results.
python tools/dev/dw_time.py   [env NQ=100000, MECAT_HIP_LIB=<variant .so>]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mecat_amd.hip as M            # noqa: E402
from mecat_amd import workload as W  # noqa: E402

n, L, err, G, seed, ont = W.CONFIGS["config2"]
nq = int(os.environ.get("NQ", str(n)))
codes, lens = W.synth_reads(n, L, err, G, seed, ont)
pac, offs, nb = W.pack_volume(codes, lens)
dev = torch.device("cuda", 0)
ctx = M.Context(0)
vol = M.Volume(ctx, pac, offs, nb, 0)
idx = M.Index(ctx, vol)
p = M.default_params(0)
dc = torch.zeros((nq, p.maxc, 12), dtype=torch.int32, device=dev)
dn = torch.zeros((nq,), dtype=torch.int32, device=dev)
dj = torch.empty((nq * p.maxc, 5), dtype=torch.int32, device=dev)
dr = torch.empty((nq * p.maxc, 8), dtype=torch.int32, device=dev)
M.seed_reads_dev(ctx, idx, vol, vol, 0, nq, p, dc.data_ptr(), dn.data_ptr())
nj = M.jobs_from_candidates_dev(ctx, dc.data_ptr(), dn.data_ptr(), nq, p.maxc, 0, 1, 0, 0, 1, dj.data_ptr())
for it in range(2):
    ctx.set_profiling(it == 1)
    ctx.reset_stats()
    t0 = time.time()
    M.align_candidates_dev(ctx, vol, vol, dj.data_ptr(), nj, p.min_align_size, dr.data_ptr())
    ctx.sync()
    print("align %.1f ms, %d jobs" % ((time.time() - t0) * 1e3, nj))
for k, (c, ms) in sorted(ctx.kernel_stats().items(), key=lambda kv: -kv[1][1]):
    print("  %-16s %3d launches %8.2f ms" % (k, c, ms))
r = dr[:nj].to(torch.int64)
w = torch.arange(1, 9, dtype=torch.int64, device=dev)
print("handed over", ctx.debug_counter(8), "blocks", ctx.debug_counter(3), "cells", ctx.debug_counter(4),
      "checksum", int(((r * w).sum(dim=1) % 1000003).sum()), "ok", int(r[:, 7].sum()))
