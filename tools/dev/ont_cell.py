"""development: kernel times of seeding one nanopore-mode grid cell at config-5 volume size (107 k reads x 20 kb = 2.14 Gbase)"""
import os, sys, time, numpy as np
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mecat_amd.hip as M
from mecat_amd import workload as W
n = int(os.environ.get("N", "100000"))
codes, lens = W.synth_reads(n, 20000, 0.12, int(1.3e9), 5, 1)
ctx = M.Context(0)
if os.environ.get("OFFDIAG"):      # an off-diagonal cell: the second half of the reads against the table of the first half (no bucket cuts)
    h = n // 2
    cut = int(lens[:h].astype(np.int64).sum())
    pac, offs, nb = W.pack_volume(codes[:cut], lens[:h])
    vol = M.Volume(ctx, pac, offs, nb, 0)
    pacq, offsq, nbq = W.pack_volume(codes[cut:], lens[h:])
    volq = M.Volume(ctx, pacq, offsq, nbq, h)
    n = n - h
else:
    pac, offs, nb = W.pack_volume(codes, lens)
    vol = M.Volume(ctx, pac, offs, nb, 0)
    volq = vol
print("bases", nb)
t0 = time.time(); idx = M.Index(ctx, vol); ctx.sync(); print("index %.1f ms" % ((time.time() - t0) * 1e3))
p = M.default_params(1)
dc = torch.zeros((n, p.maxc, 12), dtype=torch.int32, device="cuda"); dn = torch.zeros((n,), dtype=torch.int32, device="cuda")
for it in range(2):
    ctx.set_profiling(it == 1); ctx.reset_stats()
    t0 = time.time()
    M.seed_reads_strided_dev(ctx, idx, vol, volq, 0, 1, n, p, dc.data_ptr(), dn.data_ptr()); ctx.sync()
    print("seed %.1f ms" % ((time.time() - t0) * 1e3), "cands", int(dn.sum()), "hits", ctx.counters()["hits"], "walked", ctx.debug_counter(15), "wide filter kept", ctx.debug_counter(12), "(room)")
import hashlib
print("candidates sha256", hashlib.sha256(dc.cpu().numpy().tobytes()).hexdigest()[:16], "counts sha256", hashlib.sha256(dn.cpu().numpy().tobytes()).hexdigest()[:16])
for k, (c, ms) in sorted(ctx.kernel_stats().items(), key=lambda kv: -kv[1][1]):
    print("  %-16s %3d launches %8.2f ms" % (k, c, ms))
