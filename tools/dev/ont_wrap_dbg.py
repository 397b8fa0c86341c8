import os, sys, numpy as np
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mecat_amd.hip as M
from mecat_amd import workload as W
n = 3000
codes, lens = W.synth_reads(n, 20000, 0.12, int(1.3e9 * n / 1e5), 5, 1)
pac, offs, nb = W.pack_volume(codes, lens)
ctx = M.Context(0); vol = M.Volume(ctx, pac, offs, nb, 0); idx = M.Index(ctx, vol)
counts, offsets = idx.download()
starts = np.concatenate([[0], np.cumsum(counts, dtype=np.int64)])
rs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
worst = []
for r in range(0, 60):
    read = codes[rs[r]: rs[r + 1]].astype(np.int64)
    for strand in (0, 1):
        s = read if strand == 0 else (3 - read)[::-1]
        K = (len(s) - 13) // 10 + 1
        pos_all = []
        for km in range(K):
            kid = 0
            for c in s[km * 10: km * 10 + 13]:
                kid = (kid << 2) | int(c)
            p = offsets[starts[kid]: starts[kid + 1]]
            if strand == 0:
                p = p[p != offs[r, 0] + km * 10]
            pos_all.append(p)
        pos_all = np.concatenate(pos_all)
        seg = pos_all // 2000
        u, c = np.unique(seg, return_counts=True)
        m = c.argmax()
        worst.append((int(c[m]), r, strand, int(u[m]), len(pos_all)))
worst.sort(reverse=True)
print("volume segments", nb // 2000, "own region of read 0:", offs[0, 0] // 2000, (offs[0, 0] + offs[0, 1]) // 2000)
for w in worst[:12]:
    cnt, r, strand, seg, tot = w
    # which read owns that segment
    owner = int(np.searchsorted(offs[:, 0], seg * 2000, side="right") - 1)
    print("max %3d hits in one segment: read %d strand %d, segment %d (read %d's area), %d hits in all" % (cnt, r, strand, seg, owner, tot))
print("strands with a segment of >= 16 hits:", sum(1 for w in worst if w[0] >= 16), "of", len(worst))
