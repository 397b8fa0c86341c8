import os, sys, subprocess, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) < 2:
    env = dict(os.environ, MECAT_SEED_STATS="1")
    p = subprocess.run([sys.executable, __file__, "child"], env=env, stderr=subprocess.PIPE, text=True)
    rows = np.array([[int(x) for x in l.split()[1:]] for l in p.stderr.splitlines() if l.startswith("SEEDSTAT")])
    print("strands", len(rows))
    names = ["hits_all", "kept", "nseg", "nrec", "ngated"]
    for st in (0, 1):
        r = rows[rows[:, 0] == st]
        print("strand", st, len(r))
        for i, nm in enumerate(names):
            v = r[:, i + 1]
            print("  %-9s mean %9.1f  p50 %7d p90 %7d p99 %7d p99.9 %7d max %8d  sum %.3e" % (nm, v.mean(), *np.percentile(v, [50, 90, 99, 99.9]).astype(int), v.max(), v.sum()))
    k = rows[:, 2]
    for cap in (4096, 6144, 8192, 12288, 16384):
        print("kept <= %d: %.4f of strands, %.4f of kept hits" % (cap, (k <= cap).mean(), k[k <= cap].sum() / k.sum()))
    sys.exit(0)
import mecat_amd.hip as M
from mecat_amd import workload as W
n = int(os.environ.get("N", "100000"))
codes, lens = W.synth_reads(n, 15000, 0.15, int(5e7 * n / 1e5), 2, 0)
pac, offs, nb = W.pack_volume(codes, lens)
ctx = M.Context(0); vol = M.Volume(ctx, pac, offs, nb, 0); idx = M.Index(ctx, vol); p = M.default_params(0)
cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, min(n, int(os.environ.get("NQ", "20000"))), p)
print("cands", int(cnt.sum()))
