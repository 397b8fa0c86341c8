"""Throughput of the mecat2cns re-aligner on config-2-style candidates, next to the unmodified
reference (oracle/_ref/libref_cns.so, one thread) on a sample of the same jobs.  Not part of bench.py's metric.
    python tools/dev/bench_cns.py [nreads]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))      # tests/helpers.py
import torch  # noqa: F401,E402  (before the HIP library, see conftest.py)
import helpers as H  # noqa: E402
from mecat_amd import hip as M, workload as W  # noqa: E402


def main():
    nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    L, err = 15000, 0.15
    codes, lens = W.synth_reads(nreads, L, err, nreads * L // 30, 2, 0)
    pac, offs, nb = W.pack_volume(codes, lens)
    ctx = M.Context(0)
    vol = M.Volume(ctx, pac, offs, nb, 0)
    idx = M.Index(ctx, vol)
    p = M.default_params(0)
    cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, nreads, p)
    jobs = W.jobs_from_candidates(cands, cnt, 0)
    cap = 49152
    # device-resident run (results and columns stay on the device, like bench.py's extension stage)
    dj = torch.from_numpy(jobs.view(np.int32).reshape(-1, 5).copy()).cuda()
    dres = torch.empty((len(jobs), 16), dtype=torch.int32, device="cuda")
    dops = torch.empty((len(jobs), 2, cap // 16), dtype=torch.int32, device="cuda")
    L_ = M.lib()
    for rep in range(2):
        ctx.set_profiling(True); ctx.reset_stats()
        c0 = ctx.counters()
        torch.cuda.synchronize()
        t0 = time.time()
        rc = L_.mhip_cns_align_candidates_dev(ctx.h, vol.h, vol.h, dj.data_ptr(), len(jobs), 0.15, 500, cap, dres.data_ptr(), dops.data_ptr())
        assert rc == 0, L_.mhip_last_error()
        torch.cuda.synchronize()
        dt = time.time() - t0
    print("cns counters:", {k: ctx.counters()[k] - c0[k] for k in ("dw_blocks", "dw_cells")})
    print("kernels (ms): " + ", ".join("%s %.1f" % (k, v[1]) for k, v in sorted(ctx.kernel_stats().items(), key=lambda kv: -kv[1][1])[:8]))
    ctx.reset_stats()
    M.align_candidates(ctx, vol, vol, jobs, 500)
    print("same jobs through mecat2pw's aligner (ms): " + ", ".join("%s %.1f" % (k, v[1]) for k, v in sorted(ctx.kernel_stats().items(), key=lambda kv: -kv[1][1])[:4]),
          {k: ctx.counters()[k] for k in ("dw_blocks", "dw_cells")})
    r = dres.cpu().numpy()
    ok = r[:, 0] != 0
    aligned = int((r[:, 2] - r[:, 1])[ok].sum())
    print("cns re-align: %d reads, %d jobs, %.3f s -> %.1f k alignments/s, %.2f aligned Gbase/s (ok %d)" %
          (nreads, len(jobs), dt, len(jobs) / dt / 1e3, aligned / 1e9 / dt, int(ok.sum())))
    if H.ref_cns_available():
        R = H.ref_cns()
        rng = np.random.default_rng(1)
        pick = rng.choice(len(jobs), size=min(400, len(jobs)), replace=False)
        starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
        res = np.zeros(5, np.int32)
        nb_ref, t0 = 0, time.time()
        for j in pick:
            jb = jobs[j]
            q = codes[starts[jb["qid_local"]]: starts[jb["qid_local"] + 1]].astype(np.int8)
            if jb["chain"]:
                q = (3 - q)[::-1].copy()
            t = codes[starts[jb["sid_local"]]: starts[jb["sid_local"] + 1]].astype(np.int8)
            okr = R.refc_get_alignment(q.ctypes.data, int(jb["qstart"]), len(q), t.ctypes.data, int(jb["sstart"]), len(t), 0.15, 500, res.ctypes.data, None, None)
            assert okr == int(r[j, 0]) and (not okr or tuple(res[:4]) == tuple(r[j, 1:5]))
            nb_ref += int(res[1] - res[0]) if okr else 0
        dtr = time.time() - t0
        print("reference (1 thread, %d of the same jobs, results identical): %.2f k alignments/s, %.4f aligned Gbase/s" %
              (len(pick), len(pick) / dtr / 1e3, nb_ref / 1e9 / dtr))


if __name__ == "__main__":
    main()
