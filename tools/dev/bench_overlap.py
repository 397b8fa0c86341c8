"""Dev experiment (not a test): how well do seeding (latency/HBM bound) and dw extension (VALU bound) overlap when run
from two contexts on two streams?  python tools/dev/bench_overlap.py [dw_waves ...]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))      # tests/helpers.py
import mecat_amd.hip as M            # noqa: E402
from mecat_amd import workload as W  # noqa: E402


def main():
    n, L, err, G, seed, ont = W.CONFIGS["config2"]
    codes, lens = W.synth_reads(n, L, err, G, seed, ont)
    pac, offs, nb = W.pack_volume(codes, lens)
    dev = torch.device("cuda", 0)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ca, cb = M.Context(0, sa.cuda_stream), M.Context(0, sb.cuda_stream)
    vol = M.Volume(ca, pac, offs, nb, 0)
    idx = M.Index(ca, vol)
    p = M.default_params(0)
    dc = torch.zeros((n, p.maxc, 12), dtype=torch.int32, device=dev)
    dn = torch.zeros((n,), dtype=torch.int32, device=dev)
    dj = torch.empty((n * p.maxc, 5), dtype=torch.int32, device=dev)
    dr = torch.empty((n * p.maxc, 8), dtype=torch.int32, device=dev)

    def seed_all():
        M.seed_reads_dev(ca, idx, vol, vol, 0, n, p, dc.data_ptr(), dn.data_ptr())
        ca.sync()

    seed_all()
    nj = M.jobs_from_candidates_dev(cb, dc.data_ptr(), dn.data_ptr(), n, p.maxc, 0, 1, 0, 0, 1, dj.data_ptr())

    def align_all():
        M.align_candidates_dev(cb, vol, vol, dj.data_ptr(), nj, p.min_align_size, dr.data_ptr())
        cb.sync()

    for waves in [int(a) for a in sys.argv[1:]] or [16, 12, 8]:
        os.environ["MECAT_DW_WAVES"] = str(waves)
        align_all()
        t0 = time.perf_counter(); seed_all(); ts = time.perf_counter() - t0
        t0 = time.perf_counter(); align_all(); ta = time.perf_counter() - t0
        th = [threading.Thread(target=align_all), threading.Thread(target=seed_all)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        tb = time.perf_counter() - t0
        print("dw_waves=%d: seed %.0f ms, align %.0f ms, sum %.0f ms, concurrent %.0f ms" % (waves, ts * 1e3, ta * 1e3, (ts + ta) * 1e3, tb * 1e3), flush=True)


if __name__ == "__main__":
    main()
