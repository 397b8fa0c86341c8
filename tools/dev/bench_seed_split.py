"""Dev experiment (not a test): do two seeding calls on disjoint read ranges overlap when issued from two contexts / streams?
python tools/dev/bench_seed_split.py"""
import os
import sys
import threading
import time

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
    h = n // 2

    def seed(ctx, lo, hi):
        M.seed_reads_dev(ctx, idx, vol, vol, lo, hi, p, dc[lo:].data_ptr(), dn[lo:].data_ptr())
        ctx.sync()

    for _ in range(2):
        t0 = time.perf_counter(); seed(ca, 0, n); t_all = time.perf_counter() - t0
        ref = dn.clone()
        t0 = time.perf_counter(); seed(ca, 0, h); seed(ca, h, n); t_seq = time.perf_counter() - t0
        th = [threading.Thread(target=seed, args=(ca, 0, h)), threading.Thread(target=seed, args=(cb, h, n))]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        t_con = time.perf_counter() - t0
        print("seed all %.0f ms; halves sequential %.0f ms; halves concurrent %.0f ms; same counts: %s"
              % (t_all * 1e3, t_seq * 1e3, t_con * 1e3, bool((ref == dn).all().item())), flush=True)


if __name__ == "__main__":
    main()
