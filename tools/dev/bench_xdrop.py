import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))      # tests/helpers.py
import torch
from mecat_amd import hip as M, workload as W
codes, lens = W.synth_reads(5000, 10000, 0.12, 1_700_000, 7, 1)
pac, offs, nb = W.pack_volume(codes, lens)
ctx = M.Context(0)
vol = M.Volume(ctx, pac, offs, nb, 0)
idx = M.Index(ctx, vol)
p = M.default_params(1)
t0 = time.time(); cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, len(lens), p); t1 = time.time()
jobs = W.jobs_from_candidates(cands, cnt, 0)
t2 = time.time(); res = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size, tech=1); t3 = time.time()
ok = res["ok"] != 0
print("ONT 5000x10kb: cands %d seed %.3fs ; xalign %d jobs %.3fs ok %d aligned %.3f Gbase -> %.3f Gbase/s" % (cnt.sum(), t1-t0, len(jobs), t3-t2, ok.sum(), (res["query_end"]-res["query_start"])[ok].sum()/1e9, (res["query_end"]-res["query_start"])[ok].sum()/1e9/(t3-t2)))


t2 = time.time(); res2 = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size, tech=0); t3 = time.time()
print("same jobs with dw: %.3fs" % (t3-t2))
