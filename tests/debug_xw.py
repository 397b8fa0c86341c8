"""Dev script: wave X-drop kernel vs lane kernel on the tiny ONT set, job by job."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import helpers as H
from mecat_amd import hip as M, workload as W
G = json.load(open(os.path.join(H.GOLDEN, "golden.json")))
g = G["sets"]["tiny_ont"]["gen"]
codes, lens = H.synth_reads(g["nreads"], g["L"], g["err"], g["genome"], g["seed"], g["ont"])
pac, offs, nb = W.pack_volume(codes, lens)
ctx = M.Context(0)
vol = M.Volume(ctx, pac, offs, nb, 0)
idx = M.Index(ctx, vol)
p = M.default_params(1)
cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, len(lens), p)
jobs = W.jobs_from_candidates(cands, cnt, 0)
os.environ["MECAT_XD_KERNEL"] = "1"
a = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size, tech=1).copy()
os.environ.pop("MECAT_XD_KERNEL", None)
b = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size, tech=1).copy()
import ctypes
bad = [i for i in range(len(jobs)) if a[i].tobytes() != b[i].tobytes()]
print(len(jobs), "jobs,", len(bad), "differ")
for i in bad[:6]:
    print(jobs[i], "qlen", lens[jobs[i]["qid_local"]], "tlen", lens[jobs[i]["sid_local"]]); print("  lane", a[i]); print("  wave", b[i])
# repeat: is the wave kernel deterministic?
c = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size, tech=1).copy()
print("wave kernel run-to-run differences:", sum(1 for i in range(len(jobs)) if c[i].tobytes() != b[i].tobytes()))
