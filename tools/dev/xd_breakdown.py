"""dev (GPU box): xd_extend_w's section clocks from the -DMECAT_XD_STATS build (`make xdstats`; MECAT_HIP_LIB=mecat_amd/lib/libmecat_hip_xdstats.so):
where a wave's life goes — staging, row loop, traceback — and the event counts behind it.  ONT-style candidates (N reads x 10 kb)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mecat_amd import hip as M, workload as W
n = int(os.environ.get("N", "5000"))
if os.environ.get("PB"):      # PacBio-style candidates (bench.py's xdrop_extend extra runs the X-drop aligner on config 2's): 15 kb @ 15 %, 30x
    codes, lens = W.synth_reads(n, 15000, 0.15, n * 15000 // 30, 2, 0)
else:
    codes, lens = W.synth_reads(n, 10000, 0.12, int(1_700_000 * n / 5000), 7, 1)
pac, offs, nb = W.pack_volume(codes, lens)
ctx = M.Context(0); vol = M.Volume(ctx, pac, offs, nb, 0); idx = M.Index(ctx, vol)
p = M.default_params(0 if os.environ.get("PB") else 1)
cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, len(lens), p)
jobs = W.jobs_from_candidates(cands, cnt, 0)
M.align_candidates(ctx, vol, vol, jobs, p.min_align_size, tech=1)
ctx.set_profiling(True); ctx.reset_stats()
t0 = time.time(); res = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size, tech=1); dt = time.time() - t0
ks = ctx.kernel_stats(); c = ctx.counters()
d = {s: ctx.debug_counter(s) for s in range(9, 25)}; d[9] = ctx.debug_counter(32)
life, stage, rows, trace, rows2, nwin, nsteps, nq, waves = d[16], d[17], d[18], d[19], d[20], d[21], d[22], d[23], max(1, d[24])
nrows, blocks, cells = d[9], c["dw_blocks"], c["dw_cells"]
print("jobs %d, %.1f ms (%s)" % (len(jobs), dt * 1e3, ", ".join("%s %.1f" % (k, v[1]) for k, v in ks.items() if k.startswith("xd"))))
print("waves %d, blocks %d, rows %d (%.1f per block), cells/row %.1f, two-pass rows %.1f %%" % (waves, blocks, nrows, nrows / max(1, blocks), cells / max(1, nrows), 100.0 * rows2 / max(1, nrows)))
print("wave life %.0f ticks; stage %.1f %%, rows %.1f %%, trace %.1f %%, rest %.1f %%" % (life / waves, 100 * stage / life, 100 * rows / life, 100 * trace / life, 100 * (life - stage - rows - trace) / life))
print("per block: stage %.0f ticks, rows %.0f (%.1f per row), trace %.0f (%.1f window loads, %.1f steps: %.0f ticks per step incl. loads), query refills %.1f" % (
    stage / blocks, rows / blocks, rows / max(1, nrows), trace / blocks, nwin / blocks, nsteps / blocks, trace / max(1, nsteps), nq / blocks))
