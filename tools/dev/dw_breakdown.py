"""Dev helper (GPU box): dw_extend2's section clocks and row statistics from the -DMECAT_DW_STATS build of the library
(`make dwstats`; run with MECAT_HIP_LIB=mecat_amd/lib/libmecat_hip_dwstats.so).  Prints a markdown table (profiles/rNN_dw_row_breakdown.md)."""
import os, sys, time, numpy as np
import torch
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mecat_amd.hip as M
from mecat_amd import workload as W
n = int(os.environ.get("N", "20000"))
codes, lens = W.synth_reads(n, 15000, 0.15, int(5e7 * n / 1e5), 2, 0)
pac, offs, nb = W.pack_volume(codes, lens)
ctx = M.Context(0); vol = M.Volume(ctx, pac, offs, nb, 0); idx = M.Index(ctx, vol); p = M.default_params(0)
cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, n, p)
jobs = W.jobs_from_candidates(cands, cnt, 0)
M.align_candidates(ctx, vol, vol, jobs, p.min_align_size)      # warm-up
ctx.reset_stats()
ctx.set_profiling(True)
res = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size)
ks = ctx.kernel_stats()
c = ctx.counters()
D = [ctx.debug_counter(i) for i in range(64)]
rows, idle, wide = D[9], D[10], D[11]
life, t_setup, t_rows, t_ended, t_trace, t_acct = D[16:22]
n_outer, n_setup, n_ended, n_pass3, waves, snake2 = D[22:28]
nh, q16 = D[32:40], D[40:46]
print("## dw_extend2 breakdown (%d reads x 15 kb @ 15 %%, %d jobs, %d blocks, %d cells; kernel %.1f ms)" % (n, len(jobs), c["dw_blocks"], c["dw_cells"], ks["dw_extend2"][1]))
print()
print("| section of the wave's life (s_memrealtime ticks summed over %d waves) | share |" % waves)
print("|---|---|")
tot = t_setup + t_rows + t_ended + t_trace + t_acct
for name, v in (("unit pull + block setup + staging", t_setup), ("row loop", t_rows), ("end-of-block row (lowest end diagonal + band from the ring)", t_ended),
                ("tail traceback", t_trace), ("block accounting / result store", t_acct)):
    print("| %s | %.1f %% |" % (name, 100.0 * v / tot))
print("| (sections / wave life) | %.3f |" % (tot / life))
print()
print("dual rows %d (%.1f per outer iteration, %d outer iterations, %.1f %% of them with a set-up, %.1f %% ended a block)" % (rows, rows / n_outer, n_outer, 100.0 * n_setup / n_outer, 100.0 * n_ended / n_outer))
print("dual rows with one idle half %.1f %%, with a second pass %.1f %%, with three or more %.2f %%; cells per dual row %.1f; extra snake steps per pass %.4f"
      % (100.0 * idle / rows, 100.0 * wide / rows, 100.0 * n_pass3 / rows, c["dw_cells"] / rows, snake2 / max(1, rows)))
ur = sum(nh)
print()
print("| band width of a unit row | <= 8 | <= 16 | <= 24 | <= 32 | <= 48 | <= 64 | <= 96 | more |")
print("|---|---|---|---|---|---|---|---|---|")
print("| share of %d unit rows | " % ur + " | ".join("%.1f %%" % (100.0 * v / ur) for v in nh) + " |")
print()
print("| ceil(band / 16) of a unit row | 1 | 2 | 3 | 4 | 5-6 | more |")
print("|---|---|---|---|---|---|---|")
print("| share | " + " | ".join("%.1f %%" % (100.0 * v / ur) for v in q16) + " |")
