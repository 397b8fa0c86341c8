#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one round into the summaries kept under profiles/.

    python profiles/summarize.py r01 gpurun_out/prof_r01c gpurun_out/prof_r01c_FETCH_SIZE gpurun_out/prof_r01c_WRITE_SIZE 2

argv: round tag, kernel-trace/stats dir, FETCH_SIZE pmc dir, WRITE_SIZE pmc dir, hot-path passes in each PMC run
(warm-up + steps).  Writes <tag>_kernel_stats.csv (copy), <tag>_hbm_counters.md, <tag>_hbm_traffic.json.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    if n.startswith("xd_extend_w<true>"):          # the overflow launch of the X-drop kernel: the library's own name for it (LAUNCH)
        return "xd_extend_wide"
    return n.split("(")[0].split("<")[0][:60]


def main():
    tag, kdir, fdir, wdir, passes = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
    here = os.environ.get("PROFILES_OUT") or os.path.dirname(os.path.abspath(__file__))
    os.makedirs(here, exist_ok=True)
    cmd = os.environ.get("PROFILE_CMD", "python bench.py --steps 1 --warmup 1 --no-cpu --no-extras --no-e2e")
    label = os.environ.get("PROFILE_LABEL", "config 2")
    # the run may hold several traced processes (bench.py calls the valu_peak calibration binary): keep the one with the hot path
    def pick(pattern):
        c = [f for f in glob.glob(os.path.join(kdir, "**", pattern), recursive=True)]
        c.sort(key=lambda f: -os.path.getsize(f))
        for f in c:
            if pattern.endswith("agent_info.csv") or "dw_extend" in open(f).read() or "xd_extend" in open(f).read():
                return f
        return c[0] if c else None
    for f in ("kernel_stats", "agent_info"):
        src = pick("*_%s.csv" % f)
        if src:
            shutil.copy(src, os.path.join(here, "%s_%s.csv" % (tag, f)))
    tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for d, col in ((fdir, 1), (wdir, 2)):
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(fn)):
                k = short(r["Kernel_Name"])
                if k.startswith("k_"):
                    continue          # valu_peak's calibration kernels
                if col == 1:
                    tot[k][0] += 1
                tot[k][col] += float(r["Counter_Value"])
    # average launch duration per kernel from the kernel-trace stats of the same command
    dur = {}
    ks = pick("*_kernel_stats.csv")
    if ks:
        for r in csv.DictReader(open(ks)):
            dur[short(r["Name"])] = float(r["AverageNs"])
    rows = sorted(tot.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))
    traffic = {}
    with open(os.path.join(here, "%s_hbm_counters.md" % tag), "w") as f:
        f.write("# %s - HBM traffic counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)\n\n" % tag)
        f.write("Command: `%s` (%s), i.e. %d passes of the hot path per run.\n" % (cmd, label, passes))
        f.write("Counter unit is KiB; bytes = value x 1024.  MI355X_MICROARCH.md (HBM): FETCH_SIZE reports half the bytes of a\n"
                "16 B/lane coalesced streaming read and is uncalibrated for other widths; none of these kernels issues 16 B/lane\n"
                "streams (4-8 B/lane gathers and scatters), so the values are reported uncorrected.  Infinity-Cache hits count.\n\n")
        f.write("Achieved = (fetch + write) per launch / average launch duration of the kernel-trace run; peak 8 TB/s (6.3 TB/s achievable).\n\n")
        f.write("| kernel | launches | fetch GB / pass | write GB / pass | fetch+write MB / launch | avg launch ms | achieved TB/s | of 8 TB/s |\n|---|---|---|---|---|---|---|---|\n")
        for k, (n, fk, wk) in rows:
            if "at::" in k or "rocclr" in k or "rocprim" in k.lower():
                continue
            fb, wb = fk * 1024.0, wk * 1024.0
            traffic[k] = {"launches_per_pass": n / passes, "fetch_bytes_per_launch": fb / max(n, 1), "write_bytes_per_launch": wb / max(n, 1)}
            ms = dur.get(k, 0.0) / 1e6
            tbs = ((fb + wb) / max(n, 1)) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            traffic[k]["avg_launch_ms"] = ms
            f.write("| %s | %d | %.2f | %.2f | %.1f | %.3f | %.2f | %.0f %% |\n" % (k, n, fb / passes / 1e9, wb / passes / 1e9, (fb + wb) / max(n, 1) / 1e6, ms,
                                                                  tbs, 100.0 * tbs / 8.0))
    json.dump(traffic, open(os.path.join(here, "%s_hbm_traffic.json" % tag), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
