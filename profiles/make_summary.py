#!/usr/bin/env python3
"""profiles/make_summary.py <round> — one page with the round's measured bench lines (reads profiles/<round>_*.json, writes
profiles/<round>_summary.md).  Nothing is measured here: the page quotes the committed stdout lines of bench.py."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"

def load(name):
    with open(os.path.join(HERE, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])

c2 = load(R + "_bench_config2.json"); c3 = load(R + "_bench_config3.json")
c5 = load(R + "_bench_config5.json"); cc = load(R + "_config5_cell_bench.json")
src = json.load(open(os.path.join(HERE, R + "_pmc_source.json")))
out = ["# %s — the round's measured lines in one place\n" % R,
       "All on one MI355X (gpurun box), synthetic inputs as SURVEY.md §8d prescribes, volumes resident in HBM when the clock starts.  Every JSON\n"
       "file named here is the unedited stdout line of `bench.py`; the rocprofv3 passes were taken at kernel-source digest `%s`.\n" % src["kernel_source_digest"],
       "| workload (`bench.py --workload`) | file | kernel sources | steps | candidates | ms per pass | index / seed / extend ms | candidates/s | aligned Gbase/s | checked against the reference inside the run |",
       "|---|---|---|---|---|---|---|---|---|---|"]

def row(name, f, d):
    ph = d["phase_ms"]
    par = d.get("parity_vs_reference")
    fs = d.get("cpu_baseline", {}).get("full_size_same_host")
    if par: chk = "%d cell hash(es) of the unmodified `mecat2pw -j 0`: %s" % (par["cells_checked"], "identical" if par["identical"] else "DIFFERENT")
    elif fs: chk = "candidate count of the full-size reference run on the same host: %s" % ("same" if fs.get("same_count_as_gpu") else "DIFFERENT")
    else: chk = "-"
    out.append("| %s | `%s` | `%s` | %d | %d | %.1f | %.1f / %.1f / %.1f | %.3g | %.2f | %s |" % (
        name, f, d["roofline"].get("kernel_source_digest"), d["steps"], d["candidates"], d["ms_per_step"], ph["index"], ph["seed"], ph["align"],
        d["value"], d["aligned_gbase_per_s"], chk))

row("config2 (default; BASELINE configs[1])", R + "_bench_config2.json", c2)
row("config3 (configs[2]: 3 volumes, 6 cells)", R + "_bench_config3.json", c3)
row("config5_cell (cell (0,1) of configs[4], `-x 1`)", R + "_config5_cell_bench.json", cc)
row("config5 (configs[4] whole: 19 volumes, 190 cells, `-x 1`)", R + "_bench_config5.json", c5)
c4name = os.path.join(HERE, R + "_bench_config4.json")
if os.path.exists(c4name):
    c4 = load(R + "_bench_config4.json")
    out.append("")
    out.append("Config 4 (`--workload config4`, BASELINE configs[3] up to the consensus table; `%s_bench_config4.json`): %d templates (%.2f Gbase), %d re-alignments, "
               "%d accepted, %.2f s per pass = %.3g template bases/s, %.0f templates/s; device kernels %.0f ms of it (`cns_forward` %.0f, `cns_trace` %.0f, "
               "`cns_extend` %.0f; strings: `cns_push_gaps` + `cns_strings_build` %.0f); the unmodified `consensus_one_read_can_pacbio` on the same host (%d processes, CPU quota %s cores): %.3g template bases/s." % (
                   R, c4["config"]["templates"], c4["config"]["template_bases"] / 1e9, c4["alignments_per_step"], c4["accepted_per_step"], c4["ms_per_step"] / 1e3,
                   c4["value"], c4["templates_per_s"], c4["gpu_kernel_ms_per_step"], c4["kernel_ms_per_step"].get("cns_forward", 0), c4["kernel_ms_per_step"].get("cns_trace", 0),
                   c4["kernel_ms_per_step"].get("cns_extend", 0), c4["kernel_ms_per_step"].get("cns_push_gaps", 0) + c4["kernel_ms_per_step"].get("cns_strings_build", 0), c4["cpu_baseline"]["cores"], c4["cpu_baseline"].get("cpu_quota_cores"), c4["cpu_baseline"]["value"]))
out.append("")
if c5["roofline"].get("kernel_source_digest") != cc["roofline"].get("kernel_source_digest"):
    out.append("(The whole-config-5 line takes six minutes and was taken one commit before the others: the sources differ in the record-pool size of "
               "`asm_seed`, a kernel this workload does not run.)\n")
b2, bc = c2["cpu_baseline"], cc.get("cpu_baseline", c5["cpu_baseline"])
fs = b2["full_size_same_host"]
out.append("CPU side, same runs.  Config 2: the unmodified `mecat2pw -j 0` on the full FASTA with %d threads of the box's %d CPUs — %.0f s hot path "
           "(%.0f s of it `create_ref_index`), %d candidates (the GPU's count), %.0f candidates/s: the GPU line is %.0f x that; on the %s: %.0f candidates/s "
           "`-j 0`, %.0f overlaps/s `-j 1`.  Config-5 cell (%s): %.0f candidates/s `-j 0`, %.0f overlaps/s `-j 1`.\n" % (
               fs["cores"], fs["host_cpus"], fs["hot_path_s"], fs["create_ref_index_s"], fs["candidates"], fs["candidates_per_s"], b2["gpu_over_cpu_full_size_j0"],
               b2["sample"], b2["value"], b2["j1_overlaps_per_s"], bc["sample"], bc["value"], bc["j1_overlaps_per_s"]))
x = c2
out.append("Extras of the config-2 line: `xdrop_extend` %.0f k alignments/s, `cns_realign` %.2f M alignments/s, `cns_accept` %.0f templates/s "
           "(parity with the unmodified accept loop: %s), `asm_overlap` %.2f s for %.0f Mbases (%.1f x the unmodified tool on the same host, identical "
           "output: %s), `e2e` -j 0 %.2f s / -j 1 %.2f s.\n" % (
               x["xdrop_extend"]["alignments_per_s"] / 1e3, x["cns_realign"]["alignments_per_s"] / 1e6, x["cns_accept"]["templates_per_s"],
               x["cns_accept"].get("parity_vs_reference", {}).get("identical"), x["asm_overlap"]["seconds"], x["asm_overlap"].get("bases", 0) / 1e6,
               x["asm_overlap"].get("speedup_vs_reference_threads", 0), x["asm_overlap"].get("cpu_baseline", {}).get("identical_output"),
               x["e2e"]["j0"]["wall_s"], x["e2e"]["j1"]["wall_s"]))
r2, rc = c2["roofline"], cc["roofline"]
out.append("Roofline blocks.  Config 2 — dominant kernel `%s`: %.1f GB/s of algorithmic bytes = %.4f of the HBM peak, measured HBM traffic %.2f GB per launch "
           "against %.2f GB algorithmic, VALU issue %.2f of the data-sheet ceiling of the kernel's instruction mix.  Config-5 cell — `%s`: %.4f of the HBM peak, "
           "measured traffic %.0f GB per launch (the script cells of every DP cell, written once and read back by the traceback: DESIGN.md §3.4), %.3g DP cells/s; "
           "phases index / seed / extend at %.3f / %.3f / %.5f of the HBM peak.\n" % (
               r2["kernel"], r2["achieved"], r2["frac"], (r2["traffic"] or 0) / 1e9, r2["algorithmic_bytes_per_launch"] / 1e9, r2.get("valu_issue", {}).get("frac", float("nan")),
               rc["kernel"], rc["frac"], (rc["traffic"] or 0) / 1e9, rc["xdrop"]["dp_cells_per_s"],
               rc["phases"]["index"]["frac"], rc["phases"]["seed"]["frac"], rc["phases"]["align"]["frac"]))
out.append("rocprofv3 summaries: `%s_kernel_stats.csv`, `%s_hbm_counters.md`, `%s_instruction_mix.md` (config 2); `%s_config5_cell_*` (the nanopore chain: "
           "`seed_filter_wide`, `seed_emit`, `seed_sort_pass`, `seed_build`, `xd_extend_w`); `%s_cns_kernels.md`, `%s_asm_kernels.md` (`cns_extend`, `asm_seed`, "
           "`asm_extend`)." % ((R,) * 6)
           + ("  `%s_xd_breakdown.md`: what bounds the X-drop kernel and what the rebuild changed." % R if os.path.exists(os.path.join(HERE, R + "_xd_breakdown.md")) else "")
           + ("  `%s_dw_attempt.md`: the round's bounded attempt on `dw_extend2`." % R if os.path.exists(os.path.join(HERE, R + "_dw_attempt.md")) else "")
           + ("  `%s_parity_sweeps.md`: the randomised sweeps against the oracle." % R if os.path.exists(os.path.join(HERE, R + "_parity_sweeps.md")) else ""))
# round 6 additions: the end-to-end legs of the multi-volume configs and the simulated multi-GPU shares, when their files exist
def maybe(name):
    try:
        return load(name)
    except (OSError, ValueError, IndexError):
        try:
            return json.load(open(os.path.join(HERE, name)))
        except (OSError, ValueError):
            return None
e3, e5 = maybe(R + "_e2e_config3.json"), maybe(R + "_e2e_config5.json")
if e3 or e5:
    out.append("")
    out.append("End to end through the drop-in binary (`bench.py --workload … --e2e`; FASTA in /dev/shm, stages from the driver's own timers):\n")
    out.append("| workload | task | wall s | lines | hot path s (index + seed + extend) | split s | largest stage outside the hot path |")
    out.append("|---|---|---|---|---|---|---|")
    for nm, e in (("config3", e3), ("config5", e5)):
        if not e:
            continue
        for key, task in (("j0", "-j 0"), ("j1", "-j 1 -g 1")):
            r = e["e2e"].get(key)
            if r and "wall_s" in r:
                big = list(r["largest_stage_outside_the_hot_path"].items())[0]
                out.append("| %s | `%s` | %.1f | %d | %.1f | %.2f | %s %.2f s |" % (nm, task, r["wall_s"], r["lines"], r["hot_path_s"], r["stages_s"]["split_raw_dataset"], big[0], big[1]))
sim = maybe(R + "_simulated_scaling.json")
if sim:
    out.append("")
    out.append("Simulated multi-GPU shares (`bench.py --simulate-ranks`: every rank's share of the library's sharded calls alone on one GPU, no transport; "
               "`%s_simulated_scaling.json`): compute-only bound T(1) / slowest rank, index rebuilt on every rank / built in key-range shards:\n" % R)
    out.append("| workload | P | slowest rank ms (rebuilt / sharded) | bound (rebuilt / sharded) | imbalance max/mean (seed, align) |")
    out.append("|---|---|---|---|---|")
    for wl, d in sim.items():
        if not isinstance(d, dict) or "P" not in d:
            continue
        for P, v in d["P"].items():
            if "slowest_rank_ms" not in v:
                continue
            sr, cb, im = v["slowest_rank_ms"], v["compute_only_speedup_bound"], v["imbalance_max_over_mean"]
            out.append("| %s | %s | %.1f / %.1f of %.1f | %.2f / %.2f | %.3f, %.3f |" % (wl, P, sr["index_rebuilt_on_every_rank"], sr["index_in_key_range_shards"], d["one_gpu_step_ms"],
                       cb["index_rebuilt_on_every_rank"], cb["index_in_key_range_shards"], im["seed"], im["align"] or 0))
    rm = c5.get("simulated_rows_mode")
    if rm:
        out.append("")
        out.append("Config 5 in rows mode, from the row times of the whole-config run above (`mhip_shard_deal_rows`, nothing moves): " + "; ".join(
            "P = %s: bound %.2f x (imbalance %.3f)" % (P, v["compute_only_speedup_bound"], v["imbalance_max_over_mean"]) for P, v in rm["P"].items()) + ".")
with open(os.path.join(HERE, R + "_summary.md"), "w") as f:
    f.write("\n".join(out) + "\n")
print("\n".join(out))
