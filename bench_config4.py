"""bench.py --workload config4: BASELINE.json's configs[3], "mecat2cns consensus on config-2 overlaps: shared dw d-path HIP kernel,
1 x MI355X, corrected-bases/sec vs CPU mecat2cns" — the part of it SURVEY.md §8f row N1 puts on the device.

What a step is.  mecat2cns loads a partition of candidates, and for every template read (every read with >= 4 candidates, >= 4 750
bases) sorts the template's candidates by (score desc, qid, qext), re-aligns them one after the other with its O(ND) aligner (dw.cpp) —
at most 200, until 60 are accepted or the template is covered — and hands the accepted, gap-normalised alignments to the consensus
table (mecat_correction.cpp:388-450).  One step = that, for ALL templates of config 2 (100 000 reads x 15 kb @ 15 %, the candidates of
this repository's own -j 0 pass): mhip_cns_accept_templates sorts, re-aligns every offered candidate speculatively on the GPU (forward
rows + paths, cns_fwd.h), replays the accept decisions and rebuilds the normalised strings of the accepted alignments on the host
cores.  The consensus table and the POA vote behind it stay with mecat2cns (out of scope), so the rate is TEMPLATE bases
per second through this stage, not corrected bases per second.

Contract line as bench.py's: `roofline` for the dominant kernel (the re-aligner's forward pass; algorithmic bytes per job = the two
aligned spans at 2 bits a base + the columns at 2 bits + the 64-byte result record, counted on the device by cns_stitch), `cpu_baseline`
= the UNMODIFIED consensus_one_read_can_pacbio (oracle/_ref/libref_cns_accept.so) on every host core over the first >= 2 000
templates of the same records, same run.  One GPU (the stage has no exchange step: templates are independent, --gpus N would run
replicas)."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_leg(codes, lens, rec, tb, ids, cores):
    """the unmodified accept loop on `cores` processes, templates dealt round-robin; wall = the slowest process (reads loaded before)"""
    from mecat_amd import workload as W
    so = os.path.join(ROOT, "oracle", "_ref", "libref_cns_accept.so")
    if not os.path.exists(so):
        return {"error": "oracle/_ref/libref_cns_accept.so not built"}
    K = min(len(ids), max(2000, 100 * cores))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="mecat_cns4_", dir=base)
    try:
        fa = os.path.join(d, "reads.fa")
        W.write_fasta(fa, codes, lens)
        np.savez(os.path.join(d, "in.npz"), rec=np.ascontiguousarray(rec[: tb[K]]), tb=tb[: K + 1], ids=ids[:K], n=len(lens))
        bench = os.path.join(ROOT, "bench.py")
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, bench, "--cns-cpu-leg", d, str(k), str(cores)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                  env=dict(os.environ, OMP_NUM_THREADS="1")) for k in range(cores)]
        outs = []
        for p in procs:
            o, e = p.communicate(timeout=1500)
            if p.returncode != 0:
                return {"error": e[-300:]}
            outs.append(json.loads(o.strip().splitlines()[-1]))
        wall = time.time() - t0
        dt = max(o["seconds"] for o in outs)
        tbases = int(lens[ids[:K]].astype(np.int64).sum())
        return {"kind": "reference", "cores": cores, "templates": K, "candidates_offered": int(sum(o["candidates_offered"] for o in outs)),
                "accepted": int(sum(o["accepted"] for o in outs)), "seconds": dt, "wall_incl_loading_reads_s": wall,
                "value": tbases / dt, "unit": "template bases/s", "templates_per_s": K / dt,
                "sample": "first %d templates of the GPU leg's records, unmodified consensus_one_read_can_pacbio up to the consensus table "
                          "(oracle/_ref/libref_cns_accept.so), %d processes on this host's cores, templates dealt round-robin; the slowest "
                          "process's time counts, loading the reads (%.1f s per process) does not" % (K, cores, float(np.mean([o["load_reads_s"] for o in outs])))}
    finally:
        subprocess.run(["rm", "-rf", d])


def run(args):
    import torch
    from mecat_amd import hip as M, workload as W
    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("bench.py --workload config4 runs on one GPU (templates are independent: more GPUs would be replicas)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n, L, err, G, seed, ont = W.CONFIGS["config2"]
    t0 = time.time()
    codes, lens = W.synth_reads(n, L, err, G, seed, ont)
    pac, offs, num_bases = W.pack_volume(codes, lens)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = M.Context(0, stream.cuda_stream)
    vol = M.Volume(ctx, pac, offs, num_bases, 0)
    params = M.default_params(ont)
    # the candidates: this repository's own -j 0 pass over config 2 (index + seeding), as mecat2cns would find them in its partition files
    idx = M.Index(ctx, vol)
    cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, n, params)
    idx.free()
    ec = W.ext_candidates_from_table(cands, cnt, lens)
    rec, tb, ids = W.cns_templates(ec, n)
    T = len(ids)
    if os.environ.get("MECAT_BENCH_TEMPLATES"):
        T = min(T, int(os.environ["MECAT_BENCH_TEMPLATES"]))
    rec = np.ascontiguousarray(rec[: tb[T]])
    tb = tb[: T + 1]
    tbases = int(lens[ids[:T]].astype(np.int64).sum())
    threads = int(os.environ.get("MECAT_BENCH_THREADS", min(64, os.cpu_count() or 1)))
    log("[bench] config4: %d candidates of config 2 -> %d templates (%d records, %.2f Gbase of templates), set up in %.1f s" % (len(ec), T, len(rec), tbases / 1e9, time.time() - t0))
    del cands, ec

    def step():
        r = rec.copy()          # (the call sorts its records in place)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        acc, strs, nja = M.cns_accept_templates(ctx, vol, pac, r, tb, ont, params.min_align_size if ont else 2000, 0.4 if ont else 0.9, threads=threads)
        torch.cuda.synchronize()
        return time.perf_counter() - c0, len(acc), int(nja), len(strs)

    ctx.set_profiling(True)
    for _ in range(args.warmup):
        step()
    ctx.reset_stats()
    outs = [step() for _ in range(args.steps)]
    kstats = ctx.kernel_stats()
    counters = ctx.counters()
    alg_bytes = ctx.debug_counter(34) / args.steps
    ctx.set_profiling(False)
    dt = float(np.mean([o[0] for o in outs]))
    dom = max(((k, v) for k, v in kstats.items() if k.startswith("cns_")), key=lambda kv: kv[1][1])
    dname, (dl, dms) = dom
    avg_ms = dms / max(1, dl)
    launches_per_step = max(1, dl // args.steps)
    alg_per_launch = alg_bytes / launches_per_step
    achieved = alg_per_launch / 1e9 / (avg_ms / 1e3) if avg_ms > 0 else 0.0
    kern_ms = {k: v[1] / args.steps for k, v in sorted(kstats.items(), key=lambda kv: -kv[1][1])}
    gpu_ms = sum(kern_ms.values())
    line = {
        "metric": "template bases/sec through mecat2cns' candidate re-alignment + accept stage (config-2 overlaps)", "value": tbases / dt, "unit": "template bases/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "config4: mecat2cns' per-template stage (sort, <= 200 re-alignments per template, accept replay, normalised strings) on the %d "
                               "candidates of config 2 (%d reads x %d bp @ %.0f%% error, genome %d, seed %d): %d templates, every one of them"
                               % (int(cnt.sum()), n, L, err * 100, G, seed, T),
                   "templates": T, "template_bases": tbases, "candidate_records": int(len(rec)), "host_threads_for_accept_replay": threads},
        "templates_per_s": T / dt, "alignments_per_step": outs[-1][2], "alignments_per_s": outs[-1][2] / dt, "accepted_per_step": outs[-1][1],
        "aligned_string_bytes_per_step": outs[-1][3],
        "aligned_gbase_per_s_realign_kernels": counters["aligned_bases"] / args.steps / 1e9 / (gpu_ms / 1e3) if gpu_ms > 0 else None,
        "kernel_ms_per_step": kern_ms, "gpu_kernel_ms_per_step": gpu_ms,
        "roofline": {"bound": "hbm", "kernel": dname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_per_launch, "avg_launch_ms": avg_ms, "launches": dl,
                     "note": "algorithmic bytes of a re-alignment = (query span + target span) / 4 + columns / 4 + the 64-byte result record, summed on the "
                             "device (cns_stitch) over the step's jobs, / the forward kernel's launches; like dw it is integer VALU work on LDS-resident "
                             "rows: the HBM fraction is small by construction (SURVEY.md §8d), its row log (16 bytes per d-row) and the traffic "
                             "counters are in profiles/"},
    }
    # HBM bytes of the dominant kernel: PMC counters cannot be read from inside the run; they are quoted from the committed rocprofv3 passes of
    # this command (tools/dev/profile_next_rows.sh -> profiles/<round>_config4_hbm_traffic.json) while profiles/<round>_pmc_source.json names
    # the kernel sources that are compiled now (same rule as bench.py's config-2 line)
    try:
        import re
        import bench as B0
        prof = os.path.join(ROOT, "profiles")
        sfile = sorted(f for f in os.listdir(prof) if re.match(r"r\d+_pmc_source\.json$", f))[-1]
        tag = sfile[: -len("_pmc_source.json")]
        digest = B0.src_digest()
        line["roofline"]["kernel_source_digest"] = digest
        if json.load(open(os.path.join(prof, sfile))).get("kernel_source_digest") == digest:
            t = json.load(open(os.path.join(prof, tag + "_config4_hbm_traffic.json"))).get("dw_extend2<true>" if dname == "cns_forward" else dname)
            if t:
                line["roofline"]["traffic"] = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
                line["roofline"]["traffic_source"] = ("profiles/%s_config4_hbm_traffic.json: per launch of a 23 700-template run of this command (launches there are slices of "
                                                      "the same size class); same kernel sources" % tag)
        else:
            line["roofline"]["traffic_note"] = "committed PMC passes (profiles/%s) describe other kernel sources: not quoted" % sfile
    except Exception as e:      # noqa: BLE001
        line["roofline"]["traffic_note"] = "no committed PMC pass: %r" % (e,)
    if not args.no_cpu:
        try:
            line["cpu_baseline"] = cpu_leg(codes, lens, rec, tb, ids, os.cpu_count() or 1)
            if "value" in line["cpu_baseline"]:
                line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        except Exception as e:      # noqa: BLE001
            line["cpu_baseline"] = {"error": repr(e)[:300]}
    print(json.dumps(W.annotate_cpu_baseline(line)))
