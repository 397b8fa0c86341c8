"""bench.py's multi-volume workloads (`--workload config3 | config5_cell`): the hot path over a grid of volumes, as the mecat2pw driver
walks it (mecat2pw/pw.cpp:65-81, pw_impl.cpp:843-881): for every grid row i one index build of reference volume i, then for every cell
(i, j >= i) the candidates of all reads of query volume j against it and the extension of every candidate — dw for PacBio-style sets,
the X-drop aligner for nanopore-style sets (`-x 1`, pw_impl.cpp:638-644).  Volumes are resident in HBM when the clock starts.

  config3        BASELINE.json configs[2]: 500 000 x 12 kb @ 15 %, 30x of a 200 Mb genome, seed 3 — three volumes, six cells
  config5_cell   one off-diagonal cell of configs[4] (2 M ONT-style reads x 20 kb @ 12 %, seed 5: 19 volumes, 190 cells): volumes 0 and 1
                 exactly as the splitter cuts them, reads of volume 1 against the index of volume 0, -x 1 gates, X-drop extension
  config5        configs[4] whole: 19 volumes (40 Gbase), all 190 cells, -x 1 -j 1 — minutes per step on one GPU (use --steps 1 --warmup 0);
                 with N >= 2 ranks the grid rows are dealt out by cost as the driver's rows mode does (no data moves)

Same contract line as the config-2 run (bench.py), with `roofline.phases` = algorithmic bytes / phase time for index,
seeding and extension, and for the X-drop kernel its DP cells/s and issue rates.  N > 1: every cell sharded over the ranks by the
library's own calls (chunks of 500 query reads, candidate / result all-gathers over RCCL), as the driver's `cells` mode does.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def can_lines(cands, counts, q_lens, q_start_id, ref_lens, ref_start_id):
    """the `.can` lines of a cell's candidate table (candidate_detect, mecat2pw/pw_impl.cpp:767-792) -> list of bytes"""
    n, maxc = cands.shape
    mask = np.arange(maxc)[None, :] < counts[:, None]
    rid = np.broadcast_to(np.arange(n, dtype=np.int64)[:, None], mask.shape)[mask]
    c = cands[mask]
    qext, sext = c["loc2"].astype(np.int64), c["loc1"].astype(np.int64)
    both = (qext != 0) & (sext != 0)
    qext[both] += 6
    sext[both] += 6
    qsize = q_lens[rid].astype(np.int64)
    rev = c["chain"] == 1
    qext[rev] = qsize[rev] - 1 - qext[rev]
    ssize = ref_lens[c["readno"] - ref_start_id].astype(np.int64)
    cols = np.stack([rid + q_start_id, c["readno"].astype(np.int64), c["chain"].astype(np.int64), np.zeros(len(c), np.int64), qext, sext,
                     c["score"].astype(np.int64), qsize, ssize], axis=1)
    return [b"%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n" % tuple(r) for r in cols.tolist()]


def cpu_baseline_grid(workload, nthreads):
    """the unmodified reference (oracle/_ref/mecat2pw) on a bounded sample with the workload's read length, error model and the coverage
    a query read meets: all-vs-all of sn reads on a genome scaled so that a read finds as many true partners as in the measured cells."""
    from mecat_amd import workload as W
    name, nvols, cells, _mcs = W.GRIDS[workload]
    n, L, err, G, seed, ont = W.CONFIGS[name]
    sn = 16000
    if workload == "config5_cell":
        # a read of volume 1 meets the 107 641 reads of volume 0 (1.65x of the genome); in a one-volume all-vs-all a read meets the reads
        # with lower ids, half the set on average: the sample's coverage is twice that of one volume
        per_vol = W.MCS / (L * (1.0 - 0.05 * err) + 1)
        sG = int(G * sn / (2 * per_vol))
        what = "coverage %.2fx = what a read of volume 1 finds in volume 0, twice (a one-volume all-vs-all pairs a read with the lower ids only)" % (sn * L / sG)
    else:
        sG = max(int(G * sn / n), 2 * L)
        what = "same coverage as the whole set (%.0fx)" % (sn * L / sG)
    nthreads = max(1, min(nthreads, (sn + 499) // 500))          # the reference hands out chunks of 500 reads (pw_impl.h:15)
    codes, lens = W.synth_reads(sn, L, err, sG, seed, ont)
    d = tempfile.mkdtemp(prefix="mecat_cpu_")
    try:
        fa = os.path.join(d, "s.fa")
        W.write_fasta(fa, codes, lens)
        ref = os.path.join(ROOT, "oracle", "_ref", "mecat2pw")
        sample = "%d %s-style reads x %d bp @ %.0f%% error, genome %d, seed %d: %s; mecat2pw %s-t %d" % (
            sn, "ONT" if ont else "PacBio", L, err * 100, sG, seed, what, "-x 1 " if ont else "", nthreads)
        if not os.path.exists(ref):
            return {"value": None, "unit": "candidates/s", "cores": nthreads, "kind": "reference", "sample": sample, "error": "oracle/_ref/mecat2pw missing"}
        res = {}
        for task, nm in ((0, "can"), (1, "m4")):
            out = os.path.join(d, "o." + nm)
            t0 = time.time()
            p = subprocess.run([ref, "-j", str(task), "-d", fa, "-o", out, "-w", os.path.join(d, "w" + nm), "-t", str(nthreads), "-g", "1",
                                "-x", str(ont)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            wall = time.time() - t0
            if p.returncode != 0:
                return {"value": None, "unit": "candidates/s", "cores": nthreads, "kind": "reference", "sample": sample, "error": p.stderr[-300:]}
            tm = dict(re.findall(r"\[([a-z_ 0-9]+)\] takes ([0-9.]+) secs", p.stderr))
            hot = float(tm.get("create_ref_index", 0)) + float(tm.get("process volume 0", 0))
            ab = nl = 0
            with open(out) as f:
                for ln in f:
                    nl += 1
                    if task == 1:
                        x = ln.split("\t", 8)
                        ab += int(x[6]) - int(x[5])
            res[nm] = (nl, hot, wall, ab, float(tm.get("create_ref_index", 0)))
        ncan, hot0, wall0, _, ix0 = res["can"]
        nm4, hot1, wall1, ab, _ = res["m4"]
        return {"value": ncan / hot0, "unit": "candidates/s", "cores": nthreads, "kind": "reference", "sample": sample,
                "candidates": ncan, "hot_path_s": hot0, "create_ref_index_s": ix0, "wall_s": wall0,
                "j1_overlaps": nm4, "j1_overlaps_per_s": nm4 / hot1, "j1_aligned_gbase_per_s": ab / 1e9 / hot1, "j1_hot_path_s": hot1,
                "j1_candidates_per_s": ncan / hot1}
    finally:
        subprocess.run(["rm", "-rf", d])


def pmc_summaries(workload, digest):
    """committed rocprofv3 --pmc passes of this command (profiles/rNN_<workload>_*), quoted only while they describe these sources"""
    prof_dir = os.path.join(ROOT, "profiles")
    try:
        sfile = sorted(f for f in os.listdir(prof_dir) if re.match(r"r\d+_%s_pmc_source\.json$" % re.escape(workload), f))[-1]
    except (OSError, IndexError):
        return None, "no committed PMC pass of this workload"
    meta = json.load(open(os.path.join(prof_dir, sfile)))
    if meta.get("kernel_source_digest") != digest:
        return None, "committed PMC passes (profiles/%s) describe other kernel sources (%s): not quoted" % (sfile, meta.get("kernel_source_digest"))
    tag = sfile[: -len("_pmc_source.json")]
    out = {"tag": tag}
    for k in ("hbm_traffic", "instruction_mix"):
        try:
            out[k] = json.load(open(os.path.join(prof_dir, "%s_%s.json" % (tag, k))))
        except (OSError, ValueError):
            out[k] = {}
    return out, None


def run_e2e(args):
    """SURVEY.md §8d's second denominator for the multi-volume workloads (missing 2): wall clock of the drop-in binary on the
    workload's own FASTA — split into volumes, every grid row's index build, every cell, text output, merge — for `-j 0` and `-j 1 -g 1`
    (with `-x 1` for the nanopore-style sets), the stages summed from the driver's own MECAT_TRACE / timer lines.  The FASTA is streamed
    into /dev/shm by the generator binary (40 Gbase of config 5 never sit in this process)."""
    from mecat_amd import workload as W
    name, nvols, cells, mcs = W.GRIDS[args.workload]
    n, L, err, G, seed, ont = W.CONFIGS[name]
    exe = os.path.join(ROOT, "mecat_amd", "bin", "mecat2pw")
    gen = os.path.join(ROOT, "mecat_amd", "bin", "synth_reads")
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="mecat_e2e_", dir=base)
    threads = min(32, os.cpu_count() or 1)
    out = {"workload": "%s: %d reads x %d bp @ %.0f%% error, genome %d, seed %d%s" % (name, n, L, err * 100, G, seed, ", ONT-style, -x 1" if ont else ""),
           "threads": threads, "dir": base or "tmp"}
    try:
        fa = os.path.join(d, "reads.fa")
        t0 = time.time()
        subprocess.run([gen, fa, str(n), str(L), str(err), str(G), str(seed), str(ont)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out["fasta_bytes"] = os.path.getsize(fa)
        out["fasta_generated_s"] = time.time() - t0
        log("[e2e] %s: FASTA of %.1f GB in %.1f s" % (name, out["fasta_bytes"] / 1e9, out["fasta_generated_s"]))
        env = dict(os.environ, MECAT_TRACE="1")
        if mcs != W.MCS:
            env["MECAT_HIP_MCS"] = str(mcs)
        for task, key in ((0, "j0"), (1, "j1")):
            o = os.path.join(d, "out." + key)
            w = os.path.join(d, "w_" + key)
            time.sleep(3.0)
            t0 = time.time()
            p = subprocess.run([exe, "-j", str(task), "-d", fa, "-o", o, "-w", w, "-t", str(threads), "-g", "1", "-x", str(ont)], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, text=True, env=env)
            wall = time.time() - t0
            if p.returncode != 0:
                out[key] = {"error": p.stderr[-400:]}
                continue
            txt = p.stderr + p.stdout
            st = {"split_raw_dataset": 0.0, "create_ref_index": 0.0, "process_volumes": 0.0, "seed": 0.0, "jobs": 0.0, "extend": 0.0, "format": 0.0,
                  "write_and_copies": 0.0, "load_volume": 0.0, "volume_upload": 0.0, "merge_results": 0.0, "ctx_wait": 0.0}
            for nm, v in re.findall(r"\[([a-z_ 0-9]+)\] takes ([0-9.]+) secs", txt):
                if nm == "split_raw_dataset":
                    st["split_raw_dataset"] += float(v)
                elif nm == "create_ref_index":
                    st["create_ref_index"] += float(v)
                elif nm.startswith("process volume"):
                    st["process_volumes"] += float(v)
            for m in re.finditer(r"stages: seed ([0-9.]+) s, jobs ([0-9.]+) s, extend ([0-9.]+) s, format ([0-9.]+) s, write \+ copies ([0-9.]+) s", txt):
                for k2, v in zip(("seed", "jobs", "extend", "format", "write_and_copies"), m.groups()):
                    st[k2] += float(v)
            for nm, k2 in (("load_volume", "load_volume"), ("volume_upload", "volume_upload"), ("merge_results", "merge_results"), ("wait for ctx_create", "ctx_wait")):
                st[k2] += sum(float(v) for v in re.findall(r"\[trace\] %s\s+([0-9.]+) s" % re.escape(nm), txt))
            nl = ab = 0
            with open(o) as f:
                for ln in f:
                    nl += 1
                    if task == 1:
                        x = ln.split("\t", 8)
                        ab += int(x[6]) - int(x[5])
            rec = {"wall_s": wall, "lines": nl, "lines_per_s": nl / wall, "output_bytes": os.path.getsize(o), "stages_s": st,
                   "hot_path_s": st["create_ref_index"] + st["seed"] + st["extend"],
                   "note": "stages_s: the driver's own timers summed over the grid rows; format / write run on a second thread beside the GPU stages (they overlap); "
                           "hot_path_s = index builds + seeding + extension as the driver saw them"}
            rec["wall_over_hot_path"] = wall / rec["hot_path_s"] if rec["hot_path_s"] > 0 else None
            big = max(((k2, v) for k2, v in st.items() if k2 not in ("process_volumes", "seed", "extend", "create_ref_index")), key=lambda kv: kv[1])
            rec["largest_stage_outside_the_hot_path"] = {big[0]: big[1]}
            if task == 1:
                rec["aligned_gbase_per_s"] = ab / 1e9 / wall
            out[key] = rec
            log("[e2e] %s -j %d: %.1f s wall, %d lines; hot path %.1f s; stages %s" % (name, task, wall, nl, rec["hot_path_s"], json.dumps(st)))
            subprocess.run(["rm", "-rf", w, o])
    finally:
        subprocess.run(["rm", "-rf", d])
    print(json.dumps({"metric": "end-to-end wall clock of the drop-in binary", "e2e": out}), flush=True)


def run(args):
    if getattr(args, "e2e", False):
        return run_e2e(args)
    import torch
    import torch.distributed as dist
    import bench as B
    from mecat_amd import hip as M
    from mecat_amd import workload as W

    name, nvols, cells, mcs = W.GRIDS[args.workload]
    n_all, L, err, G, seed, ont = W.CONFIGS[name]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MECAT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but %d visible GPUs" % (world, torch.cuda.device_count()))
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    t0 = time.time()
    # MECAT_BENCH_VOLCACHE=<dir>: keep the generated volumes there between runs (the profiling passes start this command seven times)
    cache = os.environ.get("MECAT_BENCH_VOLCACHE")
    cfile = os.path.join(cache, "%s_%d.npz" % (name, nvols)) if cache else None
    if cfile and os.path.exists(cfile):
        z = np.load(cfile)
        hv = [{"pac": z["pac%d" % k], "offs": z["offs%d" % k], "lens": z["lens%d" % k], "num_bases": int(z["meta"][k, 0]), "start_read_id": int(z["meta"][k, 1])}
              for k in range(len(z["meta"]))]
    else:
        hv = W.synth_volumes(name, nvols, mcs=mcs)
        if cfile and rank == 0:
            os.makedirs(cache, exist_ok=True)
            arrs = {"meta": np.array([[v["num_bases"], v["start_read_id"]] for v in hv], dtype=np.int64)}
            for k, v in enumerate(hv):
                arrs.update({"pac%d" % k: v["pac"], "offs%d" % k: v["offs"], "lens%d" % k: v["lens"]})
            np.savez(cfile + ".tmp.npz", **arrs)
            os.replace(cfile + ".tmp.npz", cfile)
    if rank == 0:
        log("[bench] %s: %d volumes of %s (%s reads, %s bases incl. pads), generated + packed in %.1f s" % (
            args.workload, len(hv), name, "+".join(str(len(v["lens"])) for v in hv), "+".join(str(v["num_bases"]) for v in hv), time.time() - t0))
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = M.Context(local_rank, stream.cuda_stream)
    vols = [M.Volume(ctx, v["pac"], v["offs"], v["num_bases"], v["start_read_id"]) for v in hv]
    for v in hv:
        del v["pac"]
    params = M.default_params(ont)
    maxc = params.maxc
    rows = sorted(set(i for i, _ in cells))
    nmax = max(len(hv[j]["lens"]) for _, j in cells)
    # N > 1: as the driver chooses (host/main.cpp): with at least as many volumes as ranks the grid ROWS are dealt out by cost
    # (mhip_shard_deal_rows: row i = V - i cells; no data moves, no communicator), otherwise every cell is sharded over the ranks
    rows_mode = world > 1 and len(rows) >= world and os.environ.get("MECAT_HIP_SHARD", "") != "cells"      # (config5_cell: one row, always cells)
    owner = None
    if rows_mode:
        owner, heaviest = M.deal_rows(len(hv), np.array(rows, dtype=np.int32), world)
        rows = [i for i in rows if owner[i] == rank]
        cells = [(i, j) for (i, j) in cells if owner[i] == rank]

    comm = None
    CH = M.SHARD_CHUNK
    if world > 1 and not rows_mode:
        if backend == "nccl":
            box = [M.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = M.Comm(ctx, world, rank, unique_id=box[0])
        else:
            box = [tempfile.mkdtemp(prefix="mecat_bench_comm_") if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = M.Comm(ctx, world, rank, hostfile_dir=box[0], run_id="bench")
        comm.barrier()
    if comm is None:
        d_cands = torch.zeros((nmax, maxc, 12), dtype=torch.int32, device=dev)
        d_counts = torch.zeros((nmax,), dtype=torch.int32, device=dev)
        d_jobs = torch.empty((nmax * maxc + maxc, 5), dtype=torch.int32, device=dev)
        d_res = torch.empty((d_jobs.shape[0], 8), dtype=torch.int32, device=dev)
    # (the library decides by measurement: mhip_index_build_auto builds the first table both ways, timed, and keeps the faster way — done
    # here, before the warm-up, so that no timed step carries the measurement)
    index_how = {"chosen": "replicated", "measured": False}
    if comm is not None:
        i0, index_how = comm.index_build_auto(vols[rows[0]])
        i0.free()
        index_how["bytes_received_measuring"] = comm.bytes_received()
        tr0, nr0 = comm.info()
        if tr0 == 0 and nr0 != world:
            raise SystemExit("[bench] rank %d: the RCCL communicator has %d ranks, --gpus asked for %d" % (rank, nr0, world))
    keep = {"num_kmers": {}, "cell": {}}
    L_ = M.lib()
    # cells whose sorted `.can` lines the golden file pins; with many cells (config 5 whole) only those are formatted and hashed
    golden_cells = None
    if len(cells) > 8:
        try:
            big = json.load(open(os.path.join(ROOT, "tests", "golden", "big.json")))[name]
            golden_cells = set(k for r in big.get("rows", {}).values() for k in r.get("cells", {}))
        except Exception:  # noqa: BLE001
            golden_cells = set()

    def one_step(collect=False):
        ms = {"index": 0.0, "seed": 0.0, "align": 0.0}
        evs = []
        for i in rows:
            e0 = torch.cuda.Event(enable_timing=True); e0.record(stream)
            idx = comm.index_build_auto(vols[i])[0] if comm is not None else M.Index(ctx, vols[i])
            e1 = torch.cuda.Event(enable_timing=True); e1.record(stream)
            evs.append(("index", e0, e1, i, i))
            keep["num_kmers"][i] = idx.num_kmers
            for (ci, j) in cells:
                if ci != i:
                    continue
                nj_reads = len(hv[j]["lens"])
                a = torch.cuda.Event(enable_timing=True); a.record(stream)
                if comm is None:
                    M.seed_reads_dev(ctx, idx, vols[i], vols[j], 0, nj_reads, params, d_cands.data_ptr(), d_counts.data_ptr())
                else:
                    comm.seed_reads_sharded(idx, vols[i], vols[j], 0, nj_reads, params, chunk=CH, cell_shift=j, host=False)
                b = torch.cuda.Event(enable_timing=True); b.record(stream)
                njobs = 0
                if not args.no_align:
                    if comm is None:
                        njobs = M.jobs_from_candidates_dev(ctx, d_cands.data_ptr(), d_counts.data_ptr(), nj_reads, maxc, 0, 1, hv[i]["start_read_id"], 0, 1,
                                                           d_jobs.data_ptr())
                        if ont:
                            M._chk(L_.mhip_xalign_candidates_dev(ctx.h, vols[i].h, vols[j].h, d_jobs.data_ptr(), njobs, params.min_align_size, d_res.data_ptr()))
                        else:
                            M.align_candidates_dev(ctx, vols[i], vols[j], d_jobs.data_ptr(), njobs, params.min_align_size, d_res.data_ptr())
                    else:
                        _, njobs = comm.align_sharded(vols[i], vols[j], params.min_align_size, tech=ont, host=False)
                c = torch.cuda.Event(enable_timing=True); c.record(stream)
                evs.append(("seed", a, b, i, j))
                evs.append(("align", b, c, i, j))
                if collect and comm is None:
                    stream.synchronize()
                    h_cnt = d_counts[:nj_reads].cpu().numpy()
                    cell = {"candidates": int(h_cnt.sum()), "jobs": int(njobs)}
                    if collect == "lines" and (golden_cells is None or "%d,%d" % (i, j) in golden_cells):
                        h_c = d_cands[:nj_reads].cpu().numpy().view(M.CAND_DTYPE).reshape(nj_reads, maxc)
                        lines = can_lines(h_c, h_cnt, hv[j]["lens"], hv[j]["start_read_id"], hv[i]["lens"], hv[i]["start_read_id"])
                        lines.sort()
                        h = hashlib.sha256()
                        for ln in lines:
                            h.update(ln)
                        cell["can_lines"] = len(lines)
                        cell["can_sorted_sha256"] = h.hexdigest()
                    if not args.no_align and njobs:
                        r = d_res[:njobs]
                        ok = r[:, 0] != 0
                        cell["overlaps_ok"] = int(ok.sum().item())
                        cell["aligned_bases"] = int(((r[:, 2] - r[:, 1]).to(torch.int64) * ok).sum().item())
                    keep["cell"]["%d,%d" % (i, j)] = cell
            stream.synchronize()
            idx.free()
        per_row = {}
        for k, a, b, ri, _cj in evs:
            t = a.elapsed_time(b)
            ms[k] += t
            per_row[ri] = per_row.get(ri, 0.0) + t
        if not collect:
            keep["row_ms"] = per_row      # the last timed step's time per grid row (index build + every cell of the row)
        return ms

    if getattr(args, "simulate_ranks", ""):
        # cells mode as a measurement (bench.py's simulate_ranks for one cell, here over the grid): every rank's share
        # of every cell — its key range of the row's index, its chunks of the query volume — alone on this GPU, no transport
        if world != 1:
            raise SystemExit("--simulate-ranks runs on one GPU")

        def wall(f):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3, r

        one_step()
        base = one_step()
        out = {"what": "every rank's share of the sharded calls of every cell, alone on one MI355X, no transport (mhip_comm_init_solo)", "workload": args.workload,
               "one_gpu_phase_ms": base, "one_gpu_step_ms": float(sum(base.values())), "P": {}}
        for P in [int(x) for x in args.simulate_ranks.split(",") if x]:
            ranks = []
            for r in range(P):
                cm = M.Comm(ctx, P, r, solo=True)
                best = None
                for rep in range(2):
                    t = {"index_slice_ms": 0.0, "index_rebuilt_ms": 0.0, "seed_ms": 0.0, "align_ms": 0.0}
                    cand = 0
                    s0 = cm.bytes_sent()
                    for i in rows:
                        ms, ix = wall(lambda: cm.index_build_sharded(vols[i]))
                        t["index_slice_ms"] += ms
                        ix.free()
                        ms, idx = wall(lambda: M.Index(ctx, vols[i]))
                        t["index_rebuilt_ms"] += ms
                        for (ci, j) in cells:
                            if ci != i:
                                continue
                            nq = len(hv[j]["lens"])
                            ms, _ = wall(lambda: cm.seed_reads_sharded(idx, vols[i], vols[j], 0, nq, params, chunk=CH, cell_shift=j, host=False))
                            t["seed_ms"] += ms
                            cand += cm.local_jobs()
                            if not args.no_align:
                                ms, _ = wall(lambda: cm.align_sharded(vols[i], vols[j], params.min_align_size, tech=ont, host=False))
                                t["align_ms"] += ms
                        idx.free()
                    t["bytes_to_each_peer"] = (cm.bytes_sent() - s0) // max(1, P - 1)
                    t["candidates"] = cand
                    if best is None or t["seed_ms"] + t["align_ms"] < best["seed_ms"] + best["align_ms"]:
                        best = t
                best["rank"] = r
                ranks.append(best)
                cm.close()
            rep_t = np.array([x["index_rebuilt_ms"] + x["seed_ms"] + x["align_ms"] for x in ranks])
            shd_t = np.array([x["index_slice_ms"] + x["seed_ms"] + x["align_ms"] for x in ranks])
            one = float(sum(base.values()))
            out["P"][str(P)] = {"ranks": ranks, "sum_of_rank_candidates": int(sum(x["candidates"] for x in ranks)),
                                "imbalance_max_over_mean": {"seed": float(max(x["seed_ms"] for x in ranks) / np.mean([x["seed_ms"] for x in ranks])),
                                                            "align": float(max(x["align_ms"] for x in ranks) / max(1e-9, np.mean([x["align_ms"] for x in ranks])))},
                                "slowest_rank_ms": {"index_rebuilt_on_every_rank": float(rep_t.max()), "index_in_key_range_shards": float(shd_t.max())},
                                "compute_only_speedup_bound": {"index_rebuilt_on_every_rank": one / float(rep_t.max()), "index_in_key_range_shards": one / float(shd_t.max())}}
            log("[simulate] %s cells mode P=%d: slowest rank %.1f / %.1f ms (index rebuilt / sharded), 1 GPU %.1f ms -> bound %.2fx / %.2fx"
                % (args.workload, P, rep_t.max(), shd_t.max(), one, one / rep_t.max(), one / shd_t.max()))
        print(json.dumps(out), flush=True)
        for v in vols:
            v.free()
        ctx.close()
        return

    ctx.set_profiling(True)
    for _ in range(args.warmup):
        one_step()
    ctx.reset_stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [one_step() for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    kstats = ctx.kernel_stats()
    counters = ctx.counters()
    dbg = {s: ctx.debug_counter(s) for s in (9, 12, 15, 32)}
    ctx.set_profiling(False)

    # result statistics and the reference pins, after the timed region: one more (untimed) pass that brings every cell's table to the host
    exch = None
    if world == 1:
        one_step(collect="lines" if not args.no_cpu or os.environ.get("MECAT_BENCH_PARITY") else True)
        ncand = sum(c["candidates"] for c in keep["cell"].values())
        aln_ok = sum(c.get("overlaps_ok", 0) for c in keep["cell"].values())
        aligned_bases = sum(c.get("aligned_bases", 0) for c in keep["cell"].values())
    elif rows_mode:
        exch = {"transport": "none (rows mode: every grid row is computed by one rank, nothing is exchanged)", "rccl_ranks": 0,
                "rows_of_rank": [int(np.sum(owner == r)) for r in range(world)],
                "cells_of_rank": [int(sum(len(hv) - i for i in range(len(hv)) if owner[i] == r)) for r in range(world)], "heaviest_rank_cells": int(heaviest)}
        t = torch.tensor([counters["candidates"] // args.steps, counters["aln_ok"] // args.steps, counters["aligned_bases"] // args.steps],
                         dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        ncand, aln_ok, aligned_bases = (int(x) for x in t.tolist())
    else:
        tr, nr = comm.info()
        exch = {"bytes_received_per_step": (comm.bytes_received() - index_how.get("bytes_received_measuring", 0)) / max(1, args.steps + args.warmup),
                "transport": "rccl" if tr == 0 else "host files (test hook)", "rccl_ranks": nr, "index_build": index_how,
                "ms": kstats.get("xg_exchange", (0, 0.0))[1] / args.steps, "calls_per_step": kstats.get("xg_exchange", (0, 0.0))[0] / args.steps,
                "index_exchange_ms": kstats.get("xg_exchange_index", (0, 0.0))[1] / args.steps,
                "index_build_kernels_ms": sum(v[1] for k, v in kstats.items() if k.startswith(("ix_", "idx"))) / args.steps,
                "index_rebase_slots_ms": sum(v[1] for k, v in kstats.items() if k.startswith("xg_index")) / args.steps}
        ncand = int(counters["candidates"] // args.steps)
        aln_ok = int(counters["aln_ok"] // args.steps)
        aligned_bases = int(counters["aligned_bases"] // args.steps)
        t = torch.tensor([ncand, aln_ok, aligned_bases], dtype=torch.int64, device=dev)
        dist.all_reduce(t)          # every rank counted the candidates of its own reads
        ncand, aln_ok, aligned_bases = (int(x) for x in t.tolist())
    if world > 1 and exch is not None:
        # every rank's own phase times (HIP events on its stream, mean over the timed steps): where a short curve fell short
        mine = {"rank": rank, "phase_ms": {k: float(np.mean([o[k] for o in outs])) for k in ("index", "seed", "align")}}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        exch["per_rank"] = allr

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        phase = {k: float(np.mean([o[k] for o in outs])) for k in ("index", "seed", "align")}
        per_step = lambda k: counters[k] / args.steps          # noqa: E731
        lookups, hits, cands_c = per_step("lookups"), per_step("hits"), per_step("candidates")
        # algorithmic bytes per step: index = two passes over the 2-bit volume + the 4^13 table three times + the kept
        # positions; seeding = the query reads' 2-bit bases (both strands) + 8 B per lookup + 4 B per bucket hit + 48 B per candidate;
        # extension = the aligned spans of both reads at 2 bits per base + a 32-byte result
        b_idx = sum(2 * (hv[i]["num_bases"] / 4) + 3 * 4 * (1 << 26) + 4 * keep["num_kmers"][i] for i in rows)
        b_seed = sum(2 * int(hv[j]["lens"].astype(np.int64).sum()) / 4 for _, j in cells) / (1 if comm is None else world) + 8 * lookups + 4 * hits + 48 * cands_c
        b_aln = per_step("aligned_bases") * 2 / 4 + 32 * cands_c
        pbytes = {"index": b_idx, "seed": b_seed, "align": b_aln}
        phases = {k: {"algorithmic_bytes": float(pbytes[k]), "ms": phase[k],
                      "achieved_GBs": pbytes[k] / 1e9 / (phase[k] / 1e3) if phase[k] > 0 else None,
                      "frac": pbytes[k] / 1e9 / (phase[k] / 1e3) / HBM_PEAK_GBS if phase[k] > 0 else None} for k in pbytes}
        dom = max(kstats.items(), key=lambda kv: kv[1][1]) if kstats else ("none", (1, 0.0))
        dname, (dl, dms) = dom
        pk = "index" if dname.startswith(("ix_", "idx")) else ("align" if dname.startswith(("dw_", "xd_")) else "seed")
        # the kernels of the dominant kernel's phase share the phase's algorithmic bytes; the dominant kernel is priced with all of them
        avg_ms = dms / max(1, dl)
        launches_per_step = max(1, dl // args.steps)
        alg_per_launch = pbytes[pk] / launches_per_step
        achieved = alg_per_launch / 1e9 / (avg_ms / 1e3) if avg_ms > 0 else 0.0
        roof = {"bound": "hbm", "kernel": dname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "algorithmic_bytes_per_launch": alg_per_launch, "avg_launch_ms": avg_ms, "launches": dl, "phases": phases,
                "note": "algorithmic bytes of the kernel's phase (SURVEY.md §8d) / launches per step; scratch / sort traffic not counted"}
        digest = B.src_digest()
        roof["kernel_source_digest"] = digest
        pmc, why = pmc_summaries(args.workload, digest) if world == 1 else (None, "PMC passes are single-GPU")
        if pmc is None:
            roof["traffic_note"] = why
        else:
            t = pmc["hbm_traffic"].get(dname)
            if t:
                roof["traffic"] = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
                roof["traffic_source"] = "profiles/%s_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command, same sources)" % pmc["tag"]
            # seeding phase: counted traffic of all its kernels against its algorithmic bytes
            seedk = [k for k in pmc["hbm_traffic"] if k.startswith("seed_")]
            if seedk:
                tr = sum((pmc["hbm_traffic"][k]["fetch_bytes_per_launch"] + pmc["hbm_traffic"][k]["write_bytes_per_launch"]) * pmc["hbm_traffic"][k]["launches_per_pass"]
                         for k in seedk)
                phases["seed"]["traffic_bytes"] = tr
                phases["seed"]["traffic_over_algorithmic"] = tr / pbytes["seed"] if pbytes["seed"] else None
        if dname.startswith("xd_"):
            # the X-drop kernel is integer work on LDS-resident rows (HBM fraction tiny by construction, like dw): DP cells per second
            # and, from the committed SQ passes, its instruction issue rates against the nominal per-SIMD rates (one wave64 VALU
            # instruction per 4 cycles for everything but the plain 32-bit class, one SALU instruction per 4 cycles)
            xs = {"dp_cells_per_step": per_step("dw_cells"), "dp_cells_per_s": per_step("dw_cells") / (phase["align"] / 1e3) if phase["align"] > 0 else None,
                  "rows_per_step": dbg[32] / args.steps, "blocks_per_step": per_step("dw_blocks"),
                  "cells_per_row": per_step("dw_cells") / max(1.0, dbg[32] / args.steps)}
            if pmc is not None and dname in pmc["instruction_mix"] and avg_ms > 0:
                im = pmc["instruction_mix"][dname]
                nominal = 1024 * 2.4 / 4
                xs.update({"valu_wave_insts_per_s_G": im["valu_insts_per_launch"] / (avg_ms / 1e3) / 1e9,
                           "salu_wave_insts_per_s_G": im["salu_insts_per_launch"] / (avg_ms / 1e3) / 1e9,
                           "issue_ceiling_4cycle_G": nominal,
                           "valu_frac_of_4cycle_issue": im["valu_insts_per_launch"] / (avg_ms / 1e3) / 1e9 / nominal,
                           "salu_frac_of_issue": im["salu_insts_per_launch"] / (avg_ms / 1e3) / 1e9 / nominal,
                           "valu_insts_per_dp_cell": im["valu_insts_per_launch"] * launches_per_step / max(1.0, per_step("dw_cells")),
                           "source": "profiles/%s_instruction_mix.json (SQ_INSTS_VALU / SQ_INSTS_SALU per launch, same sources)" % pmc["tag"]})
            roof["xdrop"] = xs
        wl = "%s: " % args.workload
        if args.workload == "config5_cell":
            wl += ("grid cell (0, 1) of config 5 (%d ONT-style reads x %d bp @ %.0f%% error, genome %d, seed %d: 19 volumes): reads of volume 1 (%d) against "
                   "volume 0 (%d reads, %d bases), k=13, -x 1 -j 1 (index + seed + X-drop)" % (n_all, L, err * 100, G, seed, len(hv[1]["lens"]), len(hv[0]["lens"]),
                                                                                             hv[0]["num_bases"]))
        else:
            wl += ("%d reads x %d bp @ %.0f%% error, genome %d, seed %d, k=13: %d volumes, all %d grid cells, -j 1 (index per row + seed + %s per cell)"
                   % (n_all, L, err * 100, G, seed, len(hv), len(W.GRIDS[args.workload][2]), "X-drop" if ont else "dw"))
        line = {
            "metric": "candidate overlaps/sec", "value": ncand / (ms_step / 1e3), "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": wl, "volumes": [{"reads": len(v["lens"]), "bases": int(v["num_bases"]), "start_read_id": int(v["start_read_id"])} for v in hv],
                       "cells": ["%d,%d" % c for c in cells] if len(cells) <= 32 else "%d cells" % len(cells),
                       "parallelism": "1 GPU" if world == 1 else ("rows mode: the %d grid rows dealt out by cost (row i = V - i cells), heaviest rank %d cells; no data moves; "
                                                                   "roofline / phase figures are rank 0's own rows" % (len(hv), int(heaviest))) if rows_mode else "every cell sharded: chunks of %d reads, chunk c of volume j -> rank (c + j) mod %d; RCCL count-then-payload "
                                      "all-gather; index %s" % (CH, world, "built in k-mer key-range shards + all-gather" if index_how["chosen"] == "sharded" else "rebuilt on every rank")},
            "candidates": ncand, "overlaps_ok": aln_ok, "aligned_gbase_per_s": aligned_bases / 1e9 / (ms_step / 1e3),
            "overlaps_per_s": aln_ok / (ms_step / 1e3), "phase_ms": phase,
            "candidates_per_s_index_seed_phases": ncand / ((phase["index"] + phase["seed"]) / 1e3),
            "aligned_gbase_per_s_align_phase": (aligned_bases / 1e9 / (phase["align"] / 1e3)) if phase["align"] > 0 else None,
            "counters_per_step": {k: v / args.steps for k, v in counters.items()},
            "kernel_ms_per_step": {k: v[1] / args.steps for k, v in sorted(kstats.items(), key=lambda kv: -kv[1][1])},
            "roofline": roof,
        }
        if exch:
            line["exchange"] = exch
        # rows mode as a measurement: with at least as many grid rows as ranks the driver deals whole rows out
        # (mhip_shard_deal_rows; nothing is exchanged), so P ranks take max-over-ranks of the summed row times — computed here from the times
        # this one GPU just measured for every row (index build + its cells; the untimed collecting pass, which also copies the tables out)
        if world == 1 and keep.get("row_ms") and len(keep["row_ms"]) >= 2:
            rm = keep["row_ms"]
            allrows = sorted(rm)
            tot = float(sum(rm.values()))
            sim = {"row_ms": {str(i): rm[i] for i in allrows}, "P": {}}
            for P in (2, 4, 8):
                if P > len(allrows):
                    continue
                own, _h = M.deal_rows(len(hv), np.array(allrows, dtype=np.int32), P)
                per = [float(sum(rm[i] for i in allrows if own[i] == r)) for r in range(P)]
                sim["P"][str(P)] = {"rank_ms": per, "rows_of_rank": [[int(i) for i in allrows if own[i] == r] for r in range(P)],
                                    "imbalance_max_over_mean": max(per) / (sum(per) / P), "compute_only_speedup_bound": tot / max(per)}
            line["simulated_rows_mode"] = sim
        # reference pins of the cells (tests/golden/big.json: line counts and sorted-output hashes of the unmodified mecat2pw -j 0)
        if keep["cell"]:
            line["cells"] = keep["cell"]
            try:
                big = json.load(open(os.path.join(ROOT, "tests", "golden", "big.json")))[name]
                checked, same = 0, True
                for key, c in keep["cell"].items():
                    i = key.split(",")[0]
                    g = big.get("rows", {}).get(i, {}).get("cells", {}).get(key)
                    if g and "can_sorted_sha256" in c:
                        checked += 1
                        same = same and g["lines"] == c["can_lines"] and g["sorted_sha256"] == c["can_sorted_sha256"]
                if checked:
                    line["parity_vs_reference"] = {"cells_checked": checked, "identical": bool(same),
                                                   "source": "tests/golden/big.json: sorted `.can` lines of the unmodified mecat2pw -j 0 per grid cell"}
            except Exception as e:  # noqa: BLE001
                line["parity_vs_reference"] = {"error": repr(e)[:200]}
        log("[bench] kernel ms/step: " + ", ".join("%s=%.2f" % (k, v) for k, v in line["kernel_ms_per_step"].items()))
        if args.stats:
            json.dump({"kernels": {k: {"launches": v[0], "total_ms": v[1]} for k, v in kstats.items()}, "line": line}, open(args.stats, "w"), indent=1)
        if world == 1:              # the CPU leg below needs no device: give the memory back first
            for v in vols:
                v.free()
            ctx.close()
            vols = []
        if world == 1 and not args.no_cpu:
            try:
                line["cpu_baseline"] = cpu_baseline_grid("config5_cell" if args.workload == "config5" else args.workload, os.cpu_count() or 1)
                gb = json.load(open(os.path.join(ROOT, "tests", "golden", "big.json"))).get(name, {})
                if args.workload in ("config5_cell", "config5") and "rows" in gb and "0" in gb["rows"] and "seconds" in gb["rows"]["0"]:
                    r0 = gb["rows"]["0"]
                    line["cpu_baseline"]["full_size_reference"] = {
                        "what": "grid row 0 of config 5 (19 of the 190 cells), unmodified mecat2pw -j 0 -x 1", "threads": r0.get("threads"),
                        "host": "build container, not this host", "seconds": r0["seconds"], "candidates": r0["lines"],
                        "candidates_per_s": r0["lines"] / r0["seconds"], "source": "tests/golden/big.json"}
                elif "j0_seconds" in gb and "can_lines" in gb:
                    line["cpu_baseline"]["full_size_reference"] = {
                        "what": "the whole workload, unmodified mecat2pw -j 0", "threads": gb.get("reference_threads"), "host": "build container, not this host",
                        "seconds": gb["j0_seconds"], "candidates": gb["can_lines"], "candidates_per_s": gb["can_lines"] / gb["j0_seconds"],
                        "source": "tests/golden/big.json"}
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": os.cpu_count(), "kind": "reference", "sample": "failed: %r" % (e,)}
        print(json.dumps(W.annotate_cpu_baseline(line)), flush=True)
    released = rank == 0 and world == 1
    if not released:
        for v in vols:
            v.free()
    if comm is not None:
        comm.barrier()
        comm.close()
    if not released:
        ctx.close()
    if world > 1:
        dist.destroy_process_group()
