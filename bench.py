#!/usr/bin/env python3
"""bench.py — the mecat2pw hot path on MI355X: index build -> seed/DDF filter -> dw extension.

One "step" = one full pass of the hot path over the workload, inputs (the 2-bit volume) already resident in HBM:
    index build of the volume  ->  candidates of every read (both strands)  ->  dw extension of every candidate
Workload (BASELINE.json configs[1]): 100 000 synthetic PacBio-style reads x 15 kb @ 15 % error, 30x of a
50 Mb random genome, seed 2, k = 13, all-vs-all (one volume, one grid cell).  `--workload config1` = 1 000 x 10 kb.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (one rank per GPU, RCCL)
    python bench.py --gpus N ...            (no launcher: the script re-executes itself under torch.distributed.run with N ranks)

N > 1 (strong scaling, total work fixed): the library's own multi-GPU calls, as in the mecat2pw driver — from four ranks on the
index is built in k-mer key-range shards and all-gathered (mhip_index_build_sharded; with two or three ranks every rank rebuilds
it: 4.7 GB of positions over one or two xGMI links cost more than the 23 ms rebuild), the reads are dealt out in chunks of 500
(chunk c -> rank c mod N), each rank seeds and extends its own, and the candidate lists and extension results are all-gathered
count-then-payload over RCCL (mhip_seed_reads_sharded / mhip_align_sharded), so every rank holds the complete candidate table
and overlap set.

Prints ONE JSON line on rank 0 (see the task contract): value = candidate overlaps/sec of the whole job, plus
aligned Gbase/sec, a `roofline` block for the dominant kernel and a `cpu_baseline` block (the unmodified reference
mecat2pw from oracle/_ref, or the oracle port, timed on this host on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(workload, nthreads):
    """reference mecat2pw (kind "reference") on a bounded sample with the same read length / error / coverage"""
    from mecat_amd import workload as W
    n, L, err, G, seed, ont = W.CONFIGS[workload]
    # the reference hands out reads in chunks of 500 (CHUNK_SIZE, mecat2pw/pw_impl.h:15): at most sn/500 threads are busy
    sn = min(n, 16000)
    nthreads = max(1, min(nthreads, (sn + 499) // 500))
    sG = max(int(G * sn / n), 2 * L)
    codes, lens = W.synth_reads(sn, L, err, sG, seed, ont)
    d = tempfile.mkdtemp(prefix="mecat_cpu_")
    fa = os.path.join(d, "s.fa")
    W.write_fasta(fa, codes, lens)
    ref = os.path.join(ROOT, "oracle", "_ref", "mecat2pw")
    sample = ("%d reads x %d bp @ %.0f%% error, genome %d (same coverage), seed %d; the sampled volume (%.2f Gbase) is %.1fx smaller than "
              "%s's, so random 13-mer hits per lookup are that much fewer: the CPU rate is optimistic for the CPU"
              % (sn, L, err * 100, sG, seed, sn * L * 1.05 / 1e9, n / sn, workload))
    if os.path.exists(ref):
        res = {}
        for task, name in ((0, "can"), (1, "m4")):
            out = os.path.join(d, "o." + name)
            t0 = time.time()
            p = subprocess.run([ref, "-j", str(task), "-d", fa, "-o", out, "-w", os.path.join(d, "w" + name), "-t", str(nthreads),
                                "-g", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            wall = time.time() - t0
            if p.returncode != 0:
                return {"value": None, "unit": "candidates/s", "cores": nthreads, "kind": "reference", "sample": sample,
                        "error": p.stderr[-300:]}
            tm = dict(re.findall(r"\[([a-z_ 0-9]+)\] takes ([0-9.]+) secs", p.stderr))
            hot = float(tm.get("create_ref_index", 0)) + float(tm.get("process volume 0", 0))
            lines = open(out).read().splitlines()
            res[name] = (len(lines), hot, wall, lines)
        ncan, hot0, wall0, _ = res["can"]
        nm4, hot1, wall1, m4 = res["m4"]
        abases = sum(int(x.split("\t")[6]) - int(x.split("\t")[5]) for x in m4)
        return {"value": ncan / hot0, "unit": "candidates/s", "cores": nthreads, "kind": "reference", "sample": sample,
                "candidates": ncan, "hot_path_s": hot0, "wall_s": wall0,
                "j1_overlaps_per_s": nm4 / hot1, "j1_aligned_gbase_per_s": abases / 1e9 / hot1, "j1_hot_path_s": hot1}
    # no reference build on this host: the oracle port (oracle/liboracle.so, single thread) on a smaller sample
    sn2 = min(sn, 2000)
    codes, lens = W.synth_reads(sn2, L, err, max(int(G * sn2 / n), 2 * L), seed, ont)
    t0 = time.time()
    ncan = oracle_port_candidates(codes, lens, ont)
    dt = time.time() - t0
    return {"value": ncan / dt, "unit": "candidates/s", "cores": 1, "kind": "port", "sample": "%d reads x %d bp (oracle port)" % (sn2, L),
            "candidates": ncan, "hot_path_s": dt}


def cpu_full_size(workload, codes, lens, nthreads, gpu_candidates):
    """north_star: "next to reference mecat2pw timed on the node's own host cores (core count stated) in the same run" — the unmodified
    reference (oracle/_ref/mecat2pw -j 0) on the FULL workload's FASTA, once, on this host (-t = min(cores, 64): its index build has
    every thread scan the whole volume for its own key range, lookup_table.cpp:36-58, so threads beyond the memory system add nothing)"""
    from mecat_amd import workload as W
    ref = os.path.join(ROOT, "oracle", "_ref", "mecat2pw")
    if not os.path.exists(ref):
        return None
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="mecat_cpufull_", dir=base)
    try:
        fa = os.path.join(d, "reads.fa")
        W.write_fasta(fa, codes, lens)
        out = os.path.join(d, "o.can")
        t0 = time.time()
        p = subprocess.run([ref, "-j", "0", "-d", fa, "-o", out, "-w", os.path.join(d, "w"), "-t", str(nthreads)], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=1500)
        wall = time.time() - t0
        if p.returncode != 0:
            return {"error": p.stderr[-300:]}
        tm = dict(re.findall(r"\[([a-z_ 0-9]+)\] takes ([0-9.]+) secs", p.stderr))
        hot = float(tm.get("create_ref_index", 0)) + float(tm.get("process volume 0", 0))
        nl = sum(1 for _ in open(out))
        return {"kind": "reference", "what": "unmodified mecat2pw -j 0 on the full %s FASTA (%d bytes), this host, same run" % (workload, os.path.getsize(fa)),
                "cores": nthreads, "host_cpus": os.cpu_count(), "candidates": nl, "same_count_as_gpu": nl == gpu_candidates,
                "hot_path_s": hot, "create_ref_index_s": float(tm.get("create_ref_index", 0)), "wall_s": wall,
                "candidates_per_s": nl / hot if hot > 0 else None, "candidates_per_s_wall": nl / wall}
    finally:
        subprocess.run(["rm", "-rf", d])


def oracle_port_candidates(codes, lens, ont):
    """index + candidates of every read with oracle/liboracle.so (the checker; cpu_baseline leg only)"""
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.orc_bench_candidates.restype = C.c_int64
    lib.orc_bench_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    return int(lib.orc_bench_candidates(codes.ctypes.data, lens.ctypes.data, len(lens), ont))


def cpu_baseline_config1():
    """BASELINE.json configs[0]: the reference on 1 000 x 10 kb @ 15 %, -t 4 (and -t 2: only two 500-read chunks exist)"""
    from mecat_amd import workload as W
    ref = os.path.join(ROOT, "oracle", "_ref", "mecat2pw")
    if not os.path.exists(ref):
        return None
    n, L, err, G, seed, ont = W.CONFIGS["config1"]
    codes, lens = W.synth_reads(n, L, err, G, seed, ont)
    d = tempfile.mkdtemp(prefix="mecat_c1_")
    fa = os.path.join(d, "c1.fa")
    W.write_fasta(fa, codes, lens)
    out = {"workload": "config1: %d reads x %d bp @ %.0f%% error, genome %d, seed %d" % (n, L, err * 100, G, seed), "kind": "reference"}
    for t in (4, 2):
        for task, name in ((0, "j0"), (1, "j1")):
            o = os.path.join(d, "o_%s_%d" % (name, t))
            t0 = time.time()
            p = subprocess.run([ref, "-j", str(task), "-d", fa, "-o", o, "-w", os.path.join(d, "w_%s_%d" % (name, t)), "-t", str(t), "-g", "1"],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            wall = time.time() - t0
            if p.returncode != 0:
                out["error"] = p.stderr[-200:]
                return out
            tm = dict(re.findall(r"\[([a-z_ 0-9]+)\] takes ([0-9.]+) secs", p.stderr))
            hot = float(tm.get("create_ref_index", 0)) + float(tm.get("process volume 0", 0))
            nl = sum(1 for _ in open(o))
            out["%s_t%d" % (name, t)] = {"lines": nl, "hot_path_s": hot, "wall_s": wall, "lines_per_s": nl / hot if hot > 0 else None}
    return out


def cns_cpu_baseline(codes, lens, rec, tb, ids, templates=100):
    """BASELINE config 4's CPU side, same run: the UNMODIFIED consensus_one_read_can_pacbio (oracle/_ref/libref_cns_accept.so = mecat2cns
    compiled from the reference's sources) on the first `templates` templates of the very records the GPU leg was given — one thread, as a
    mecat2cns worker runs it (the reference parallelises over templates: multiply by its thread count).  Runs in a child process with
    OMP_NUM_THREADS=1: the reference is built with -D_GLIBCXX_PARALLEL as its own Makefile does, and on a many-core host every std::sort
    inside a worker would otherwise start a thread team."""
    from mecat_amd import workload as W
    so = os.path.join(ROOT, "oracle", "_ref", "libref_cns_accept.so")
    if not os.path.exists(so):
        return None
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="mecat_cns_", dir=base)
    try:
        fa = os.path.join(d, "reads.fa")
        W.write_fasta(fa, codes, lens)
        K = min(templates, len(ids))
        np.savez(os.path.join(d, "in.npz"), rec=np.ascontiguousarray(rec[: tb[K]]), tb=tb[: K + 1], ids=ids[:K], n=len(lens))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cns-cpu-leg", d], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS="1"), timeout=600)
        if p.returncode != 0:
            return {"error": p.stderr[-300:]}
        out = json.loads(p.stdout.strip().splitlines()[-1])
        out["sample"] = "first %d templates of the GPU leg's records, consensus_one_read_can_pacbio up to the consensus table, one thread" % K
        return out
    finally:
        subprocess.run(["rm", "-rf", d])


def cns_cpu_leg(d, part=0, parts=1):
    """child process of cns_cpu_baseline (and of bench_config4.cpu_leg: templates part, part + parts, ... of the sample)"""
    A = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_cns_accept.so"))
    A.refa_load_reads.argtypes = [C.c_char_p]
    A.refa_consensus_can.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
    g = np.load(os.path.join(d, "in.npz"))
    rec, tb, ids = g["rec"], g["tb"], g["ids"]
    t0 = time.time()
    if A.refa_load_reads(os.path.join(d, "reads.fa").encode()) != int(g["n"]):
        raise SystemExit("refa_load_reads")
    t_load = time.time() - t0
    sbuf = np.zeros(64_000_000, dtype=np.int8)
    meta = np.zeros((128, 4), dtype=np.int32)
    naln = nacc = 0
    t0 = time.time()
    for t in range(part, len(ids), parts):
        b, e = int(tb[t]), int(tb[t + 1])
        cand = np.ascontiguousarray(rec[b:e]).copy()
        used = C.c_long()
        k = A.refa_consensus_can(0, cand.ctypes.data, e - b, int(ids[t]), 2000, 0.9, meta.ctypes.data, sbuf.ctypes.data, len(sbuf), C.byref(used))
        if k < 0:
            raise SystemExit("refa_consensus_can")
        naln += min(e - b, 200)
        nacc += k
    dt = time.time() - t0
    print(json.dumps({"kind": "reference", "cores": 1, "templates": len(ids), "candidates_offered": naln, "accepted": nacc, "seconds": dt,
                      "templates_per_s": len(ids) / dt, "load_reads_s": t_load}))


def e2e_cli(workload, codes, lens, threads):
    """SURVEY.md §8d's second denominator: wall clock of the drop-in binary (FASTA split, H2D/D2H, kernels, text output), -j 0 and
    -j 1 -g 1, on the same workload.  The FASTA sits in /dev/shm (or the temp dir) so that no disk speed enters."""
    from mecat_amd import workload as W
    exe = os.path.join(ROOT, "mecat_amd", "bin", "mecat2pw")
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="mecat_e2e_", dir=base)
    res = {"threads": threads, "dir": base or "tmp"}
    try:
        fa = os.path.join(d, "reads.fa")
        W.write_fasta(fa, codes, lens)
        res["fasta_bytes"] = os.path.getsize(fa)
        for task, name in ((0, "j0"), (1, "j1")):
            o = os.path.join(d, "out." + name)
            best = None
            for rep in range(2):                     # the first run also pays the one-off page-in of the device libraries
                w = os.path.join(d, "w_%s_%d" % (name, rep))
                # the driver takes a second or two to take back the ~60 GB of a process that has just exited; a CLI started inside
                # that window spends 20-30 ms per GB in hipMalloc (0.6 -> 1.5 s wall): the runs are spaced so that each measures itself
                time.sleep(3.0)
                t0 = time.time()
                p = subprocess.run([exe, "-j", str(task), "-d", fa, "-o", o, "-w", w, "-t", str(threads), "-g", "1"],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, MECAT_TRACE="1"))
                wall = time.time() - t0
                log("e2e -j %d run %d: %.3f s; %s" % (task, rep, wall, "; ".join(
                    l.strip() for l in p.stderr.splitlines() if "[trace]" in l or "takes" in l)))
                if p.returncode != 0:
                    res["error"] = p.stderr[-300:]
                    return res
                best = wall if best is None else min(best, wall)
                subprocess.run(["rm", "-rf", w])
            nl, ab = 0, 0
            with open(o) as f:
                for ln in f:
                    nl += 1
                    if task == 1:
                        x = ln.split("\t", 8)
                        ab += int(x[6]) - int(x[5])
            if task == 0:
                res["j0"] = {"wall_s": best, "candidates": nl, "candidates_per_s": nl / best}
            else:
                res["j1"] = {"wall_s": best, "overlaps": nl, "overlaps_per_s": nl / best, "aligned_gbase_per_s": ab / 1e9 / best}
    finally:
        subprocess.run(["rm", "-rf", d])
    return res


def src_digest():
    """sha256 over the kernel sources: the committed PMC numbers are only quoted while they describe THIS code"""
    import hashlib
    h = hashlib.sha256()
    cs = os.path.join(ROOT, "mecat_amd", "csrc")
    for f in sorted(os.listdir(cs)):
        h.update(open(os.path.join(cs, f), "rb").read())
    return h.hexdigest()[:16]


def issue_ceilings():
    """live calibration (mecat_amd/bin/valu_peak, ~1 s): wave64 instruction issue rates of this device, G wave-instr/s"""
    exe = os.path.join(ROOT, "mecat_amd", "bin", "valu_peak")
    try:
        p = subprocess.run([exe, "3000", "v_add_u32,v_alignbit_b32,s_add_u32,ds_read_b32,mix_dw_rowbody"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=120)
        r = json.loads(p.stdout)["rates"]
        # mix_dw_rowbody: independent streams with the d-path row body's own instruction mix (7 four-cycle + 4 two-cycle VALU, 4 SALU,
        # 1 LDS read per 16); its VALU share is what a kernel of this mix can issue at best, by resident waves per SIMD (1, 2, 4, 8)
        return {"valu_2cycle_class": max(r["v_add_u32"]), "valu_4cycle_class": max(r["v_alignbit_b32"]), "salu": max(r["s_add_u32"]),
                "lds": max(r["ds_read_b32"]), "valu_dw_mix": max(r["mix_dw_rowbody"]) * 11 / 16,
                "valu_dw_mix_by_waves_per_simd": {str(w): v * 11 / 16 for w, v in zip((1, 2, 4, 8), r["mix_dw_rowbody"])},
                "source": "mecat_amd/bin/valu_peak, live"}
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)[:200]}


def simulate_ranks(args, M, W, ctx, vol, params, n, ont, one_step, stream, d_counts):
    """no multi-GPU node has been available to any round, so the scaling estimate is turned into a measurement as far as
    one GPU allows.  For P in --simulate-ranks and every rank r < P: a communicator WITHOUT transport (mhip_comm_init_solo) makes the library's
    own sharded calls — mhip_index_build_sharded, mhip_seed_reads_sharded, mhip_align_sharded, the calls bench.py --gpus P and the driver
    make — do exactly rank r's share on the device: its key range of the index, its chunks of 500 reads, the pack / scatter kernels of the
    exchanges.  Reported: per-rank wall ms per phase (second of two runs), max / mean imbalance, the bytes every rank would put on the links,
    and the compute-only bound  T(1 GPU) / max over ranks  for both ways of building the index.  What is NOT in it: link time and the
    RCCL calls themselves."""
    import torch
    CH = M.SHARD_CHUNK

    def wall(f):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, r

    one_step()
    base = np.mean([one_step()["ms"] for _ in range(3)], axis=0)
    ncand = int(d_counts.sum().item())
    idx_full = M.Index(ctx, vol)
    out = {"what": "every rank's share of the sharded calls, alone on one MI355X, no transport (mhip_comm_init_solo)", "workload": args.workload,
           "one_gpu_phase_ms": {"index": float(base[0]), "seed": float(base[1]), "align": float(base[2])}, "one_gpu_step_ms": float(base.sum()),
           "candidates": ncand, "P": {}}
    for P in [int(x) for x in args.simulate_ranks.split(",") if x]:
        ranks = []
        for r in range(P):
            cm = M.Comm(ctx, P, r, solo=True)
            ms_i = ms_s = ms_a = 0.0
            b_i = b_s = b_a = 0
            kst = {}
            for rep in range(3):      # (the fastest of three: the first run of a rank also sizes its scratch buffers)
                if rep == 2 and r == 0:
                    ctx.set_profiling(True)
                    ctx.reset_stats()
                s0 = cm.bytes_sent()
                t_i, ix = wall(lambda: cm.index_build_sharded(vol))
                ix.free()
                s1 = cm.bytes_sent()
                t_s, _ = wall(lambda: cm.seed_reads_sharded(idx_full, vol, vol, 0, n, params, chunk=CH, cell_shift=0, host=False))
                s2 = cm.bytes_sent()
                t_a, _ = wall(lambda: cm.align_sharded(vol, vol, params.min_align_size, tech=ont, host=False)) if not args.no_align else (0.0, None)
                s3 = cm.bytes_sent()
                ms_i, ms_s, ms_a = (t_i, t_s, t_a) if rep == 0 else (min(ms_i, t_i), min(ms_s, t_s), min(ms_a, t_a))
                if rep == 2 and r == 0:
                    kst = {k: round(v[1], 3) for k, v in sorted(ctx.kernel_stats().items(), key=lambda kv: -kv[1][1]) if v[1] >= 0.05}
                    ctx.set_profiling(False)
                b_i, b_s, b_a = (s1 - s0) // max(1, P - 1), (s2 - s1) // max(1, P - 1), (s3 - s2) // max(1, P - 1)
            ranks.append({"rank": r, "reads": M.lib().mhip_shard_local_count(0, n, CH, 0, r, P), "candidates": cm.local_jobs(),
                          "index_slice_ms": ms_i, "seed_ms": ms_s, "align_ms": ms_a, **({"kernel_ms_of_one_pass": kst} if kst else {}),
                          "bytes_to_each_peer": {"index_slice": int(b_i), "candidates": int(b_s), "results": int(b_a)}})
            cm.close()
        seed = np.array([x["seed_ms"] for x in ranks]); aln = np.array([x["align_ms"] for x in ranks]); isl = np.array([x["index_slice_ms"] for x in ranks])
        tot_rep = float(base[0]) + seed + aln
        tot_shd = isl + seed + aln
        recv = {k: int(sum(x["bytes_to_each_peer"][k] for x in ranks)) for k in ("index_slice", "candidates", "results")}
        out["P"][str(P)] = {
            "ranks": ranks,
            "sum_of_rank_candidates": int(sum(x["candidates"] for x in ranks)),
            "imbalance_max_over_mean": {"seed": float(seed.max() / seed.mean()), "align": float(aln.max() / aln.mean()) if aln.mean() > 0 else None,
                                        "index_slice": float(isl.max() / isl.mean())},
            "slowest_rank_ms": {"index_rebuilt_on_every_rank": float(tot_rep.max()), "index_in_key_range_shards": float(tot_shd.max())},
            "compute_only_speedup_bound": {"index_rebuilt_on_every_rank": float(base.sum() / tot_rep.max()), "index_in_key_range_shards": float(base.sum() / tot_shd.max())},
            "bytes_received_per_rank_about": {k: int(v * (P - 1) / P) for k, v in recv.items()},
            "link_time_at_153_GBps_per_link_ms": {k: float(v / P / 153e9 * 1e3) for k, v in recv.items()},
            "note": "bytes_received_per_rank_about = what the P - 1 peers send to one rank; link_time: each peer's share over its own xGMI link (7 links x "
                    "153 GB/s, MI355X_MICROARCH.md), links in parallel — arithmetic, not measured",
        }
        log("[simulate] P=%d: slowest rank %.1f ms (index rebuilt) / %.1f ms (index sharded), 1 GPU %.1f ms -> compute-only bound %.2fx / %.2fx; imbalance seed %.3f align %.3f"
            % (P, tot_rep.max(), tot_shd.max(), base.sum(), base.sum() / tot_rep.max(), base.sum() / tot_shd.max(), seed.max() / seed.mean(),
               (aln.max() / aln.mean()) if aln.mean() > 0 else 0.0))
    idx_full.free()
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-full", type=int, default=1, help="1 (default): also time the unmodified reference -j 0 on the full workload on this host, once")
    ap.add_argument("--no-align", action="store_true", help="-j 0 only (index + candidates)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed cns_realign / xdrop_extend measurements")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end CLI leg (FASTA -> .can / .m4 wall clock)")
    ap.add_argument("--e2e", action="store_true", help="multi-volume workloads (config3, config5): ONLY the end-to-end leg — the drop-in binary on the workload's FASTA, "
                                                      "-j 0 and -j 1 -g 1, wall clock with the stages traced (prints its own JSON line)")
    ap.add_argument("--stats", default="", help="write per-kernel stats JSON here (rank 0)")
    ap.add_argument("--simulate-ranks", default="", help="e.g. 2,4,8: on ONE GPU, run every rank's share of the sharded calls alone (no transport) and report "
                                                         "per-rank times, imbalance, link bytes and the compute-only speed-up bound (prints its own JSON line)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher around it (no WORLD_SIZE in the environment): start N ranks of this script, one per
    # GPU, through torch.distributed.run on the loop-back address, and become that launcher (the run must not
    # quietly fall back to one GPU).  Under a launcher (WORLD_SIZE set) the ranks must be exactly --gpus.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        log("[bench] --gpus %d without a launcher: exec %s" % (args.gpus, " ".join(cmd)))
        os.execv(sys.executable, cmd)

    from mecat_amd import workload as W
    if args.workload == "config4":      # BASELINE configs[3]: mecat2cns' re-alignment + accept stage on config 2's candidates (bench_config4.py)
        import bench_config4
        bench_config4.run(args)
        return
    if args.workload in W.GRIDS:        # multi-volume workloads (config 3, a grid cell of config 5): bench_grid.py, same contract line
        import bench_grid
        bench_grid.run(args)
        return

    import torch
    import torch.distributed as dist
    from mecat_amd import hip as M

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MECAT_BENCH_BACKEND=gloo is a test hook: several ranks sharing the GPUs that exist (ranks are folded onto the visible
    # devices), to exercise the sharded path on a one-GPU box.  The measured configuration is always nccl (= RCCL).
    backend = os.environ.get("MECAT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d (or without a launcher: "
                         "the script starts its own ranks)" % (args.gpus, world, args.gpus))
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but %d visible GPUs (one rank per GPU; MECAT_BENCH_BACKEND=gloo folds ranks onto the GPUs "
                         "that exist, as a test hook only)" % (world, torch.cuda.device_count()))
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    n, L, err, G, seed, ont = W.CONFIGS[args.workload]
    t0 = time.time()
    codes, lens = W.synth_reads(n, L, err, G, seed, ont)
    pac, offs, num_bases = W.pack_volume(codes, lens)
    if rank != 0 or world > 1 or (args.no_e2e and args.no_cpu):
        del codes
    if rank == 0:
        log("[bench] %s: %d reads, %d bases incl. pads, generated+packed in %.1fs" % (args.workload, n, num_bases, time.time() - t0))

    # a dedicated (non-null) stream shared by torch (events, RCCL) and the library's launches
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = M.Context(local_rank, stream.cuda_stream)
    assert stream.cuda_stream != 0
    vol = M.Volume(ctx, pac, offs, num_bases, 0)
    del pac
    params = M.default_params(ont)
    maxc = params.maxc

    # N > 1: the (only) grid cell is sharded over the ranks by the library itself — chunks of 500 reads, chunk c -> rank c mod P
    # (mecat_hip.h: mhip_seed_reads_sharded / mhip_align_sharded, the calls the mecat2pw driver makes in its multi-GPU mode):
    # every rank rebuilds the index, seeds and extends the reads of its own chunks, candidate lists and results are exchanged
    # count-then-payload over the library's own RCCL communicator.  torch.distributed only carries the unique id, the barriers
    # and the max-over-ranks clock of the bench contract.
    comm = None
    CH = M.SHARD_CHUNK
    if world > 1:
        if backend == "nccl":
            box = [M.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = M.Comm(ctx, world, rank, unique_id=box[0])
        else:                       # test hook (ranks folded onto one GPU): the library's host-file transport
            box = [tempfile.mkdtemp(prefix="mecat_bench_comm_") if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = M.Comm(ctx, world, rank, hostfile_dir=box[0], run_id="bench")
        comm.barrier()
        n_local = M.lib().mhip_shard_local_count(0, n, CH, 0, rank, world)
    else:
        n_local = n
        d_cands = torch.zeros((n, maxc, 12), dtype=torch.int32, device=dev)
        d_counts = torch.zeros((n,), dtype=torch.int32, device=dev)
        d_jobs = torch.empty((n * maxc + maxc, 5), dtype=torch.int32, device=dev)
        d_res = torch.empty((d_jobs.shape[0], 8), dtype=torch.int32, device=dev)

    # the index of the cell's reference volume: from four ranks on built by the ranks together (key-range shards + all-gather,
    # mhip_index_build_sharded), below that rebuilt on every rank; MECAT_HIP_INDEX_SHARD=0 / 1 overrides, as in the driver
    # (the library decides by measurement: mhip_index_build_auto builds the first table both ways, timed barrier to barrier, and keeps the
    # faster way — done here, before the warm-up, so that no timed step carries the measurement)
    index_how = {"chosen": "replicated", "measured": False}
    if comm is not None:
        i0, index_how = comm.index_build_auto(vol)
        i0.free()
        index_how["bytes_received_measuring"] = comm.bytes_received()
        tr0, nr0 = comm.info()
        if tr0 == 0 and nr0 != world:      # the bench contract: N ranks means N GPUs in one RCCL communicator
            raise SystemExit("[bench] rank %d: the RCCL communicator has %d ranks, --gpus %d asked for %d" % (rank, nr0, args.gpus, world))
    keep = {}
    released = False

    def one_step():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        h0 = time.perf_counter()
        ev[0].record(stream)
        idx = comm.index_build_auto(vol)[0] if comm is not None else M.Index(ctx, vol)
        ev[1].record(stream)
        keep["num_kmers"] = idx.num_kmers
        njobs = 0
        if comm is None:
            M.seed_reads_strided_dev(ctx, idx, vol, vol, 0, 1, n, params, d_cands.data_ptr(), d_counts.data_ptr())
            ev[2].record(stream)
            if not args.no_align:
                njobs = M.jobs_from_candidates_dev(ctx, d_cands.data_ptr(), d_counts.data_ptr(), n, maxc, 0, 1, 0, 0, 1, d_jobs.data_ptr())
                M.align_candidates_dev(ctx, vol, vol, d_jobs.data_ptr(), njobs, params.min_align_size, d_res.data_ptr())
        else:
            comm.seed_reads_sharded(idx, vol, vol, 0, n, params, chunk=CH, cell_shift=0, host=False)      # seeding + candidate all-gather
            ev[2].record(stream)
            if not args.no_align:
                _, njobs = comm.align_sharded(vol, vol, params.min_align_size, tech=ont, host=False)      # extension + result all-gather
        ev[3].record(stream)
        idx_handle = idx
        stream.synchronize()
        h1 = time.perf_counter()
        out = {"njobs": njobs, "ms": [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]}
        keep["njobs"] = njobs
        idx_handle.free()
        out["host_ms"] = [(h1 - h0) * 1e3, (time.perf_counter() - h1) * 1e3]
        return out

    if args.simulate_ranks:
        if world != 1:
            raise SystemExit("--simulate-ranks runs on one GPU (it plays the ranks one after the other)")
        simulate_ranks(args, M, W, ctx, vol, params, n, ont, one_step, stream, d_counts)
        vol.free()
        ctx.close()
        return

    ctx.set_profiling(True)         # on during warm-up too, so the event pool exists before the timed region
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
    ctx.set_profiling(False)

    # result statistics, after the timed region.  N = 1: read from the last step's output buffers.  N > 1: one more (untimed) pass
    # of the same deterministic calls with the tables brought to the host — every rank holds the complete tables.
    exch = None
    if comm is None:
        ncand = int(d_counts.sum().item())
        aln_ok = aligned_bases = 0
        if not args.no_align:
            r = d_res[:keep["njobs"]]
            ok = r[:, 0] != 0
            aln_ok = int(ok.sum().item())
            aligned_bases = int(((r[:, 2] - r[:, 1]).to(torch.int64) * ok).sum().item())
    else:
        tr, nr = comm.info()
        exch = {"bytes_received_per_step": (comm.bytes_received() - index_how.get("bytes_received_measuring", 0)) / max(1, args.steps + args.warmup),
                "transport": "rccl" if tr == 0 else "host files (test hook)", "rccl_ranks": nr,
                # HIP events around every exchange on the launch stream (rank 0's view), per step
                "ms": kstats.get("xg_exchange", (0, 0.0))[1] / args.steps, "calls_per_step": kstats.get("xg_exchange", (0, 0.0))[0] / args.steps,
                "index_exchange_ms": kstats.get("xg_exchange_index", (0, 0.0))[1] / args.steps,
                "index_build_kernels_ms": sum(v[1] for k, v in kstats.items() if k.startswith(("ix_", "idx"))) / args.steps,
                "index_rebase_slots_ms": sum(v[1] for k, v in kstats.items() if k.startswith("xg_index")) / args.steps}
        idx = comm.index_build_auto(vol)[0]
        _, h_cnt = comm.seed_reads_sharded(idx, vol, vol, 0, n, params, chunk=CH, cell_shift=0, host=True)
        ncand = int(h_cnt.sum())
        aln_ok = aligned_bases = 0
        if not args.no_align:
            h_res, nj = comm.align_sharded(vol, vol, params.min_align_size, tech=ont, host=True)
            assert nj == ncand
            okm = h_res["ok"] != 0
            aln_ok = int(okm.sum())
            aligned_bases = int((h_res["query_end"].astype(np.int64) - h_res["query_start"])[okm].sum())
        idx.free()
        exch["records_per_step"] = ncand
        exch["index_build"] = index_how          # how the cell's table is built, and the two measured times the choice rests on
        # every rank's own phase times (HIP events on its stream, mean over the timed steps) and wall time: where a short curve fell short
        mine = {"rank": rank, "local_candidates": comm.local_jobs(), "bytes_sent_total": comm.bytes_sent(),
                "phase_ms": [float(x) for x in np.mean([o["ms"] for o in outs], axis=0)],
                "host_ms_per_step": float(np.mean([sum(o["host_ms"]) for o in outs]))}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        exch["per_rank"] = allr
        exch["note"] = "count-then-payload: 4 B per read + 48 B per candidate (+ 32 B per extension result) from the P - 1 peers"

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        phase = np.mean([o["ms"] for o in outs], axis=0)
        print("[bench] host ms/step: submit+wait=%.1f, free=%.1f" % tuple(np.mean([o["host_ms"] for o in outs], axis=0)), file=sys.stderr)
        # dominant kernel by summed HIP-event time on the launch stream
        dom = max(kstats.items(), key=lambda kv: kv[1][1]) if kstats else ("none", (1, 0.0))
        dname, (dl, dms) = dom
        per_step = lambda k: counters[k] / args.steps          # noqa: E731
        lookups, hits, cands_c = per_step("lookups"), per_step("hits"), per_step("candidates")
        strands = 2 * n_local
        # algorithmic bytes per step, this rank's share
        N = num_bases
        b_idx = 2 * (N / 4) + 3 * 4 * (1 << 26) + 4 * keep["num_kmers"]      # §8d: two volume passes, the 4^13 table x3, the kept positions
        if comm is None:
            local_bases = int(lens.astype(np.int64).sum())
        else:
            from mecat_amd import shard as S
            local_bases = int(lens.astype(np.int64)[S.local_reads(0, n, CH, 0, rank, world)].sum())
        b_seed = (local_bases * 2) / 4 + 8 * lookups + 4 * hits + 48 * cands_c
        b_aln = per_step("aligned_bases") * 2 / 4 + 32 * cands_c          # this rank's candidates
        phase_of = {"idx": b_idx, "seed": b_seed, "dw": b_aln}
        pk = "idx" if dname.startswith("idx") else ("dw" if dname.startswith("dw") else "seed")
        launches_per_step = max(1, dl // args.steps)
        alg_per_launch = phase_of[pk] / launches_per_step
        avg_ms = dms / max(1, dl)
        achieved = alg_per_launch / 1e9 / (avg_ms / 1e3) if avg_ms > 0 else 0.0
        roof = {"bound": "hbm", "kernel": dname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": alg_per_launch, "avg_launch_ms": avg_ms, "launches": dl,
                "phase_bytes_per_step": {k: float(v) for k, v in phase_of.items()},
                "note": "algorithmic bytes of the kernel's phase (SURVEY.md §8d) / launches; scratch/sort traffic not counted"}
        # HBM bytes and SQ instruction counts of the same kernel come from committed rocprofv3 --pmc passes of this command (PMC
        # counters cannot be read from inside the run).  They are quoted only while profiles/*_pmc_source.json records the digest
        # of the kernel sources that are compiled now; otherwise traffic stays null and the line says why.
        prof_dir = os.path.join(ROOT, "profiles")
        digest = src_digest()
        roof["kernel_source_digest"] = digest
        pmc_ok = False
        try:
            sfile = sorted(f for f in os.listdir(prof_dir) if re.match(r"r\d+_pmc_source\.json$", f))[-1]
            meta = json.load(open(os.path.join(prof_dir, sfile)))
            pmc_ok = meta.get("kernel_source_digest") == digest
            if not pmc_ok:
                roof["traffic_note"] = "committed PMC passes (profiles/%s) describe other kernel sources (%s): not quoted" % (sfile, meta.get("kernel_source_digest"))
            tag = sfile[: -len("_pmc_source.json")]
        except (OSError, IndexError, ValueError):
            roof["traffic_note"] = "no committed PMC pass with a source digest"
            tag = None
        if pmc_ok and args.workload == "config2" and world == 1:
            try:
                t = json.load(open(os.path.join(prof_dir, tag + "_hbm_traffic.json"))).get(dname)
                if t:
                    roof["traffic"] = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
                    roof["traffic_source"] = "profiles/%s_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command, same sources)" % tag
            except (OSError, ValueError, KeyError):
                pass
        # dw is integer VALU work on LDS-resident state (HBM fraction tiny by construction): price it against the
        # instruction-issue rates of this device, calibrated live by mecat_amd/bin/valu_peak.  CDNA4 issues a wave64 VALU
        # instruction in 2 cycles only for the simple 32-bit class (add/sub/logic/mov/lshr, two waves of a SIMD co-issuing);
        # min/max, shifts left, compares, selects, every VOP3 / DPP / SGPR-operand form take 4.
        if pk == "dw" and world == 1:
            ceil = issue_ceilings()
            vi = {"unit": "G wave-instr/s", "ceilings": ceil}
            if pmc_ok and args.workload == "config2" and avg_ms > 0 and "error" not in ceil:
                try:
                    im = json.load(open(os.path.join(prof_dir, tag + "_instruction_mix.json"))).get(dname)
                    ach = im["valu_insts_per_launch"] / (avg_ms / 1e3) / 1e9
                    # Ceiling = the two measured issue rates of this device (valu_peak, live: independent streams of plain 32-bit adds = the
                    # 2-cycle class, of v_alignbit = the 4-cycle class, best over 1-8 waves per SIMD), weighted harmonically by the share of
                    # 4-cycle-class instructions in the kernel's d-row loops (tools/isa_mix.py --row-loops on hipcc's assembly; the loops are
                    # 92 % of the kernel's time, profiles/r03_dw_row_breakdown.md).  Never clamped: a kernel above its "ceiling" means the
                    # ceiling is wrong.  The mix stream's own rate (mix_dw_rowbody) is kept beside it for comparison.
                    f4 = im.get("valu_4cycle_fraction_row_loops", im.get("valu_4cycle_fraction_static"))
                    vi.update({"achieved": ach, "salu_achieved": im["salu_insts_per_launch"] / (avg_ms / 1e3) / 1e9,
                               "source": "profiles/%s_instruction_mix.json (SQ_INSTS_VALU / SQ_INSTS_SALU per launch, same sources)" % tag,
                               "mix_stream_rate": ceil["valu_dw_mix"]})
                    if f4 is not None:
                        # The ceiling is the data-sheet issue rate of the kernel's own instruction mix: 1024 SIMDs x 2.4 GHz / 4 for the
                        # 4-cycle class and / 2 for the plain 32-bit class, weighted by the share f4 of 4-cycle instructions in the d-row
                        # loops.  The class rates valu_peak measures live stay in `ceilings` as calibration, but are not used as `peak`:
                        # its 2-cycle stream reaches 74 % of nominal and the kernel's mix beats a ceiling made from it (frac 1.10 in
                        # round 3) — a ceiling the kernel beats is not a ceiling.
                        nominal = 1.0 / (f4 / (1024 * 2.4 / 4) + (1.0 - f4) / (1024 * 2.4 / 2))
                        measured = 1.0 / (f4 / ceil["valu_4cycle_class"] + (1.0 - f4) / ceil["valu_2cycle_class"])
                        vi.update({"peak": nominal, "frac": ach / nominal, "valu_4cycle_fraction_row_loops": f4,
                                   "measured_class_rate_mix": measured})
                        vi["note"] = ("peak = 1 / (f4 / R4 + (1 - f4) / R2) with the data-sheet rates R4 = 1024 SIMDs x 2.4 GHz / 4, R2 = ... / 2 and f4 = the share "
                                      "of 4-cycle-class instructions in dw_extend2's d-row loops; measured_class_rate_mix = the same formula on valu_peak's live "
                                      "class rates (calibration only: the kernel's mix outruns it); 8 waves per SIMD (64 VGPRs, 19 KB LDS per 4 waves)")
                except (OSError, ValueError, KeyError, TypeError):
                    pass
            roof["valu_issue"] = vi
        if pk == "dw":
            roof["dw_cells_per_s"] = per_step("dw_cells") / (phase[2] / 1e3) if phase[2] > 0 else None
            roof["dw_snake_bases_per_s"] = per_step("snake_bases") / (phase[2] / 1e3) if phase[2] > 0 else None
        line = {
            "metric": "candidate overlaps/sec", "value": ncand / (ms_step / 1e3), "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: %d reads x %d bp @ %.0f%% error, genome %d, seed %d, k=13, all-vs-all, -j 1 (index+seed+dw)"
                                   % (args.workload, n, L, err * 100, G, seed), "reads": n, "bases": int(num_bases),
                       "parallelism": "1 GPU" if world == 1 else "grid cell sharded: chunks of %d reads, chunk c -> rank c mod %d; RCCL count-then-payload all-gather; index %s"
                                      % (CH, world, "built in k-mer key-range shards + all-gather" if index_how["chosen"] == "sharded" else "rebuilt on every rank")},
            "candidates": ncand, "overlaps_ok": aln_ok, "aligned_gbase_per_s": aligned_bases / 1e9 / (ms_step / 1e3),
            "overlaps_per_s": aln_ok / (ms_step / 1e3),
            "phase_ms": {"index": float(phase[0]), "seed": float(phase[1]), "align": float(phase[2])},
            "candidates_per_s_index_seed_phases": ncand / ((phase[0] + phase[1]) / 1e3),
            "aligned_gbase_per_s_align_phase": (aligned_bases / 1e9 / (phase[2] / 1e3)) if phase[2] > 0 else None,
            "counters_per_step": {k: v / args.steps for k, v in counters.items()},
            "roofline": roof,
        }
        if exch:
            line["exchange"] = exch
        # not part of the metric: the mecat2cns re-aligner (full aligned strings as 2-bit columns) on the first
        # 200 000 candidates of this run, device resident
        if world == 1 and not args.no_align and not args.no_extras and keep["njobs"] > 0:
            try:
                nj, cap = min(200000, keep["njobs"]), 32768
                c_res = torch.empty((nj, 16), dtype=torch.int32, device=dev)
                c_ops = torch.empty((nj, 2, cap // 16), dtype=torch.int32, device=dev)
                M.lib().mhip_cns_align_candidates_dev.restype = C.c_int
                tc = []
                for _ in range(2):
                    torch.cuda.synchronize()
                    c0 = time.perf_counter()
                    rc = M.lib().mhip_cns_align_candidates_dev(ctx.h, vol.h, vol.h, d_jobs.data_ptr(), nj, 0.15, params.min_align_size, cap,
                                                               c_res.data_ptr(), c_ops.data_ptr())
                    torch.cuda.synchronize()
                    tc.append(time.perf_counter() - c0)
                if rc == 0:
                    okc = c_res[:, 0] != 0
                    line["cns_realign"] = {"jobs": nj, "seconds": tc[-1], "alignments_per_s": nj / tc[-1], "ok": int(okc.sum().item()),
                                           "aligned_gbase_per_s": float(((c_res[:, 2] - c_res[:, 1]).to(torch.int64) * okc).sum().item()) / 1e9 / tc[-1]}
                del c_res, c_ops
            except Exception as e:          # never let the extra measurement break the contract line
                log("[bench] cns_realign skipped: %r" % (e,))
        # not part of the metric: mecat2cns' per-template stage up to the consensus table (BASELINE config 4's front half, SURVEY.md row
        # N1) on this run's candidates — candidates of the first templates as mecat2cns would load them, <= 200 per template re-aligned
        # in one launch, accept decisions replayed, gap-normalised strings of the accepted alignments rebuilt (mhip_cns_accept_templates)
        if world == 1 and not args.no_align and not args.no_extras and keep["njobs"] > 0:
            try:
                h_cnt = d_counts.cpu().numpy()
                h_cands = d_cands.cpu().numpy().view(M.CAND_DTYPE).reshape(n, maxc)
                ec = W.ext_candidates_from_table(h_cands, h_cnt, lens)
                rec, tb, ids = W.cns_templates(ec, n)
                T = min(len(ids), 10000)
                rec_orig = np.ascontiguousarray(rec[: tb[T]]).copy()      # (the call sorts its records in place; the reference leg below wants them as loaded)
                rec_s = rec_orig.copy()
                c0 = time.perf_counter()
                acc, strs, nja = M.cns_accept_templates(ctx, vol, None, rec_s, tb[: T + 1], ont, 2000 if not ont else 500, 0.9 if not ont else 0.4,
                                                        threads=min(64, os.cpu_count() or 1))
                dtc = time.perf_counter() - c0
                tbases = int(lens[ids[:T]].astype(np.int64).sum())
                line["cns_accept"] = {"workload": "BASELINE config 4 (mecat2cns' per-template stage on config 2's candidates), first %d templates; the whole-config "
                                                  "line is `bench.py --workload config4`" % T,
                                      "templates": T, "alignments": int(nja), "accepted": int(len(acc)), "seconds": dtc,
                                      "templates_per_s": T / dtc, "alignments_per_s": nja / dtc, "template_gbase_per_s": tbases / 1e9 / dtc,
                                      "aligned_string_bytes": len(strs),
                                      "note": "BASELINE config 4 up to the consensus table: sort + <= 200 re-alignments per template on the GPU + accept replay + "
                                              "normalised strings built on the GPU, accept decisions replayed on %d host threads; the consensus table / POA after it "
                                              "stay with mecat2cns" % min(64, os.cpu_count() or 1)}
                # parity: the first templates against what the UNMODIFIED consensus_one_read_can_pacbio accepted on the same records
                # (tests/golden/cns_config2.npz, written by tests/golden/make_golden_cns_config2.py in the build container)
                try:
                    import hashlib
                    g4 = np.load(os.path.join(ROOT, "tests", "golden", "cns_config2.npz"))
                    Tg = int(g4["par"][1])
                    if args.workload == "config2" and Tg <= T and np.array_equal(g4["ids"], ids[:Tg]) and np.array_equal(g4["tmpl_begin"], tb[: Tg + 1]):
                        sel = acc[acc["template_index"] < Tg]
                        ok = np.array_equal(np.bincount(sel["template_index"], minlength=Tg), g4["nacc"]) and \
                            np.array_equal(np.stack([sel["soff"], sel["send"], sel["aln_size"]], axis=1), g4["meta"])
                        first = np.concatenate([[0], np.cumsum(g4["nacc"])])
                        for t in range(Tg):
                            if not ok or g4["nacc"][t] == 0:
                                continue
                            a = sel[first[t]: first[t + 1]]
                            lo, hi = int(a["str_offset"][0]), int(a["str_offset"][-1]) + 2 * (int(a["aln_size"][-1]) + 1)
                            ok = ok and hashlib.sha256(strs[lo:hi]).hexdigest() == str(g4["sha"][t])
                        line["cns_accept"]["parity_vs_reference"] = {"templates": Tg, "accepted": int(g4["nacc"].sum()), "identical": bool(ok)}
                except Exception as e:      # noqa: BLE001
                    line["cns_accept"]["parity_vs_reference"] = {"error": repr(e)[:200]}
                keep["cns"] = (rec_orig, tb.copy(), ids.copy())
                del h_cands, ec, rec, rec_s, acc, strs
            except Exception as e:      # noqa: BLE001
                log("[bench] cns_accept skipped: %r" % (e,))
        # not part of the metric either: the X-drop aligner (nanopore mode) on the first 100 000 candidates
        if world == 1 and not args.no_align and not args.no_extras and keep["njobs"] > 0:
            try:
                nj = min(100000, keep["njobs"])
                x_res = torch.empty((nj, 8), dtype=torch.int32, device=dev)
                M.lib().mhip_xalign_candidates_dev.restype = C.c_int
                tx = []
                for _ in range(2):
                    torch.cuda.synchronize()
                    c0 = time.perf_counter()
                    rc = M.lib().mhip_xalign_candidates_dev(ctx.h, vol.h, vol.h, d_jobs.data_ptr(), nj, params.min_align_size, x_res.data_ptr())
                    torch.cuda.synchronize()
                    tx.append(time.perf_counter() - c0)
                if rc == 0:
                    okx = x_res[:, 0] != 0
                    line["xdrop_extend"] = {"jobs": nj, "seconds": tx[-1], "alignments_per_s": nj / tx[-1], "ok": int(okx.sum().item()),
                                            "aligned_gbase_per_s": float(((x_res[:, 2] - x_res[:, 1]).to(torch.int64) * okx).sum().item()) / 1e9 / tx[-1]}
                del x_res
            except Exception as e:
                log("[bench] xdrop_extend skipped: %r" % (e,))
        if args.stats:
            json.dump({"kernels": {k: {"launches": v[0], "total_ms": v[1]} for k, v in kstats.items()}, "line": line},
                      open(args.stats, "w"), indent=1)
        try:
            dbg = []
            M.lib().mhip_debug_counter.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
            for slot in range(8, 13):
                v = C.c_int64()
                M.lib().mhip_debug_counter(ctx.h, slot, C.byref(v))
                dbg.append(v.value // args.steps)
            log("[bench] dw per step: %d units handed over to the second launch (which ran %d rows, %d blocks without an end)" % (dbg[0], dbg[1], dbg[4]))
        except Exception as e:
            log("[bench] no debug counters: %r" % (e,))
        log("[bench] kernel ms/step: " + ", ".join("%s=%.2f" % (k, v[1] / args.steps) for k, v in sorted(kstats.items(), key=lambda kv: -kv[1][1])))
        if world == 1:
            # the CLI leg is another process on the same GPU: give this process's device memory (tens of GB of scratch) back first
            vol.free()
            ctx.close()
            released = True
        if world == 1 and not args.no_e2e:
            try:
                line["e2e"] = e2e_cli(args.workload, codes, lens, min(32, os.cpu_count() or 1))
            except Exception as e:  # noqa: BLE001
                line["e2e"] = {"error": repr(e)[:300]}
        # not part of the metric: one grid cell of BASELINE config 5 (volumes 0 and 1 of the 2 M-read ONT-style set, -x 1: wide seeding filter +
        # the X-drop aligner) as its own bench line, compacted — candidates/s, X-drop alignments/s, phase times and the cell's `.can` hash
        # against the unmodified reference's (tests/golden/big.json); the full line is `bench.py --workload config5_cell`
        if world == 1 and not args.no_extras and not args.no_align and args.workload == "config2":
            try:
                time.sleep(3.0)
                c0 = time.time()
                p5 = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "config5_cell", "--steps", "2", "--warmup", "1", "--no-cpu"],
                                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=dict(os.environ, MECAT_BENCH_PARITY="1"))
                l5 = [ln for ln in p5.stdout.splitlines() if ln.startswith("{")]
                if p5.returncode != 0 or not l5:
                    line["config5_cell"] = {"error": p5.stderr[-300:]}
                else:
                    d5 = json.loads(l5[-1])
                    xk = {k: v for k, v in d5.get("kernel_ms_per_step", {}).items() if k.startswith(("xd_", "seed_filter_wide", "seed_emit"))}
                    line["config5_cell"] = {"workload": d5["config"]["workload"], "value": d5["value"], "unit": d5["unit"], "ms_per_step": d5["ms_per_step"],
                                            "phase_ms": d5["phase_ms"], "candidates": d5["candidates"], "xdrop_alignments_per_s": d5["candidates"] / (d5["phase_ms"]["align"] / 1e3),
                                            "overlaps_ok": d5["overlaps_ok"], "aligned_gbase_per_s": d5["aligned_gbase_per_s"], "kernel_ms_per_step": xk,
                                            "parity_vs_reference": d5.get("parity_vs_reference"), "roofline": {k: d5["roofline"].get(k) for k in ("kernel", "achieved", "frac", "traffic")},
                                            "wall_s_incl_generating_two_volumes": time.time() - c0}
            except Exception as e:  # noqa: BLE001
                line["config5_cell"] = {"error": repr(e)[:300]}
        # not part of the metric: mecat2canu's overlapper for corrected reads as the drop-in tool, end to end through its
        # own command line on 20 000 synthetic corrected reads (161 Mbases, 2 blocks as canu lays them out); the unmodified tool on the same
        # files, same machine, belongs to the CPU leg below
        if world == 1 and not args.no_extras and args.workload == "config2":
            try:
                ad = tempfile.mkdtemp(prefix="bench_asm_")
                blocks, abases = W.asm_blocks_layout(ad, 20000, 8000, 5_000_000, 2, 77)
                T = min(64, os.cpu_count() or 1)
                # (best of two runs after a pause: a process started on a device that has just taken back tens of GB from this one and from
                # the e2e runs can spend seconds in hipMalloc while the driver reclaims the pages)
                time.sleep(5.0)
                runs = [W.asm_tool_run(os.path.join(ROOT, "mecat_amd", "bin", "mecat2asmpw"), ad, T, 1, 2, env=dict(os.environ, MECAT_ASMPW_TIMES="1")) for _ in range(2)]
                alines, asecs, aerr = min(runs, key=lambda r: r[1])
                line["asm_overlap"] = {"tool": "mecat2asmpw -T%d -S1 -E2" % T, "reads": 20000, "bases": abases, "overlaps": len(alines), "seconds": asecs,
                                       "runs_s": [r[1] for r in runs],
                                       "mbases_per_s": abases / 1e6 / asecs, "stages": [ln for ln in aerr.splitlines() if ln.startswith("[mecat2asmpw]")][-1:] + [ln for ln in aerr.splitlines() if ln.startswith("[asm_seed]")][:3]}
                if not args.no_cpu:
                    ref = os.path.join(ROOT, "oracle", "_ref", "mecat2asmpw")
                    if os.path.exists(ref):
                        rlines, rsecs, _ = W.asm_tool_run(ref, ad, T, 1, 2)
                        line["asm_overlap"]["cpu_baseline"] = {"kind": "reference", "threads": T, "seconds": rsecs, "identical_output": rlines == alines,
                                                               "sample": "the unmodified tool (oracle/_ref/mecat2asmpw) on the same files, same host"}
                        line["asm_overlap"]["speedup_vs_reference_threads"] = rsecs / asecs
                shutil.rmtree(ad, ignore_errors=True)
            except Exception as e:  # noqa: BLE001
                line["asm_overlap"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu:
            try:
                c1 = cpu_baseline_config1()
                if c1:
                    line["cpu_baseline_config1"] = c1
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline_config1"] = {"error": repr(e)[:300]}
            if "cns" in keep and "cns_accept" in line:
                try:
                    cb = cns_cpu_baseline(codes, lens, *keep["cns"])
                    if cb:
                        line["cns_accept"]["cpu_baseline"] = cb
                        if "templates_per_s" in cb:
                            line["cns_accept"]["speedup_vs_one_reference_thread"] = line["cns_accept"]["templates_per_s"] / cb["templates_per_s"]
                except Exception as e:  # noqa: BLE001
                    line["cns_accept"]["cpu_baseline"] = {"error": repr(e)[:200]}
            try:
                line["cpu_baseline"] = cpu_baseline(args.workload, os.cpu_count() or 1)
                if args.cpu_full and args.workload == "config2":
                    try:
                        fs = cpu_full_size(args.workload, codes, lens, min(os.cpu_count() or 1, 64), ncand)
                        if fs:
                            line["cpu_baseline"]["full_size_same_host"] = fs
                            if fs.get("candidates_per_s"):
                                line["cpu_baseline"]["gpu_over_cpu_full_size_j0"] = (ncand / ((phase[0] + phase[1]) / 1e3)) / fs["candidates_per_s"]
                    except Exception as e:  # noqa: BLE001
                        line["cpu_baseline"]["full_size_same_host"] = {"error": repr(e)[:200]}
                try:      # context: the unmodified reference on the FULL workload, timed in the build container (tests/golden/big.json)
                    big = json.load(open(os.path.join(ROOT, "tests", "golden", "big.json"))).get(args.workload)
                    if big and "j0_seconds" in big:
                        line["cpu_baseline"]["full_size_reference"] = {
                            "threads": big.get("reference_threads"), "host": "build container (8 cores), not this host",
                            "j0_seconds": big["j0_seconds"], "candidates_per_s": big["can_lines"] / big["j0_seconds"],
                            "j1_seconds": big.get("j1_seconds"), "overlaps_per_s": big["m4_g1_lines"] / big["j1_seconds"] if big.get("j1_seconds") else None,
                            "source": "tests/golden/big.json (make_golden_big.py: whole mecat2pw runs incl. split and output)"}
                except Exception:  # noqa: BLE001
                    pass
            except Exception as e:  # the GPU number must survive a CPU-leg problem
                line["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": os.cpu_count(), "kind": "reference",
                                        "sample": "failed: %r" % (e,)}
        print(json.dumps(W.annotate_cpu_baseline(line)), flush=True)
    if not released:
        vol.free()
    if comm is not None:
        comm.barrier()
        comm.close()
    if not released:
        ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) == 5 and sys.argv[1] == "--cns-cpu-leg":
        cns_cpu_leg(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    if len(sys.argv) == 3 and sys.argv[1] == "--cns-cpu-leg":
        cns_cpu_leg(sys.argv[2])
    else:
        main()
