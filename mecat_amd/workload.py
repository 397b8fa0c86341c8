"""Synthetic workloads of BASELINE.json (the reference ships no data): deterministic read generator + 2-bit volume packer
(mecat_amd/tools/synth_reads.c via ctypes).  Host-side plumbing only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)

# name: (nreads, L, err, genome, seed, ont) — SURVEY.md §8d table
CONFIGS = {
    "config1": (1000, 10000, 0.15, 500_000, 1, 0),
    "config2": (100_000, 15000, 0.15, 50_000_000, 2, 0),
    "config3": (500_000, 12000, 0.15, 200_000_000, 3, 0),            # three volumes (2.14 + 2.14 + 2.04 Gbase)
    "config5": (2_000_000, 20000, 0.12, 1_300_000_000, 5, 1),        # nineteen volumes, ONT-style, -x 1
    "tinyset": (3000, 3000, 0.15, 300_000, 11, 0),                   # test input of the grid runner (three volumes at a 3.5 Mbase cut)
    "tinyset_ont": (3000, 3000, 0.12, 300_000, 12, 1),
}
MCS = 2_140_000_000      # bases per volume, pads included (common/split_database.h:6)

# multi-volume bench workloads: name -> (read set, volumes needed, grid cells (i, j) = (reference volume, query volume) in the driver's
# loop order, mecat2pw/pw_impl.cpp:859-879: one index build per row i, then the cells j = i ..)
GRIDS = {
    "config3": ("config3", 3, [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)], MCS),
    "config5_cell": ("config5", 2, [(0, 1)], MCS),                   # one off-diagonal cell of config 5's 19 x 19 grid: volumes 0 and 1
    "config5": ("config5", 19, [(i, j) for i in range(19) for j in range(i, 19)], MCS),      # BASELINE configs[4] whole: 190 cells, -x 1 -j 1
    "grid_tiny": ("tinyset", 3, [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)], 3_500_000),          # tests only
    "grid_tiny_ont": ("tinyset_ont", 3, [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)], 3_500_000),
    "grid_tiny_rows": ("tinyset", 5, [(i, j) for i in range(5) for j in range(i, 5)], 2_000_000),       # five volumes: rows mode with two ranks
}

_lib = None


def synth_lib():
    global _lib
    if _lib is None:
        p = os.path.join(_HERE, "lib", "libsynth.so")
        if not os.path.exists(p):
            subprocess.run(["make", "-s", "synth"], cwd=_ROOT, check=True)
        L = C.CDLL(p)
        L.synth_reads.restype = C.c_int64
        L.synth_reads.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_uint64, C.c_void_p, C.c_int64,
                                  C.c_void_p]
        L.synth_pack_volume.restype = C.c_int64
        L.synth_pack_volume.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.synth_write_fasta.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.synth_genome.argtypes = [C.c_void_p, C.c_int64, C.c_uint64]
        L.synth_reads_range.restype = C.c_int64
        L.synth_reads_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_uint64, C.c_void_p, C.c_int64,
                                        C.c_void_p]
        L.synth_pack_volume_mt.restype = C.c_int64
        L.synth_pack_volume_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def synth_reads(nreads, L, err, genome, seed, ont=0):
    lib = synth_lib()
    cap = nreads * (int(L * 1.25) + 64)
    bases = np.empty(cap, dtype=np.uint8)
    lens = np.empty(nreads, dtype=np.int32)
    tot = lib.synth_reads(genome, nreads, L, err, ont, seed, bases.ctypes.data, cap, lens.ctypes.data)
    if tot < 0:
        raise RuntimeError("synth_reads failed: %d" % tot)
    return bases[:tot], lens


def pack_volume(codes, lens):
    """-> (pac uint8[(num_bases+3)//4], offs int32[n,2], num_bases) — one volume (must stay below MCS = 2.14 Gbase)"""
    lib = synth_lib()
    n = len(lens)
    total = int(lens.astype(np.int64).sum()) + n
    assert total < 2_140_000_000, "more than one volume; split first"
    pac = np.zeros((total + 3) // 4, dtype=np.uint8)
    offs = np.zeros((n, 2), dtype=np.int32)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    nb = lib.synth_pack_volume(codes.ctypes.data, lens.ctypes.data, n, pac.ctypes.data, offs.ctypes.data)
    assert nb == total
    return pac, offs, total


def synth_volumes(name, num_vols, keep_codes=False, mcs=MCS):
    """the first num_vols volumes of a read set, cut as the reference's splitter cuts them (a read that would take the volume beyond MCS
    opens the next one, split_database.cpp:240) -> list of dicts {pac, offs, num_bases, start_read_id, lens[, codes]}.  Reads are made in
    batches (every read has its own RNG stream), so the first two volumes of config 5 cost 4.3 Gbase of generation, not 40."""
    n, L, err, G, seed, ont = CONFIGS[name]
    lib = synth_lib()
    g = np.empty(G, dtype=np.uint8)
    lib.synth_genome(g.ctypes.data, G, seed)
    cap = int(L * 1.25) + 64
    B = 32768
    buf = np.empty(B * cap, dtype=np.uint8)
    vols, cur_codes, cur_lens, cur_bases, first, start_id = [], [], [], 0, 0, 0

    def close():
        nonlocal cur_codes, cur_lens, cur_bases, start_id
        lens = np.concatenate(cur_lens) if cur_lens else np.zeros(0, np.int32)
        codes = np.concatenate(cur_codes) if cur_codes else np.zeros(0, np.uint8)
        total = int(lens.astype(np.int64).sum()) + len(lens)
        pac = np.zeros((total + 3) // 4, dtype=np.uint8)
        offs = np.zeros((len(lens), 2), dtype=np.int32)
        nb = lib.synth_pack_volume_mt(codes.ctypes.data, lens.ctypes.data, len(lens), pac.ctypes.data, offs.ctypes.data)
        assert nb == total == cur_bases
        v = {"pac": pac, "offs": offs, "num_bases": total, "start_read_id": start_id, "lens": lens}
        if keep_codes:
            v["codes"] = codes
        vols.append(v)
        start_id += len(lens)
        cur_codes, cur_lens, cur_bases = [], [], 0

    while first < n and len(vols) < num_vols:
        nb_ = min(B, n - first)
        lens = np.empty(nb_, dtype=np.int32)
        tot = lib.synth_reads_range(g.ctypes.data, G, first, nb_, L, err, ont, seed, buf.ctypes.data, len(buf), lens.ctypes.data)
        if tot < 0:
            raise RuntimeError("synth_reads_range failed: %d" % tot)
        ends = np.cumsum(lens.astype(np.int64) + 1)
        lo, src = 0, 0
        while lo < nb_ and len(vols) < num_vols:
            # reads lo .. hi - 1 still fit the open volume: curr + rsize + 1 <= MCS
            room = mcs - cur_bases + (ends[lo - 1] if lo else 0)
            hi = int(np.searchsorted(ends, room, side="right"))
            if hi > lo:
                nbytes = int(lens[lo:hi].astype(np.int64).sum())
                cur_codes.append(buf[src: src + nbytes].copy())
                cur_lens.append(lens[lo:hi].copy())
                cur_bases += nbytes + (hi - lo)
                src += nbytes
                lo = hi
            if lo < nb_:
                close()
        first += nb_
    if len(vols) < num_vols and cur_lens:
        close()
    return vols


def write_fasta(path, codes, lens):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    if synth_lib().synth_write_fasta(path.encode(), codes.ctypes.data, lens.ctypes.data, len(lens)) != 0:
        raise RuntimeError("cannot write " + path)


def jobs_from_candidates(cands, counts, rid_begin, ref_start_id=0):
    """alignment jobs of pairwise_mapping (mecat2pw/pw_impl.cpp:674-688): the +kmer_size/2 shift applies only when both
    start points are non-zero (:681-685).  -> structured array (qid_local, sid_local, chain, qstart, sstart)"""
    from .hip import JOB_DTYPE
    n = len(counts)
    maxc = cands.shape[1]
    mask = np.arange(maxc)[None, :] < counts[:, None]
    rid = np.broadcast_to(np.arange(n, dtype=np.int32)[:, None] + rid_begin, mask.shape)[mask]
    c = cands[mask]
    qstart = c["loc2"].copy()
    sstart = c["loc1"].copy()
    both = (qstart != 0) & (sstart != 0)
    qstart[both] += 6
    sstart[both] += 6
    jobs = np.zeros(len(c), dtype=JOB_DTYPE)
    jobs["qid_local"] = rid
    jobs["sid_local"] = c["readno"] - ref_start_id
    jobs["chain"] = c["chain"]
    jobs["qstart"] = qstart
    jobs["sstart"] = sstart
    return jobs


def ext_candidates_from_table(cands, counts, lens, reads_start_id=0):
    """the `.can` records (ExtensionCandidate, common/alignment.h:8-19) of a candidate table, as candidate_detect writes them
    (mecat2pw/pw_impl.cpp:767-792): [n, 13] int32 = qdir qid qext qsize qoff qend sdir sid sext ssize soff send score"""
    n, maxc = cands.shape
    mask = np.arange(maxc)[None, :] < counts[:, None]
    rid = np.broadcast_to(np.arange(n, dtype=np.int32)[:, None], mask.shape)[mask]
    c = cands[mask]
    qext, sext = c["loc2"].copy(), c["loc1"].copy()
    both = (qext != 0) & (sext != 0)
    qext[both] += 6
    sext[both] += 6
    qsize = lens[rid].astype(np.int32)
    rev = c["chain"] == 1
    qext[rev] = qsize[rev] - 1 - qext[rev]
    out = np.zeros((len(c), 13), dtype=np.int32)
    out[:, 0], out[:, 1], out[:, 2], out[:, 3] = c["chain"], rid + reads_start_id, qext, qsize
    out[:, 7], out[:, 8], out[:, 9], out[:, 12] = c["readno"], sext, lens[c["readno"]], c["score"]
    return out


def cns_templates(ec, num_reads, min_cov=4, min_size=5000):
    """what mecat2cns does with the records before its per-template loop: every candidate once with each of its reads as the
    template, template strand forward (normalise_candidate, mecat2cns/overlaps_partition.cpp:140-165), grouped by template; templates
    with fewer than min_cov candidates or shorter than 0.95 * min_size are skipped (reads_correction_can.cpp:33-34).
    -> (records [m, 13] grouped by template, tmpl_begin [T + 1], template ids [T])"""
    a = ec.copy()
    b = ec.copy()
    b[:, [0, 1, 2, 3]] = ec[:, [6, 7, 8, 9]]
    b[:, [6, 7, 8, 9]] = ec[:, [0, 1, 2, 3]]
    rec = np.concatenate([a, b])
    rev = rec[:, 6] == 1
    rec[rev, 0] ^= 1
    rec[rev, 6] ^= 1
    rec = rec[np.argsort(rec[:, 7], kind="stable")]
    first = np.searchsorted(rec[:, 7], np.arange(num_reads + 1))
    n_t = np.diff(first)
    ssize = np.zeros(num_reads, dtype=np.int64)
    ssize[rec[:, 7]] = rec[:, 9]
    keep = (n_t >= min_cov) & (ssize >= min_size * 0.95)
    ids = np.nonzero(keep)[0]
    sel = np.concatenate([np.arange(first[t], first[t + 1]) for t in ids]) if len(ids) else np.zeros(0, np.int64)
    tb = np.concatenate([[0], np.cumsum(n_t[ids])]).astype(np.int64)
    return np.ascontiguousarray(rec[sel]), tb, ids


def asm_blocks_layout(d, nreads, L, genome, nblocks, seed, err=0.02, n_every=0, iupac=False):
    """corrected reads (2 % error) laid out as canu hands them to mecat2asmpw / mecat2trimpw (Overlapmecat2asmpw.pm:483-503): <d>/ovlprep
    with one "-allreads -allbases -b <first> -e <last>" line per block and <d>/%06d.fasta, reads numbered from 1.
    n_every > 0: every n_every-th read carries Ns (one, three, a run of five, or one at either end and one inside).
    iupac: in every read the base behind each occurrence of four fixed 7-mers becomes an IUPAC code (R, Y, K, lower-case m): reads that
    overlap on the same strand carry the same letter at the same place (a letter equals itself), reads of the other strand and reads with
    an error there do not (a letter differs from A, C, G, T); every seventh read also gets one of the other seven codes at a random place.
    -> (blocks [(first, last)], total bases)"""
    codes, lens = synth_reads(nreads, L, err, genome, seed, 0)
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    per = (nreads + nblocks - 1) // nblocks
    blocks = [(k * per + 1, min(nreads, (k + 1) * per)) for k in range(nblocks)]
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(os.path.join(d, "ovlprep"), "w") as f:
        for b, e in blocks:
            f.write("-allreads -allbases -b %d -e %d\n" % (b, e))
    rng = np.random.default_rng(seed + 1000)
    for k, (b, e) in enumerate(blocks):
        with open(os.path.join(d, "%06d.fasta" % (k + 1)), "wb") as f:
            for rid in range(b, e + 1):
                text = lut[codes[starts[rid - 1]: starts[rid]]].copy()
                if n_every and rid % n_every == 0 and len(text) > 100:
                    # bases that are not A, C, G, T: single Ns, a run of them, one at either end of the read
                    kind = (rid // n_every) % 4
                    pos = rng.integers(20, len(text) - 20, size=3)
                    if kind == 0:
                        text[pos[0]] = ord("N")
                    elif kind == 1:
                        text[pos] = ord("N")
                    elif kind == 2:
                        text[pos[0]: pos[0] + 5] = ord("N")
                    else:
                        text[0] = ord("N"); text[-1] = ord("N"); text[pos[1]] = ord("N")
                if iupac and len(text) > 100:
                    raw = text.tobytes()
                    for pat, ch in ((b"ACGTACG", b"R"), (b"TTGACCA", b"Y"), (b"GGATCCA", b"K"), (b"CATGCAT", b"m")):
                        at = raw.find(pat)
                        while at >= 0 and at + 7 < len(text):
                            text[at + 7] = ch[0]
                            at = raw.find(pat, at + 1)
                    if rid % 7 == 0:
                        text[rng.integers(20, len(text) - 20)] = b"SWBDHVn"[(rid // 7) % 7]
                f.write(b">%d\n" % rid + text.tobytes() + b"\n")
    return blocks, int(lens.sum())


def asm_tool_run(exe, d, threads, start, nblocks, env=None, timeout=1800):
    """one run of an overlapper through its own command line -> (sorted output lines, seconds, stderr)"""
    import time
    t0 = time.time()
    r = subprocess.run([exe, "-P" + d, "-T%d" % threads, "-S%d" % start, "-E%d" % nblocks], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                       env=env, timeout=timeout)
    secs = time.time() - t0
    if r.returncode != 0:
        raise RuntimeError("%s failed: %s" % (exe, r.stderr[-500:]))
    lines = []
    for t in range(threads):
        p = os.path.join(d, "%d_%d.r" % (start, t))
        lines += open(p).read().splitlines()
        os.unlink(p)
    lines.sort()
    return lines, secs, r.stderr


def cpu_quota_cores():
    """CPU time this process tree may use, in cores: the cgroup's quota (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1) — a
    container can show every hardware thread of the host (os.cpu_count()) and still be held to a fraction of them.  None without a quota."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except Exception:
        return None


def annotate_cpu_baseline(line):
    """the bench line's cpu_baseline gets the cgroup CPU quota next to the thread count (`cores` = threads started)"""
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict):
        cb["cpu_quota_cores"] = cpu_quota_cores()
        cb["host_cpus"] = os.cpu_count()
        fs = cb.get("full_size_same_host")
        if isinstance(fs, dict):
            fs["cpu_quota_cores"] = cb["cpu_quota_cores"]
    return line
