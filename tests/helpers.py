"""ctypes bindings used by the tests only: the synthetic read generator, the CPU oracle (oracle/liboracle.so) and,
when it was built in this container, the harness around the unmodified reference (oracle/_ref/libref_harness.so).

Nothing here is imported by the product package (mecat_amd/).
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
NK = 1 << 26


def _build(target):
    subprocess.run(["make", "-s", target], cwd=ROOT, check=True, stdout=subprocess.DEVNULL)


# ----------------------------------------------------------------------------- synthetic reads
_synth = None


def synth_lib():
    global _synth
    if _synth is None:
        p = os.path.join(ROOT, "mecat_amd", "lib", "libsynth.so")
        if not os.path.exists(p):
            _build("synth")
        L = C.CDLL(p)
        L.synth_reads.restype = C.c_int64
        L.synth_reads.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_uint64,
                                  C.c_void_p, C.c_int64, C.c_void_p]
        L.synth_write_fasta.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int64]
        _synth = L
    return _synth


def synth_reads(nreads, L, err, genome, seed, ont=0):
    """-> (codes uint8[total] in 0..3, lens int32[nreads])"""
    lib = synth_lib()
    cap = nreads * (int(L * 1.25) + 64)
    bases = np.empty(cap, dtype=np.uint8)
    lens = np.empty(nreads, dtype=np.int32)
    tot = lib.synth_reads(genome, nreads, L, err, ont, seed, bases.ctypes.data, cap, lens.ctypes.data)
    assert tot >= 0, tot
    return bases[:tot].copy(), lens


def synth_reads_rep(nreads, L, err, genome, seed, ont=0, nfam=6, max_copies=40, nsat=20):
    """reads of a REPEAT-STRUCTURED genome (synth_genome_repeats in mecat_amd/tools/synth_reads.c: interspersed families of 300 - 5 000
    base elements at 0 - 5 % divergence on either strand, microsatellites and homopolymer runs), everything derived from `seed`
    -> (codes uint8[total], lens int32[nreads], stats dict)"""
    lib = synth_lib()
    vp = C.c_void_p
    lib.synth_genome.argtypes = [vp, C.c_int64, C.c_uint64]
    lib.synth_genome_repeats.argtypes = [vp, C.c_int64, C.c_uint64, C.c_int, C.c_int, C.c_int, vp]
    lib.synth_reads_range.restype = C.c_int64
    lib.synth_reads_range.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_uint64, vp, C.c_int64, vp]
    g = np.empty(genome, dtype=np.uint8)
    lib.synth_genome(g.ctypes.data, genome, seed)
    st = np.zeros(3, dtype=np.int64)
    lib.synth_genome_repeats(g.ctypes.data, genome, seed, nfam, max_copies, nsat, st.ctypes.data)
    cap = nreads * (int(L * 1.25) + 64)
    bases = np.empty(cap, dtype=np.uint8)
    lens = np.empty(nreads, dtype=np.int32)
    tot = lib.synth_reads_range(g.ctypes.data, genome, 0, nreads, L, err, ont, seed, bases.ctypes.data, cap, lens.ctypes.data)
    assert tot >= 0, tot
    return bases[:tot].copy(), lens, dict(family_bases=int(st[0]), satellite_bases=int(st[1]), copies=int(st[2]))


def write_fasta(path, codes, lens):
    assert synth_lib().synth_write_fasta(path.encode(), codes.ctypes.data, lens.ctypes.data, len(lens)) == 0


# ----------------------------------------------------------------------------- oracle
class OffsetT(C.Structure):
    _fields_ = [("offset", C.c_int), ("size", C.c_int)]


class OrcVolume(C.Structure):
    _fields_ = [("num_reads", C.c_int), ("num_bases", C.c_int), ("start_read_id", C.c_int),
                ("offs", C.POINTER(OffsetT)), ("pac", C.POINTER(C.c_uint8))]


class OrcIndex(C.Structure):
    _fields_ = [("counts", C.POINTER(C.c_int)), ("starts", C.POINTER(C.c_int64)), ("offsets", C.POINTER(C.c_int)),
                ("num_kmers", C.c_int64)]


class OrcCandidate(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("loc1", "loc2", "left1", "left2", "right1", "right2", "score", "num1", "num2",
                                       "readno", "readstart")] + [("chain", C.c_char)]


class OrcParams(C.Structure):
    _fields_ = [("maxc", C.c_int), ("min_align_size", C.c_int), ("min_kmer_match", C.c_int), ("min_kmer_dist", C.c_int),
                ("ddfs_cutoff", C.c_double), ("tech", C.c_int), ("output_gapped_start_point", C.c_int)]


class OrcBackList(C.Structure):
    _fields_ = [("score", C.c_int16), ("loczhi", C.c_int16 * 40), ("seedno", C.c_int16 * 40), ("seednum", C.c_int16),
                ("index", C.c_int)]


class OrcExtCandidate(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("qdir", "qid", "qext", "qsize", "qoff", "qend", "sdir", "sid", "sext", "ssize",
                                       "soff", "send", "score")]


class OrcM4(C.Structure):
    _fields_ = [("qid", C.c_int64), ("sid", C.c_int64), ("ident", C.c_double), ("vscore", C.c_int), ("qdir", C.c_int),
                ("qoff", C.c_int64), ("qend", C.c_int64), ("qsize", C.c_int64), ("sdir", C.c_int),
                ("soff", C.c_int64), ("send", C.c_int64), ("ssize", C.c_int64), ("qext", C.c_int64), ("sext", C.c_int64)]


class OrcAlnResult(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("ok", "query_start", "query_end", "target_start", "target_end", "matches", "columns")]


assert C.sizeof(OrcCandidate) == 48 and C.sizeof(OrcBackList) == 168 and C.sizeof(OrcM4) == 104 and C.sizeof(OrcExtCandidate) == 52

_orc = None


def orc():
    global _orc
    if _orc is None:
        p = os.path.join(ROOT, "oracle", "liboracle.so")
        _build("oracle")
        L = C.CDLL(p)
        vp = C.c_void_p
        L.orc_params_default.argtypes = [C.POINTER(OrcParams), C.c_int]
        L.orc_volume_pack.restype = C.POINTER(OrcVolume)
        L.orc_volume_pack.argtypes = [vp, vp, C.c_int, C.c_int]
        L.orc_volume_load.restype = C.POINTER(OrcVolume)
        L.orc_volume_load.argtypes = [C.c_char_p]
        L.orc_volume_dump.argtypes = [C.POINTER(OrcVolume), C.c_char_p]
        L.orc_volume_free.argtypes = [C.POINTER(OrcVolume)]
        L.orc_extract_one_seq.argtypes = [C.POINTER(OrcVolume), C.c_int, vp]
        L.orc_read_id_from_offset.argtypes = [C.POINTER(OrcVolume), C.c_int]
        L.orc_index_build.restype = C.POINTER(OrcIndex)
        L.orc_index_build.argtypes = [C.POINTER(OrcVolume)]
        L.orc_index_free.argtypes = [C.POINTER(OrcIndex)]
        L.orc_bk_new.restype = vp
        L.orc_bk_new.argtypes = [C.c_int]
        L.orc_bk_free.argtypes = [vp]
        L.orc_insert_loc.argtypes = [C.POINTER(OrcBackList), C.c_int, C.c_int, C.c_float, C.c_double]
        L.orc_find_location.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int), C.c_float, C.c_int, C.c_double]
        L.orc_seed_read.argtypes = [C.POINTER(OrcVolume), C.POINTER(OrcVolume), C.POINTER(OrcIndex), vp, C.c_int, C.c_int,
                                    C.POINTER(OrcParams), vp]
        L.orc_can_record.argtypes = [C.POINTER(OrcCandidate), C.c_int, C.c_int, C.c_int, C.POINTER(OrcExtCandidate)]
        L.orc_can_line.argtypes = [C.POINTER(OrcExtCandidate), C.c_char_p]
        L.orc_aligner_new.restype = vp
        L.orc_aligner_free.argtypes = [vp]
        L.orc_align.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        L.orc_dw_go.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(OrcAlnResult)]
        L.orc_dw_counters.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_xaligner_new.restype = vp
        L.orc_xaligner_free.argtypes = [vp]
        L.orc_xdrop_align.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]
        L.orc_xdrop_go.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(OrcAlnResult)]
        L.orc_map_read_x.argtypes = [C.POINTER(OrcVolume), C.POINTER(OrcVolume), C.POINTER(OrcIndex), vp, vp, vp, C.c_int,
                                     C.POINTER(OrcParams), vp]
        L.orc_m4_fill.argtypes = [C.POINTER(OrcAlnResult), C.c_int, C.c_int, C.c_char, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.POINTER(OrcM4)]
        L.orc_m4_postfilter.argtypes = [vp, C.c_int, vp]
        L.orc_m4_line.argtypes = [C.POINTER(OrcM4), C.c_int, C.c_char_p]
        L.orc_map_read.argtypes = [C.POINTER(OrcVolume), C.POINTER(OrcVolume), C.POINTER(OrcIndex), vp, vp, C.c_int,
                                   C.POINTER(OrcParams), vp]
        L.orc_xdrop_rowpar_selfcheck.argtypes = [vp, C.c_int, vp, C.c_int]
        L.orc_cns_new.restype = vp
        L.orc_cns_free.argtypes = [vp]
        L.orc_cns_one_direction.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_double, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_cns_dw.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, vp]
        L.orc_cns_get_alignment.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, vp]
        _orc = L
    return _orc


def orc_params(tech=0, maxc=100, **kw):
    p = OrcParams()
    orc().orc_params_default(C.byref(p), tech)
    p.maxc = maxc
    for k, v in kw.items():
        setattr(p, k, v)
    return p


CAND_DTYPE = np.dtype([(n, np.int32) for n in ("loc1", "loc2", "left1", "left2", "right1", "right2", "score", "num1",
                                               "num2", "readno", "readstart", "chain")])


def orc_pack(codes, lens, start_read_id=0):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    return orc().orc_volume_pack(codes.ctypes.data, lens.ctypes.data, len(lens), start_read_id)


def vol_arrays(v):
    """-> (offs int32[n,2], pac uint8[(num_bases+3)//4]) copies"""
    vv = v.contents
    offs = np.ctypeslib.as_array(C.cast(vv.offs, C.POINTER(C.c_int)), shape=(vv.num_reads, 2)).copy()
    pac = np.ctypeslib.as_array(vv.pac, shape=((vv.num_bases + 3) // 4,)).copy()
    return offs, pac


def orc_seed_all(ref, reads, idx, params, chain_as_char=0, rids=None):
    """candidates of every read (both strands) -> list of structured arrays (chain widened to int32)"""
    L = orc()
    bk = L.orc_bk_new(ref.contents.num_bases)
    out = (OrcCandidate * params.maxc)()
    res = []
    n = reads.contents.num_reads
    for rid in (range(n) if rids is None else rids):
        k = L.orc_seed_read(ref, reads, idx, bk, rid, chain_as_char, C.byref(params), out)
        a = np.zeros(k, dtype=CAND_DTYPE)
        for i in range(k):
            c = out[i]
            a[i] = (c.loc1, c.loc2, c.left1, c.left2, c.right1, c.right2, c.score, c.num1, c.num2, c.readno, c.readstart,
                    ord(c.chain))
        res.append(a)
    L.orc_bk_free(bk)
    return res


def can_lines_from_cands(cands_per_read, reads_offs, ref_offs, reads_start_id=0, ref_start_id=0):
    """A9 formatting of candidate arrays (host logic restated in numpy for tests): list of text lines"""
    lines = []
    for rid, a in enumerate(cands_per_read):
        qsize = int(reads_offs[rid, 1])
        for c in a:
            qext, sext = int(c["loc2"]), int(c["loc1"])
            if qext and sext:
                qext += 6
                sext += 6
            qdir = int(c["chain"])
            if qdir == 1:
                qext = qsize - 1 - qext
            ssize = int(ref_offs[int(c["readno"]) - ref_start_id, 1])
            lines.append("%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d" % (rid + reads_start_id, int(c["readno"]), qdir, 0, qext, sext,
                                                               int(c["score"]), qsize, ssize))
    return lines


def sha256_lines(lines):
    h = hashlib.sha256()
    for ln in sorted(lines):
        h.update(ln.encode() + b"\n")
    return h.hexdigest()


# ----------------------------------------------------------------------------- reference harness (this container only)
_ref = None


def ref_available():
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so"))


def ref_bin():
    p = os.path.join(ROOT, "oracle", "_ref", "mecat2pw")
    return p if os.path.exists(p) else None


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so"))
        vp = C.c_void_p
        L.refh_load_volume.restype = vp
        L.refh_load_volume.argtypes = [C.c_char_p]
        L.refh_free_volume.argtypes = [vp]
        for f in ("refh_vol_num_reads", "refh_vol_num_bases", "refh_vol_start_id"):
            getattr(L, f).argtypes = [vp]
        L.refh_vol_offsets.argtypes = [vp, vp]
        L.refh_vol_pac.argtypes = [vp, vp]
        L.refh_split.argtypes = [C.c_char_p, C.c_char_p]
        L.refh_build_index.restype = vp
        L.refh_build_index.argtypes = [vp, C.c_int]
        L.refh_free_index.argtypes = [vp]
        L.refh_index_dump.restype = C.c_long
        L.refh_index_dump.argtypes = [vp, vp, vp]
        L.refh_seed_read.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
        L.refh_insert_loc.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_float]
        L.refh_find_location.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int), C.c_float, C.c_int]
        L.refh_align.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        L.refh_dw_go.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, C.POINTER(C.c_double)]
        L.refh_xdrop_go.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, C.POINTER(C.c_double)]
        L.refh_xdrop_align.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]
        _ref = L
    return _ref


_ref_cns = None


def ref_cns_available():
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_cns.so"))


def ref_cns():
    """harness around the unmodified mecat2cns aligner (oracle/ref_harness_cns.cpp)"""
    global _ref_cns
    if _ref_cns is None:
        L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_cns.so"))
        vp = C.c_void_p
        L.refc_dw.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, vp]
        L.refc_get_alignment.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, vp]
        _ref_cns = L
    return _ref_cns


def cns_pair(rng, n, err, it):
    """two noisy copies of a random sequence and a seed point near their shared diagonal (code arrays, int8)"""
    g = rng.integers(0, 4, size=n + 1200).astype(np.int8)
    a0, b0 = int(rng.integers(0, 600)), int(rng.integers(0, 600))

    def mutate(x):
        out = []
        for c in x:
            r = rng.random()
            if r < err * 0.4:
                continue
            if r < err * 0.7:
                out.append(int(rng.integers(0, 4)))
            elif r < err:
                out.append(int((c + rng.integers(1, 4)) % 4))
                continue
            out.append(int(c))
        return np.array(out, dtype=np.int8)

    q, t = mutate(g[a0: a0 + n]), mutate(g[b0: b0 + n])
    mid = max(a0, b0) + n // 3
    qs = int((mid - a0) * 1.02) if it % 5 else int(rng.integers(0, len(q)))
    ts = int((mid - b0) * 1.02) if it % 5 else int(rng.integers(0, len(t)))
    qs, ts = min(max(qs, 0), len(q) - 1), min(max(ts, 0), len(t) - 1)
    if it % 11 == 0:
        qs = 0
    if it % 13 == 0:
        ts = len(t) - 1
    return q, t, qs, ts


def dense_reads(n=1400, L=10000, flank=6250, seed=77, err=0.135):
    """A deep, repeat-rich set for the `-n >= 1000` paths: n reads of L bases from a genome of 2 * flank + 1500
    bases that holds a 500-base unit three times in tandem (several candidates for one pair of reads), every tenth read an exact copy
    of the one before it.  Every pair of reads overlaps by 2 kb or more; the error rate keeps the k-mer buckets under the index's cap
    of 128 (depth / 2 x 0.865^13), and with `-k 2 -n 1500` the reads late in the file keep more than 1 000 overlaps each, so the
    reference's per-read m4 sort (pw_impl.cpp:581) sees lists of >= 1000 records.  -> (codes uint8[total], lens int32[n]), ACGT only."""
    rng = np.random.default_rng(seed)
    unit = rng.integers(0, 4, 500, dtype=np.uint8)
    unit2 = unit.copy()
    unit2[rng.integers(0, 500, 15)] = rng.integers(0, 4, 15, dtype=np.uint8)
    g = np.concatenate([rng.integers(0, 4, flank, dtype=np.uint8), unit, unit2, unit, rng.integers(0, 4, flank, dtype=np.uint8)])
    reads, prev = [], None
    for i in range(n):
        if i % 10 == 9 and prev is not None:
            reads.append(prev.copy())
            continue
        s = int(rng.integers(0, len(g) - L + 1))
        t = g[s:s + L]
        if rng.random() < 0.5:
            t = (3 - t[::-1]).astype(np.uint8)
        u = rng.random(L)
        kept = u >= 0.25 * err
        sub = (u >= 0.25 * err) & (u < 0.40 * err)
        bases = np.where(sub, rng.integers(0, 4, L, dtype=np.uint8), t).astype(np.uint8)
        ins = rng.random(L) < 0.60 * err
        slots = np.stack([bases, rng.integers(0, 4, L, dtype=np.uint8)], axis=1).ravel()
        r = slots[np.stack([kept, ins], axis=1).ravel()]
        reads.append(r)
        prev = r
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    return np.concatenate(reads).astype(np.uint8), lens


# ----------------------------------------------------------------------------- repeat-structured sets
# name -> (nreads, L, err, genome, seed, ont, nfam, max_copies, nsat); both techs, buckets pushed to and beyond the cap of 128
REP_SETS = {
    "rep_pb": (700, 5000, 0.08, 250000, 601, 0, 10, 80, 25),
    "rep_ont": (500, 6000, 0.08, 200000, 602, 1, 4, 90, 30),
}
# the CLI pin: config-1 size (1 000 reads x 10 kb @ 15 %), repeat-rich
REP_CLI = (1000, 10000, 0.15, 500000, 603, 0, 8, 120, 40)


def rep_set(name):
    n, L, err, G, seed, ont, nfam, mc, nsat = REP_CLI if name == "rep_cli" else REP_SETS[name]
    codes, lens, st = synth_reads_rep(n, L, err, G, seed, ont, nfam, mc, nsat)
    return codes, lens, ont, st


def orc_stats_reset():
    orc().orc_stats_reset()


def orc_stats():
    """statistics of the oracle's 41st-seed rule since the last reset (test infrastructure, see oracle/mecat_oracle.c)"""
    a = np.zeros(8, dtype=np.int64)
    orc().orc_stats_get.argtypes = [C.c_void_p]
    orc().orc_stats_get(a.ctypes.data)
    return dict(insert_loc=int(a[0]), non_self=int(a[1]), tail_replaced=int(a[2]), dropped=int(a[3]), ignored=int(a[4]), non_self_dropped=int(a[5]))


def bucket_stats(codes, lens, cap=128):
    """k-mer bucket sizes of a read set, from the reads themselves (numpy): buckets at the cap, buckets dropped beyond it"""
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    c = codes.astype(np.int64)
    n = len(c)
    if n < 13:
        return dict(at_cap=0, dropped=0, largest=0)
    k = np.zeros(n - 12, dtype=np.int64)
    for j in range(13):
        k = (k << 2) | c[j: n - 12 + j]
    ok = np.ones(n - 12, dtype=bool)
    for s in starts[1:-1]:
        ok[max(0, s - 12): s] = False      # k-mers that would span two reads
    u, cnt = np.unique(k[ok], return_counts=True)
    return dict(at_cap=int((cnt == cap).sum()), near_cap=int(((cnt >= cap - 28) & (cnt <= cap)).sum()), dropped=int((cnt > cap).sum()), largest=int(cnt.max()))
