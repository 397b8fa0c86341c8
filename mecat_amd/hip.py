"""ctypes mirror of include/mecat_hip.h (see that header for the reference functions each call replaces)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # MECAT_HIP_LIB: a development build of the same library (e.g. the -DMECAT_DW_STATS variant, tools/dev/dw_breakdown.sh)
    return os.environ.get("MECAT_HIP_LIB") or os.path.join(_HERE, "lib", "libmecat_hip.so")


class MhipError(RuntimeError):
    pass


class Offset(C.Structure):
    _fields_ = [("offset", C.c_int32), ("size", C.c_int32)]


class Candidate(C.Structure):
    """candidate_save (mecat2pw/pw_impl.h:21-25)"""
    _fields_ = [(n, C.c_int32) for n in ("loc1", "loc2", "left1", "left2", "right1", "right2", "score", "num1", "num2",
                                         "readno", "readstart", "chain")]


class Params(C.Structure):
    _fields_ = [("maxc", C.c_int32), ("min_align_size", C.c_int32), ("min_kmer_match", C.c_int32),
                ("min_kmer_dist", C.c_int32), ("tech", C.c_int32), ("ddfs_cutoff", C.c_double)]


class AlnResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("ok", "query_start", "query_end", "target_start", "target_end", "matches",
                                         "columns", "blocks")]


class AlnJob(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("qid_local", "sid_local", "chain", "qstart", "sstart")]


class CnsResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("ok", "qoff", "qend", "soff", "send", "left_cols", "right_cols", "first_col", "last_col",
                                         "mat", "ins", "dele", "query_start", "query_end", "target_start", "target_end")]


CNS_DTYPE = np.dtype([(n, np.int32) for n, _ in CnsResult._fields_])
CAND_DTYPE = np.dtype([(n, np.int32) for n, _ in Candidate._fields_])
ALN_DTYPE = np.dtype([(n, np.int32) for n, _ in AlnResult._fields_])
JOB_DTYPE = np.dtype([(n, np.int32) for n, _ in AlnJob._fields_])
assert CAND_DTYPE.itemsize == C.sizeof(Candidate) == 48

_lib = None


def lib():
    """Loads libmecat_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise MhipError("%s is missing: run `make hip` (python -c 'import __graft_entry__ as g; g.build()'). "
                        "There is no CPU fallback." % p)
    L = C.CDLL(p)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.mhip_last_error.restype = C.c_char_p
    L.mhip_ctx_create.argtypes = [i32, vp, C.POINTER(vp)]
    L.mhip_ctx_destroy.argtypes = [vp]
    L.mhip_ctx_sync.argtypes = [vp]
    L.mhip_params_default.argtypes = [C.POINTER(Params), i32]
    L.mhip_ctx_set_profiling.argtypes = [vp, i32]
    L.mhip_ctx_kernel_stats.argtypes = [vp, C.c_char_p, C.POINTER(i64), C.POINTER(C.c_double)]
    L.mhip_ctx_kernel_names.argtypes = [vp, C.c_char_p, i32]
    L.mhip_ctx_reset_stats.argtypes = [vp]
    L.mhip_ctx_counters.argtypes = [vp, C.POINTER(i64)]
    L.mhip_debug_counter.argtypes = [vp, i32, C.POINTER(i64)]
    L.mhip_volume_upload.argtypes = [vp, vp, vp, i32, i32, i32, C.POINTER(vp)]
    L.mhip_volume_pack.argtypes = [vp, vp, i64, vp, vp, vp, i32, i32, i32, C.POINTER(vp), vp]
    L.mhip_volume_free.argtypes = [vp]
    L.mhip_volume_set_nplane.argtypes = [vp, vp, vp]
    L.mhip_volume_num_reads.argtypes = [vp]
    L.mhip_volume_num_bases.argtypes = [vp]
    L.mhip_index_build.argtypes = [vp, vp, C.POINTER(vp)]
    L.mhip_index_build_ex.argtypes = [vp, vp, i32, C.POINTER(vp)]
    L.mhip_index_free.argtypes = [vp]
    L.mhip_index_num_kmers.restype = i64
    L.mhip_index_num_kmers.argtypes = [vp]
    L.mhip_index_download.argtypes = [vp, vp, vp, vp]
    L.mhip_seed_reads.argtypes = [vp, vp, vp, vp, i32, i32, C.POINTER(Params), vp, vp]
    L.mhip_seed_reads_dev.argtypes = [vp, vp, vp, vp, i32, i32, C.POINTER(Params), vp, vp]
    L.mhip_seed_reads_strided_dev.argtypes = [vp, vp, vp, vp, i32, i32, i32, C.POINTER(Params), vp, vp]
    L.mhip_jobs_from_candidates_dev.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, C.POINTER(i32)]
    L.mhip_align_candidates.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.mhip_align_candidates_dev.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.mhip_xalign_candidates.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.mhip_xalign_candidates_dev.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.mhip_ctx_reserve_index.argtypes = [vp, C.c_int64]
    L.mhip_ctx_buffer.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(vp)]
    L.mhip_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.mhip_pack_candidates_dev.argtypes = [vp, vp, vp, i32, i32, vp, C.POINTER(i64)]
    L.mhip_cns_align_candidates.argtypes = [vp, vp, vp, vp, i32, C.c_double, i32, i32, vp, vp]
    L.mhip_cns_align_candidates_dev.argtypes = [vp, vp, vp, vp, i32, C.c_double, i32, i32, vp, vp]
    # multi-GPU
    L.mhip_comm_unique_id.argtypes = [vp]
    L.mhip_comm_init.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    L.mhip_comm_init_hostfile.argtypes = [vp, i32, i32, C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.mhip_comm_destroy.argtypes = [vp]
    L.mhip_comm_barrier.argtypes = [vp]
    L.mhip_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.mhip_comm_selftest.argtypes = [vp]
    L.mhip_comm_bytes_received.restype = i64
    L.mhip_comm_bytes_received.argtypes = [vp]
    L.mhip_shard_local_count.argtypes = [i32, i32, i32, i32, i32, i32]
    L.mhip_shard_first_read.argtypes = [i32, i32, i32, i32, i32, i32]
    L.mhip_shard_deal_rows.argtypes = [i32, vp, i32, i32, vp]
    L.mhip_seed_reads_chunked_dev.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, C.POINTER(Params), vp, vp]
    L.mhip_allgather_candidates.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
    L.mhip_seed_reads_sharded.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, C.POINTER(Params), vp, vp]
    L.mhip_align_sharded.argtypes = [vp, vp, vp, i32, i32, vp, C.POINTER(i64)]
    L.mhip_sharded_tables.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
    L.mhip_cns_accept_templates.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, C.c_double, i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp),
                                            C.POINTER(i64), C.POINTER(i64)]
    L.mhip_cns_free.argtypes = [vp]
    L.mhip_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.mhip_host_free.argtypes = [vp]
    assert L.mhip_abi_version() == 1
    _lib = L
    return L


def deal_rows(num_vols, todo, nranks):
    """rows mode's static deal (mhip_shard_deal_rows) -> (owner[num_vols], heaviest rank's cells)"""
    todo = np.ascontiguousarray(todo, dtype=np.int32)
    owner = np.zeros(max(1, num_vols), dtype=np.int32)
    mx = lib().mhip_shard_deal_rows(num_vols, todo.ctypes.data, len(todo), nranks, owner.ctypes.data)
    if mx < 0:
        raise MhipError("mhip_shard_deal_rows: bad arguments")
    return owner[:num_vols], mx


def _chk(rc):
    if rc != 0:
        raise MhipError(lib().mhip_last_error().decode())


def default_params(tech=0, **kw):
    p = Params()
    lib().mhip_params_default(C.byref(p), tech)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Context:
    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        _chk(lib().mhip_ctx_create(device, stream, C.byref(self.h)))
        self.device = device

    def close(self):
        if self.h:
            lib().mhip_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def sync(self):
        _chk(lib().mhip_ctx_sync(self.h))

    def set_profiling(self, on=True):
        _chk(lib().mhip_ctx_set_profiling(self.h, int(on)))

    def reset_stats(self):
        _chk(lib().mhip_ctx_reset_stats(self.h))

    def kernel_stats(self):
        buf = C.create_string_buffer(8192)
        _chk(lib().mhip_ctx_kernel_names(self.h, buf, len(buf)))
        out = {}
        for name in buf.value.decode().split("\n"):
            if not name:
                continue
            n, ms = C.c_int64(), C.c_double()
            _chk(lib().mhip_ctx_kernel_stats(self.h, name.encode(), C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out

    def counters(self):
        a = (C.c_int64 * 8)()
        _chk(lib().mhip_ctx_counters(self.h, a))
        names = ("lookups", "hits", "candidates", "dw_blocks", "dw_cells", "snake_bases", "aligned_bases", "aln_ok")
        return dict(zip(names, [int(x) for x in a]))

    def debug_counter(self, slot):
        """development counters 8..15 (13 / 14: strands taken by seed_strand / left to the kernel chain)"""
        v = C.c_int64()
        _chk(lib().mhip_debug_counter(self.h, slot, C.byref(v)))
        return int(v.value)


class Volume:
    """device-resident volume_t; `pac`, `offs` as load_volume() leaves them (uint8[(num_bases+3)//4], int32[n,2])"""

    def __init__(self, ctx, pac, offs, num_bases, start_read_id=0):
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.int32).reshape(-1, 2)
        assert len(pac) >= (num_bases + 3) // 4
        self.h = C.c_void_p()
        self.offs = offs
        self.num_reads = len(offs)
        self.num_bases = int(num_bases)
        self.start_read_id = start_read_id
        _chk(lib().mhip_volume_upload(ctx.h, pac.ctypes.data, offs.ctypes.data, len(offs), num_bases, start_read_id,
                                      C.byref(self.h)))

    @classmethod
    def from_letters(cls, ctx, text, seq_start, line_width, offs, num_bases, start_read_id=0):
        """mhip_volume_pack: the volume packed on the device from the file's bytes -> (Volume, packed bytes uint8[(num_bases+3)//4])"""
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8))
        seq_start = np.ascontiguousarray(seq_start, dtype=np.int64)
        line_width = np.ascontiguousarray(line_width, dtype=np.int32)
        offs = np.ascontiguousarray(offs, dtype=np.int32).reshape(-1, 2)
        pac = np.zeros(((num_bases + 3) // 4,), dtype=np.uint8)
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        self.offs = offs
        self.num_reads = len(offs)
        self.num_bases = int(num_bases)
        self.start_read_id = start_read_id
        _chk(lib().mhip_volume_pack(ctx.h, text.ctypes.data, len(text), seq_start.ctypes.data, line_width.ctypes.data, offs.ctypes.data, len(offs),
                                    num_bases, start_read_id, C.byref(self.h), pac.ctypes.data))
        return self, pac

    @classmethod
    def from_file(cls, ctx, path):
        """reads a wrk/vol<k> file (layout of dump_volume, common/split_database.cpp:135-153)"""
        with open(path, "rb") as f:
            hdr = np.fromfile(f, dtype=np.int32, count=3)
            offs = np.fromfile(f, dtype=np.int32, count=2 * int(hdr[0])).reshape(-1, 2)
            pac = np.fromfile(f, dtype=np.uint8, count=(int(hdr[1]) + 3) // 4)
        return cls(ctx, pac, offs, int(hdr[1]), int(hdr[2]))

    def free(self):
        if self.h:
            lib().mhip_volume_free(self.h)
            self.h = C.c_void_p()


class Index:
    def __init__(self, ctx, vol, max_bucket=None):
        self.h = C.c_void_p()
        self.ctx = ctx
        if max_bucket is None:
            _chk(lib().mhip_index_build(ctx.h, vol.h, C.byref(self.h)))
        else:
            _chk(lib().mhip_index_build_ex(ctx.h, vol.h, int(max_bucket), C.byref(self.h)))

    @property
    def num_kmers(self):
        return lib().mhip_index_num_kmers(self.h)

    def download(self, want_counts=True, want_offsets=True):
        counts = np.empty(1 << 26, dtype=np.int32) if want_counts else None
        offs = np.empty(self.num_kmers, dtype=np.int32) if want_offsets else None
        _chk(lib().mhip_index_download(self.ctx.h, self.h, counts.ctypes.data if want_counts else None,
                                       offs.ctypes.data if want_offsets else None))
        return counts, offs

    def download_aux(self):
        """(slots uint16[num_kmers], recs uint32[4^13, 4] or None, cut_step): the side arrays of the seeding stage (test hook)"""
        slots = np.empty(self.num_kmers, dtype=np.uint16)
        recs = np.empty((1 << 26, 4), dtype=np.uint32)
        cs = C.c_int()
        lib().mhip_index_download_aux.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        rc = lib().mhip_index_download_aux(self.ctx.h, self.h, slots.ctypes.data, recs.ctypes.data, C.byref(cs))
        if rc < 0:
            raise MhipError(lib().mhip_last_error().decode())
        return slots, (recs if rc == 0 else None), cs.value

    def free(self):
        if self.h:
            lib().mhip_index_free(self.h)
            self.h = C.c_void_p()


ASM_CAND_DTYPE = np.dtype([(n, np.int32) for n in ("loc1", "loc2", "left1", "left2", "right1", "right2", "score", "num1", "num2", "readno",
                                                    "readstart", "chain")])


def asm_seed_reads(ctx, idx, block, reads, rid_begin, rid_end):
    """mhip_asm_seed_reads: the candidate stage of mecat2asmpw's pairwise_mapping -> (cands [n, 100] ASM_CAND_DTYPE, counts int32[n])"""
    n = rid_end - rid_begin
    out = np.zeros((n, 100), dtype=ASM_CAND_DTYPE)
    cnt = np.zeros(n, dtype=np.int32)
    lib().mhip_asm_seed_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    _chk(lib().mhip_asm_seed_reads(ctx.h, idx.h, block.h, reads.h, rid_begin, rid_end, out.ctypes.data, cnt.ctypes.data))
    return out, cnt


def seed_reads(ctx, idx, ref, reads, rid_begin, rid_end, params):
    """-> (cands structured array [n, maxc], counts int32[n]) on the host"""
    n = rid_end - rid_begin
    out = np.zeros((n, params.maxc), dtype=CAND_DTYPE)
    cnt = np.zeros(n, dtype=np.int32)
    _chk(lib().mhip_seed_reads(ctx.h, idx.h, ref.h, reads.h, rid_begin, rid_end, C.byref(params), out.ctypes.data,
                               cnt.ctypes.data))
    return out, cnt


def seed_reads_dev(ctx, idx, ref, reads, rid_begin, rid_end, params, d_out, d_counts):
    _chk(lib().mhip_seed_reads_dev(ctx.h, idx.h, ref.h, reads.h, rid_begin, rid_end, C.byref(params), d_out, d_counts))


def seed_reads_strided_dev(ctx, idx, ref, reads, rid_begin, rid_stride, n, params, d_out, d_counts):
    _chk(lib().mhip_seed_reads_strided_dev(ctx.h, idx.h, ref.h, reads.h, rid_begin, rid_stride, n, C.byref(params), d_out, d_counts))


def jobs_from_candidates_dev(ctx, d_cands, d_counts, n_reads, maxc, rid_begin, rid_stride, ref_start_id, part_index, part_count,
                             d_jobs):
    n = C.c_int32(0)
    _chk(lib().mhip_jobs_from_candidates_dev(ctx.h, d_cands, d_counts, n_reads, maxc, rid_begin, rid_stride, ref_start_id,
                                             part_index, part_count, d_jobs, C.byref(n)))
    return n.value


def align_candidates(ctx, ref, reads, jobs, min_align_size, tech=0):
    """tech 0: dw / DiffAligner, tech 1: X-drop aligner (nanopore mode)"""
    jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
    out = np.zeros(len(jobs), dtype=ALN_DTYPE)
    if len(jobs):
        fn = lib().mhip_xalign_candidates if tech == 1 else lib().mhip_align_candidates
        _chk(fn(ctx.h, ref.h, reads.h, jobs.ctypes.data, len(jobs), min_align_size, out.ctypes.data))
    return out


def align_candidates_dev(ctx, ref, reads, d_jobs, n, min_align_size, d_out):
    _chk(lib().mhip_align_candidates_dev(ctx.h, ref.h, reads.h, d_jobs, n, min_align_size, d_out))


def cns_align_candidates(ctx, ref, reads, jobs, error_rate, min_align_size, dir_cols_cap):
    """mecat2cns re-aligner (GetAlignment).  -> (results [n] CNS_DTYPE, ops [n, 2, dir_cols_cap / 16] uint32)"""
    jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
    n = len(jobs)
    res = np.zeros(n, dtype=CNS_DTYPE)
    ops = np.zeros((n, 2, dir_cols_cap // 16), dtype=np.uint32)
    if n:
        _chk(lib().mhip_cns_align_candidates(ctx.h, ref.h, reads.h, jobs.ctypes.data, n, float(error_rate), min_align_size, dir_cols_cap,
                                             res.ctypes.data, ops.ctypes.data))
    return res, ops


def cns_expand(res, ops_row, qcodes, tcodes):
    """Rebuild the aligned strings of one result from its ops (the host side of the 2-bit column format): qcodes = the
    query read as the aligner saw it (reverse-complemented when chain != 0), tcodes = the template.  -> (qaln, saln) over
    "ACGT-", i.e. m5qaln / m5saln."""
    def unpack(words, ncols):
        w = np.asarray(words[: (ncols + 15) // 16], dtype=np.uint32)
        cols = ((w[:, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3).reshape(-1)
        return cols[:ncols].astype(np.uint8)
    left = unpack(ops_row[0], int(res["left_cols"]))[::-1]
    right = unpack(ops_row[1], int(res["right_cols"]))
    ops = np.concatenate([left, right])
    dec = np.frombuffer(b"ACGT", dtype=np.uint8)
    qi = int(res["query_start"]) + np.cumsum(ops != 1) - (ops != 1)
    ti = int(res["target_start"]) + np.cumsum(ops != 2) - (ops != 2)
    qs = np.where(ops == 1, ord("-"), dec[np.asarray(qcodes)[np.minimum(qi, len(qcodes) - 1)]]).astype(np.uint8)
    ts = np.where(ops == 2, ord("-"), dec[np.asarray(tcodes)[np.minimum(ti, len(tcodes) - 1)]]).astype(np.uint8)
    a, b = int(res["first_col"]), int(res["last_col"])
    return qs[a:b].tobytes(), ts[a:b].tobytes()


EXT_CAND_DTYPE = np.dtype([(n, np.int32) for n in ("qdir", "qid", "qext", "qsize", "qoff", "qend", "sdir", "sid", "sext", "ssize", "soff", "send",
                                                     "score")])
ACCEPTED_DTYPE = np.dtype([("template_index", np.int32), ("qid", np.int32), ("sid", np.int32), ("qoff", np.int32), ("qend", np.int32),
                           ("soff", np.int32), ("send", np.int32), ("aln_size", np.int32), ("cand_index", np.int64), ("str_offset", np.int64)])
assert EXT_CAND_DTYPE.itemsize == 52 and ACCEPTED_DTYPE.itemsize == 48


def cns_accept_templates(ctx, vol, host_pac, cands, tmpl_begin, tech, min_align_size, min_mapping_ratio, threads=8):
    """mecat2cns' accept loop for a batch of templates.  cands: [n] EXT_CAND_DTYPE (or [n, 13] int32) grouped by template, sorted in
    place.  -> (accepted [k] ACCEPTED_DTYPE, strings as a uint8 array over the library's buffer, number of alignments computed)"""
    cands = np.ascontiguousarray(cands)
    tb = np.ascontiguousarray(tmpl_begin, dtype=np.int64)
    # (host_pac is no longer read by the library: the strings are built on the device; kept in the signature)
    pac = np.ascontiguousarray(host_pac, dtype=np.uint8) if host_pac is not None else None
    acc, st = C.c_void_p(), C.c_void_p()
    na, sb, nj = C.c_int64(), C.c_int64(), C.c_int64()
    _chk(lib().mhip_cns_accept_templates(ctx.h, vol.h, pac.ctypes.data if pac is not None else None, cands.ctypes.data, tb.ctypes.data, len(tb) - 1, tech, min_align_size,
                                         float(min_mapping_ratio), threads, C.byref(acc), C.byref(na), C.byref(st), C.byref(sb), C.byref(nj)))
    a = np.ctypeslib.as_array(C.cast(acc, C.POINTER(C.c_uint8)), shape=(na.value * 48,)).view(ACCEPTED_DTYPE).copy() if na.value else np.zeros(0, ACCEPTED_DTYPE)
    lib().mhip_cns_free(acc)
    if not sb.value:
        lib().mhip_cns_free(st)
        return a, np.zeros(0, np.uint8), nj.value
    # the strings stay where the library put them (a gigabyte at config 2: no copy): a uint8 array over the C buffer, freed with the array
    import weakref
    s = np.ctypeslib.as_array(C.cast(st, C.POINTER(C.c_uint8)), shape=(sb.value,))
    weakref.finalize(s, lib().mhip_cns_free, C.c_void_p(st.value))
    return a, s, nj.value


COMM_ID_BYTES = 128
SHARD_CHUNK = 500


def comm_unique_id():
    """rank 0: the RCCL unique id (bytes) to hand to every rank"""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _chk(lib().mhip_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """mhip_comm: RCCL communicator over one context per rank (hostfile_dir: the test-hook transport)"""

    def __init__(self, ctx, nranks, rank, unique_id=None, hostfile_dir=None, run_id="0", solo=False):
        self.h = C.c_void_p()
        self.ctx, self.nranks, self.rank = ctx, nranks, rank
        if solo:        # bench hook: this rank's share of every sharded call, no transport (mecat_hip.h: mhip_comm_init_solo)
            lib().mhip_comm_init_solo.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
            _chk(lib().mhip_comm_init_solo(ctx.h, nranks, rank, C.byref(self.h)))
        elif hostfile_dir is not None:
            _chk(lib().mhip_comm_init_hostfile(ctx.h, nranks, rank, hostfile_dir.encode(), run_id.encode(), C.byref(self.h)))
        else:
            buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id if unique_id is not None else bytes(COMM_ID_BYTES))
            _chk(lib().mhip_comm_init(ctx.h, nranks, rank, buf, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().mhip_comm_destroy(self.h)
            self.h = C.c_void_p()

    def barrier(self):
        _chk(lib().mhip_comm_barrier(self.h))

    def info(self):
        """-> (transport: 0 RCCL / 1 host files, ncclCommCount of the RCCL communicator)"""
        t, n = C.c_int(), C.c_int()
        _chk(lib().mhip_comm_info(self.h, C.byref(t), C.byref(n)))
        return t.value, n.value

    def bytes_sent(self):
        lib().mhip_comm_bytes_sent.restype = C.c_int64
        lib().mhip_comm_bytes_sent.argtypes = [C.c_void_p]
        return int(lib().mhip_comm_bytes_sent(self.h))

    def local_jobs(self):
        lib().mhip_comm_local_jobs.restype = C.c_int64
        lib().mhip_comm_local_jobs.argtypes = [C.c_void_p]
        return int(lib().mhip_comm_local_jobs(self.h))

    def bytes_received(self):
        return int(lib().mhip_comm_bytes_received(self.h))

    def index_build_sharded(self, vol):
        """mhip_index_build_sharded: the volume's table built by all ranks together; every rank gets the complete table"""
        idx = Index.__new__(Index)
        idx.h = C.c_void_p()
        idx.ctx = self.ctx
        lib().mhip_index_build_sharded.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        _chk(lib().mhip_index_build_sharded(self.h, vol.h, C.byref(idx.h)))
        return idx

    def index_build_auto(self, vol):
        """mhip_index_build_auto: the faster of Index(ctx, vol) on every rank and index_build_sharded, measured by the first call on this
        communicator.  -> (index, {"replicated_ms", "sharded_ms", "chosen"})"""
        idx = Index.__new__(Index)
        idx.h = C.c_void_p()
        idx.ctx = self.ctx
        ms = (C.c_double * 2)()
        sh = C.c_int()
        lib().mhip_index_build_auto.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        _chk(lib().mhip_index_build_auto(self.h, vol.h, C.byref(idx.h), ms, C.byref(sh)))
        return idx, {"replicated_ms": ms[0], "sharded_ms": ms[1], "chosen": "sharded" if sh.value else "replicated",
                     "measured": ms[0] > 0 or ms[1] > 0}

    def seed_reads_sharded(self, idx, ref, reads, rid_begin, rid_end, params, chunk=SHARD_CHUNK, cell_shift=0, host=True):
        """-> (cands [n, maxc] structured, counts [n]) on the host when host=True, else None (tables stay on the device)"""
        n = rid_end - rid_begin
        if host:
            out = np.zeros((n, params.maxc), dtype=CAND_DTYPE)
            cnt = np.zeros(n, dtype=np.int32)
            _chk(lib().mhip_seed_reads_sharded(self.h, idx.h, ref.h, reads.h, rid_begin, rid_end, chunk, cell_shift, C.byref(params),
                                               out.ctypes.data, cnt.ctypes.data))
            return out, cnt
        _chk(lib().mhip_seed_reads_sharded(self.h, idx.h, ref.h, reads.h, rid_begin, rid_end, chunk, cell_shift, C.byref(params), None, None))
        return None

    def align_sharded(self, ref, reads, min_align_size, tech=0, host=True):
        """-> results of every candidate of the slab, read-major (host=True), and their number"""
        nj = C.c_int64()
        if not host:
            _chk(lib().mhip_align_sharded(self.h, ref.h, reads.h, tech, min_align_size, None, C.byref(nj)))
            return None, nj.value
        _, _, _, total = self.tables()
        out = np.zeros(max(total, 1), dtype=ALN_DTYPE)
        _chk(lib().mhip_align_sharded(self.h, ref.h, reads.h, tech, min_align_size, out.ctypes.data, C.byref(nj)))
        return out[: nj.value], nj.value

    def tables(self):
        """device pointers (cands, counts, results) of the last sharded calls and the number of jobs"""
        a, b, c2, nj = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
        _chk(lib().mhip_sharded_tables(self.h, C.byref(a), C.byref(b), C.byref(c2), C.byref(nj)))
        return a.value, b.value, c2.value, nj.value
