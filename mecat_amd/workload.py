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
