#!/usr/bin/env python3
"""Golden vectors for mhip_cns_accept_templates: the UNMODIFIED consensus_one_read_can_pacbio /
_nanopore of the reference (oracle/_ref/libref_cns_accept.so = mecat2cns compiled from /root/reference/src) run template by
template on candidates produced by the reference's own seeding (oracle/_ref/libref_harness.so).  Build container only:
    python tests/golden/make_golden_cns_accept.py
Writes tests/golden/cns_accept.npz: the normalised candidate records fed to the reference (inputs) and, per template, what it
accepted: (soff, send, aln_size) of every CnsAlns entry in order and a SHA-256 over its gap-normalised strings."""
import ctypes as C
import hashlib
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import helpers as H  # noqa: E402

SETS = {
    # name: (nreads, L, err, genome, seed, ont, tech, min_align_size, min_mapping_ratio)
    "pacbio": (240, 5000, 0.15, 12000, 77, 0, 0, 1000, 0.6),
    "nanopore": (160, 4000, 0.12, 8000, 78, 1, 1, 500, 0.4),
}


def candidates_of(R, codes, lens, tech, d):
    """the reference's .can records (ExtensionCandidate: qdir qid qext qsize qoff qend sdir sid sext ssize soff send score)"""
    fa = os.path.join(d, "r.fa")
    H.write_fasta(fa, codes, lens)
    wrk = os.path.join(d, "w")
    os.makedirs(wrk, exist_ok=True)
    R.refh_split(fa.encode(), wrk.encode())
    rv = R.refh_load_volume(os.path.join(wrk, "vol0").encode())
    ridx = R.refh_build_index(rv, 1)
    R.refh_set_params(100, 500 if tech else 2000, 2 if tech else 4, tech)
    buf = np.zeros((100, 12), dtype=np.int32)
    recs = []
    for rid in range(len(lens)):
        k = R.refh_seed_read(rv, rv, ridx, rid, 0, buf.ctypes.data)
        for c in buf[:k]:
            loc1, loc2, score, readno, chain = int(c[0]), int(c[1]), int(c[6]), int(c[9]), int(c[11])
            qext, sext = loc2, loc1
            if qext and sext:
                qext += 6
                sext += 6
            qsize, ssize = int(lens[rid]), int(lens[readno])
            if chain == 1:
                qext = qsize - 1 - qext
            recs.append([chain, rid, qext, qsize, 0, 0, 0, readno, sext, ssize, 0, 0, score])      # candidate_detect, pw_impl.cpp:767-792
    R.refh_free_index(ridx)
    R.refh_free_volume(rv)
    return fa, np.array(recs, dtype=np.int32)


def normalise(ec):
    """overlaps_partition.cpp:140-165: every candidate once with each read as the template (sid), template strand forward"""
    a = ec.copy()                                                   # subject_is_target
    b = ec.copy()
    b[:, [0, 1, 2, 3]] = ec[:, [6, 7, 8, 9]]
    b[:, [6, 7, 8, 9]] = ec[:, [0, 1, 2, 3]]
    out = np.concatenate([np.stack([b, a], axis=1).reshape(-1, 13)])
    rev = out[:, 6] == 1
    out[rev, 0] ^= 1
    out[rev, 6] ^= 1
    return out


def main():
    R = H.ref()
    A = C.CDLL(os.path.join(H.ROOT, "oracle", "_ref", "libref_cns_accept.so"))
    A.refa_load_reads.argtypes = [C.c_char_p]
    A.refa_consensus_can.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
    out = {}
    for name, (n, L, err, G, seed, ont, tech, mas, ratio) in SETS.items():
        d = tempfile.mkdtemp(prefix="cnsacc_")
        codes, lens = H.synth_reads(n, L, err, G, seed, ont)
        fa, ec = candidates_of(R, codes, lens, tech, d)
        norm = normalise(ec)
        order = np.argsort(norm[:, 7], kind="stable")              # group by template, file order inside (CmpExtensionCandidateBySid is unstable
        norm = norm[order]                                           # in the reference; the per-template sort by score makes the order irrelevant)
        tb = np.searchsorted(norm[:, 7], np.arange(n + 1)).astype(np.int64)
        assert A.refa_load_reads(fa.encode()) == n
        meta_all, sha_all, nacc = [], [], []
        sbuf = np.zeros(400_000_000, dtype=np.int8)
        for t in range(n):
            b, e = int(tb[t]), int(tb[t + 1])
            if e == b:
                nacc.append(0)
                sha_all.append("")
                continue
            cand = np.ascontiguousarray(norm[b:e]).copy()
            meta = np.zeros((128, 4), dtype=np.int32)
            used = C.c_long()
            k = A.refa_consensus_can(tech, cand.ctypes.data, e - b, t, mas, ratio, meta.ctypes.data, sbuf.ctypes.data, len(sbuf), C.byref(used))
            assert k >= 0
            nacc.append(k)
            meta_all.append(meta[:k, :3].copy())
            sha_all.append(hashlib.sha256(sbuf[: used.value].tobytes()).hexdigest())
        out[name + "_cands"] = norm
        out[name + "_tmpl_begin"] = tb
        out[name + "_nacc"] = np.array(nacc, dtype=np.int32)
        out[name + "_meta"] = np.concatenate(meta_all) if meta_all else np.zeros((0, 3), np.int32)
        out[name + "_sha"] = np.array(sha_all)
        out[name + "_par"] = np.array([n, L, G, seed, ont, tech, mas], dtype=np.int64)
        out[name + "_ratio"] = np.array([err, ratio])
        print(name, "templates", n, "candidates", len(norm), "accepted", int(sum(nacc)), "max per template", max(nacc), file=sys.stderr)
    np.savez_compressed(os.path.join(H.GOLDEN, "cns_accept.npz"), **out)


if __name__ == "__main__":
    main()
