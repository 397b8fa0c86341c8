#!/usr/bin/env python3
"""BASELINE config 4 ("mecat2cns consensus on config-2 overlaps") up to the consensus table, pinned to the UNMODIFIED reference:
consensus_one_read_can_pacbio (oracle/_ref/libref_cns_accept.so = mecat2cns compiled from /root/reference/src) run on the first
T templates of config 2's own candidate file — the reference mecat2pw's `-j 0` output of make_golden_big.py config2
(/tmp/mecat_big/config2/c2.can), brought into the order of a candidate table (query read ascending, list order inside a read:
the order tests/test_gpu_cns_accept.py rebuilds the same records in from the device's table) and turned into per-template
records by mecat_amd/workload.py:cns_templates (= normalise_candidate + grouping).  Build container only:
    python tests/golden/make_golden_cns_config2.py [T]
Writes tests/golden/cns_config2.npz: template ids, per template the number of accepted alignments, (soff, send, aln_size) of every
CnsAlns entry in order and a SHA-256 over its gap-normalised strings; plus the reference's wall time (one thread)."""
import ctypes as C
import hashlib
import os
import sys
import time

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from mecat_amd import workload as W  # noqa: E402

BIG = os.environ.get("MECAT_BIG_DIR", "/tmp/mecat_big")
MAS, RATIO = 2000, 0.9          # mecat2cns -a / -r defaults for PacBio reads (options.cpp:13-27)


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    n, L, err, G, seed, ont = W.CONFIGS["config2"]
    can = os.path.join(BIG, "config2", "c2.can")
    fa = os.path.join(BIG, "config2.fa")
    df = pd.read_csv(can, sep="\t", header=None, dtype=np.int64).to_numpy()      # qid sid qdir sdir qext sext score qsize ssize
    df = df[np.argsort(df[:, 0], kind="stable")]                                   # table order: query read ascending, file (= list) order inside
    ec = np.zeros((len(df), 13), dtype=np.int32)
    ec[:, 0], ec[:, 1], ec[:, 2], ec[:, 3] = df[:, 2], df[:, 0], df[:, 4], df[:, 7]
    ec[:, 6], ec[:, 7], ec[:, 8], ec[:, 9], ec[:, 12] = df[:, 3], df[:, 1], df[:, 5], df[:, 8], df[:, 6]
    rec, tb, ids = W.cns_templates(ec, n)
    T = min(T, len(ids))
    A = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_cns_accept.so"))
    A.refa_load_reads.argtypes = [C.c_char_p]
    A.refa_consensus_can.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
    assert A.refa_load_reads(fa.encode()) == n
    nacc, metas, shas = [], [], []
    sbuf = np.zeros(400_000_000, dtype=np.int8)
    t0 = time.time()
    naln = 0
    for t in range(T):
        b, e = int(tb[t]), int(tb[t + 1])
        cand = np.ascontiguousarray(rec[b:e]).copy()
        meta = np.zeros((128, 4), dtype=np.int32)
        used = C.c_long()
        k = A.refa_consensus_can(0, cand.ctypes.data, e - b, int(ids[t]), MAS, RATIO, meta.ctypes.data, sbuf.ctypes.data, len(sbuf), C.byref(used))
        assert k >= 0
        naln += min(e - b, 200)
        nacc.append(k)
        metas.append(meta[:k, :3].copy())
        shas.append(hashlib.sha256(sbuf[: used.value].tobytes()).hexdigest())
    secs = time.time() - t0
    np.savez_compressed(os.path.join(HERE, "cns_config2.npz"), ids=ids[:T].astype(np.int64), tmpl_begin=tb[: T + 1].astype(np.int64),
                        rec_sha=np.array(hashlib.sha256(np.ascontiguousarray(rec[: tb[T]]).tobytes()).hexdigest()),
                        nacc=np.array(nacc, dtype=np.int32), meta=np.concatenate(metas) if metas else np.zeros((0, 3), np.int32), sha=np.array(shas),
                        par=np.array([MAS, T], dtype=np.int64), ratio=np.array([RATIO]), ref_seconds=np.array([secs]), ref_alignments_upper=np.array([naln]))
    print("templates", T, "candidates", int(tb[T]), "accepted", int(sum(nacc)), "reference %.1f s on one thread" % secs, file=sys.stderr)


if __name__ == "__main__":
    main()
