#!/usr/bin/env python3
"""Golden look-up table of mecat2canu's overlapper: the UNMODIFIED creat_ref_index of mecat2asmpw.c
(oracle/_ref/libref_asmpw.so: the reference's C file compiled where it lies, main renamed) on one block of seeded synthetic reads
with two planted tandem repeats — one whose 13-mers occur between 129 and 256 times (kept here, dropped by mecat2pw's cap of 128)
and one whose 13-mers occur more than 256 times (dropped here too, mecat2asmpw.c:307-314).  Build container only.
Writes tests/golden/asmpw_index.npz: the reads, and every non-empty bucket (k-mer id in mecat2pw's A, C, G, T = 0..3 digit order,
occurrences, 0-based positions)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import helpers as H  # noqa: E402

GEN = dict(nreads=80, L=2500, err=0.02, genome=40000, seed=77, ont=0)
K = 13


def reads():
    codes, lens = H.synth_reads(GEN["nreads"], GEN["L"], GEN["err"], GEN["genome"], GEN["seed"], GEN["ont"])
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    out = [codes[starts[i]: starts[i + 1]].copy() for i in range(len(lens))]
    rng = np.random.default_rng(5)
    unit7 = rng.integers(0, 4, size=7).astype(np.uint8)         # 7 distinct 13-mers x ~194 occurrences: kept (<= 256)
    out[5][300:300 + 7 * 200] = np.tile(unit7, 200)
    unit5 = np.array([0, 1, 2, 3, 1], dtype=np.uint8)            # 5 distinct 13-mers x ~2 x 288 occurrences: dropped (> 256)
    for r in (10, 11):
        out[r][200:200 + 5 * 300] = np.tile(unit5, 300)
    lens = np.array([len(r) for r in out], dtype=np.int32)
    return np.concatenate(out).astype(np.uint8), lens


def reference_index(codes, lens):
    lib = C.CDLL(os.path.join(H.ROOT, "oracle", "_ref", "libref_asmpw.so"))
    lib.refasm_index.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.POINTER(C.c_int)))]
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    text = b"".join(bytes(b"ACGT"[c] for c in codes[starts[i]: starts[i + 1]]) + b"\0" for i in range(len(lens)))
    buf = C.create_string_buffer(text, len(text))
    counts, index = C.POINTER(C.c_int)(), C.POINTER(C.POINTER(C.c_int))()
    total = lib.refasm_index(buf, len(text), K, C.byref(counts), C.byref(index))
    cnt = np.ctypeslib.as_array(counts, shape=(4 ** K,)).copy()
    ids = np.nonzero(cnt)[0]
    pos = [np.ctypeslib.as_array(index[int(i)], shape=(int(cnt[i]),)).copy() for i in ids]
    assert sum(len(p) for p in pos) == total
    lib.refasm_index_free()
    return ids, cnt[ids], pos


def to_pw_id(ids):
    """k-mer id in the tool's digit order (A, T, C, G = 0..3, atcttrans :298-305) -> the id of the same 13-mer in mecat2pw's (A, C, G, T)"""
    perm = np.array([0, 3, 1, 2], dtype=np.int64)
    out = np.zeros(len(ids), dtype=np.int64)
    x = ids.astype(np.int64)
    for d in range(K):
        out |= perm[(x >> (2 * d)) & 3] << (2 * d)
    return out


def main():
    codes, lens = reads()
    ids, cnt, pos = reference_index(codes, lens)
    pw = to_pw_id(ids)
    order = np.argsort(pw, kind="stable")
    pw, cnt = pw[order], cnt[order]
    positions = np.concatenate([pos[i] - 1 for i in order]).astype(np.int32)      # the tool stores start + 1 (:500)
    assert cnt.max() <= 256 and (cnt > 128).sum() >= 5, (cnt.max(), (cnt > 128).sum())
    np.savez_compressed(os.path.join(H.GOLDEN, "asmpw_index.npz"), codes=codes, lens=lens, ids=pw.astype(np.uint32), counts=cnt.astype(np.int32),
                        positions=positions)
    print("buckets", len(pw), "positions", len(positions), "max", cnt.max(), "in (128, 256]:", int((cnt > 128).sum()), file=sys.stderr)


if __name__ == "__main__":
    main()
