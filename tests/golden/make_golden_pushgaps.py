#!/usr/bin/env python3
"""tests/golden/pushgaps.npz: the UNMODIFIED normalize_gaps(push = true) of the reference (reads_correction_aux.cpp:3-81, through
oracle/_ref/libref_cns_accept.so: refa_normalize_gaps) on 700 pairs of gapped strings without mismatch columns (what the O(ND) aligner
produces): random indel columns at 5 - 40 %, long gap runs, runs that reach the end of the string, adjacent query / template gaps,
homopolymer stretches (where pushing moves a gap a long way), lengths 1 .. 2 500.  Inputs and the reference's outputs, as bytes.
Build container only:  python tests/golden/make_golden_pushgaps.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import helpers as H  # noqa: E402


def make_pairs(seed=9, count=700):
    rng = np.random.default_rng(seed)
    pairs = []
    for it in range(count):
        n = int(rng.integers(1, 40)) if it % 10 == 0 else int(rng.integers(40, 2500))
        mode = it % 6
        alpha = 1 if mode == 4 else (2 if mode == 5 else 4)          # homopolymer / two-letter stretches push gaps far
        pg = float(rng.choice([0.05, 0.15, 0.4]))
        q, t = [], []
        while len(q) < n:
            u = rng.random()
            run = 1
            if mode in (1, 3) and rng.random() < 0.1:
                run = int(rng.integers(2, 60))                       # long gap runs
            base = "ACGT"[int(rng.integers(0, alpha))]
            if u < pg / 2:
                for _ in range(run):
                    q.append("-"); t.append("ACGT"[int(rng.integers(0, alpha))])
            elif u < pg:
                for _ in range(run):
                    q.append("ACGT"[int(rng.integers(0, alpha))]); t.append("-")
            else:
                q.append(base); t.append(base)                      # match columns only: no mismatches in an O(ND) alignment
        q, t = q[:n], t[:n]
        if mode == 3 and n > 8:                                      # a run that reaches the end of the string
            k = int(rng.integers(1, min(n - 1, 30)))
            for i in range(n - k, n):
                t[i] = "-"; q[i] = "ACGT"[int(rng.integers(0, 4))]
        pairs.append(("".join(q), "".join(t)))
    return pairs


if __name__ == "__main__":
    L = C.CDLL(os.path.join(H.ROOT, "oracle", "_ref", "libref_cns_accept.so"))
    L.refa_normalize_gaps.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
    pairs = make_pairs()
    qin, tin, qo, to, lens = [], [], [], [], []
    for q, t in pairs:
        n = len(q)
        a = C.create_string_buffer(2 * n + 8)
        b = C.create_string_buffer(2 * n + 8)
        m = L.refa_normalize_gaps(q.encode(), t.encode(), n, a, b, 2 * n + 8)
        assert m == n, (m, n)                                        # no mismatch columns: the length stays
        qin.append(q); tin.append(t); qo.append(a.value.decode()); to.append(b.value.decode()); lens.append(n)
    changed = sum(1 for i in range(len(pairs)) if qo[i] != qin[i] or to[i] != tin[i])
    print("pairs %d, changed by the push %d, characters %d" % (len(pairs), changed, sum(lens)))
    np.savez_compressed(os.path.join(H.GOLDEN, "pushgaps.npz"), lens=np.array(lens, dtype=np.int32),
                        qin=np.frombuffer("".join(qin).encode(), dtype=np.uint8), tin=np.frombuffer("".join(tin).encode(), dtype=np.uint8),
                        qout=np.frombuffer("".join(qo).encode(), dtype=np.uint8), tout=np.frombuffer("".join(to).encode(), dtype=np.uint8))
