#!/usr/bin/env python3
"""Known answers for the mecat2cns re-aligner, generated in the build container from the
UNMODIFIED reference through oracle/_ref/libref_cns.so (oracle/ref_harness_cns.cpp -> src/mecat2cns/dw.cpp):

    python tests/golden/make_golden_cns.py        ->  tests/golden/cns_kats.npz

Inputs (code arrays), parameters and expected outputs of ns_banded_sw::dw and GetAlignment; the alignment strings are
kept as SHA-256 digests."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import helpers as H  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    R = H.ref_cns()
    rng = np.random.default_rng(2025)
    qs_, ts_, par, dw_res, ga_res, dig = [], [], [], [], [], []
    for it in range(48):
        n = int(rng.integers(300, 9000))
        err = [0.0, 0.06, 0.12, 0.15, 0.3][it % 5]
        q, t, qs, ts = H.cns_pair(rng, n, err, it)
        er100 = 15 if it % 2 else 20
        min_aln = 500 if it % 3 else 50
        res = np.zeros(9, np.int32)
        s1 = np.zeros(100001, np.int8)
        s2 = np.zeros(100001, np.int8)
        ok = R.refc_dw(q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), er100 / 100.0, min_aln, res.ctypes.data, s1.ctypes.data, s2.ctypes.data)
        d1 = hashlib.sha256(s1[: res[4]].tobytes() + b"|" + s2[: res[4]].tobytes()).hexdigest()
        res2 = np.zeros(5, np.int32)
        ok2 = R.refc_get_alignment(q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), er100 / 100.0, min_aln, res2.ctypes.data, s1.ctypes.data,
                                   s2.ctypes.data)
        d2 = hashlib.sha256(s1[: res2[4]].tobytes() + b"|" + s2[: res2[4]].tobytes()).hexdigest() if ok2 else ""
        qs_.append(q); ts_.append(t)
        par.append([len(q), len(t), qs, ts, min_aln, er100])
        dw_res.append([ok] + list(res))
        ga_res.append([ok2] + list(res2))
        dig.append(d1 + ":" + d2)
    np.savez_compressed(os.path.join(OUT, "cns_kats.npz"), q=np.concatenate(qs_), t=np.concatenate(ts_), par=np.array(par, np.int32),
                        dw_res=np.array(dw_res, np.int32), ga_res=np.array(ga_res, np.int32), digests=np.array(dig))
    print("wrote cns_kats.npz:", len(par), "cases,", int(sum(r[0] for r in ga_res)), "aligned", file=sys.stderr)


if __name__ == "__main__":
    main()
