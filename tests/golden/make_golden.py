#!/usr/bin/env python3
"""Generates the committed golden vectors in tests/golden/ from the UNMODIFIED reference (oracle/_ref, built by
`make ref` from /root/reference/src).  Run in the build container only:  python tests/golden/make_golden.py

Everything written is data: expected outputs of the reference (and, for the function-level KATs, the random inputs they
were computed on).  The read sets are re-generated from their seeds by mecat_amd/tools/synth_reads.c.
"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import helpers as H  # noqa: E402

OUT = H.GOLDEN
SETS = {
    # name: (nreads, L, err, genome, seed, ont)
    "tiny": (200, 3000, 0.15, 30000, 11, 0),
    "tiny_ont": (150, 4000, 0.12, 40000, 12, 1),
    "config1": (1000, 10000, 0.15, 500000, 1, 0),
}


def sha(b):
    return hashlib.sha256(b).hexdigest()


def run_ref(fa, d, args, name):
    out = os.path.join(d, name)
    wrk = os.path.join(d, "w_" + name)
    subprocess.run([H.ref_bin(), "-d", fa, "-o", out, "-w", wrk, "-t", "4"] + args, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(open(out).read().splitlines()), wrk


def mutate(rng, s, e):
    out = []
    for b in s:
        u = rng.random()
        if u < 0.25 * e:
            continue
        out.append(int(rng.integers(0, 4)) if u < 0.4 * e else int(b))
        if rng.random() < 0.6 * e:
            out.append(int(rng.integers(0, 4)))
    return np.array(out, dtype=np.int8)


def main():
    R = H.ref()
    meta = {"sets": {}, "sizeof": {k: R.refh_sizeof(i) for i, k in enumerate(
        ["Back_List", "candidate_save", "M4Record", "ExtensionCandidate", "DPathData2", "offset_t", "volume_t"])}}
    d = tempfile.mkdtemp(prefix="golden_")
    for name, (n, L, e, G, seed, ont) in SETS.items():
        codes, lens = H.synth_reads(n, L, e, G, seed, ont)
        fa = os.path.join(d, name + ".fa")
        H.write_fasta(fa, codes, lens)
        m = {"gen": dict(nreads=n, L=L, err=e, genome=G, seed=seed, ont=ont), "total_bases": int(len(codes)),
             "lens_sha256": sha(lens.tobytes())}
        tech = ont
        can, wrk = run_ref(fa, d, ["-j", "0", "-x", str(tech)], name + ".can")
        m["vol0_sha256"] = sha(open(os.path.join(wrk, "vol0"), "rb").read())
        m["can_sorted_sha256"] = sha(("\n".join(can) + "\n").encode())
        m["can_lines"] = len(can)
        if name != "config1":
            open(os.path.join(OUT, name + ".can.sorted"), "w").write("\n".join(can) + "\n")
        # index digest + literal buckets
        rv = R.refh_load_volume(os.path.join(wrk, "vol0").encode())
        ridx = R.refh_build_index(rv, 1)
        counts = np.empty(H.NK, dtype=np.int32)
        nk = R.refh_index_dump(ridx, counts.ctypes.data, None)
        offs = np.empty(nk, dtype=np.int32)
        R.refh_index_dump(ridx, counts.ctypes.data, offs.ctypes.data)
        m["index"] = {"num_kmers": int(nk), "counts_sha256": sha(counts.tobytes()), "offsets_sha256": sha(offs.tobytes()),
                      "nonempty": int((counts > 0).sum()), "max_count": int(counts.max())}
        if name == "tiny":
            ne = np.nonzero(counts)[0]
            starts = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
            pick = ne[:: max(1, len(ne) // 50)][:50]
            m["index"]["buckets"] = {int(k): [int(x) for x in offs[starts[k]: starts[k + 1]]] for k in pick}
        # per-read candidate_save arrays
        if name != "config1":
            cs = {}
            for maxc in (100, 5):
                mkm = 2 if tech else 4
                R.refh_set_params(maxc, 500 if tech else 2000, mkm, tech)
                buf = np.zeros((maxc, 12), dtype=np.int32)
                rows, cnts = [], []
                for rid in range(n):
                    k = R.refh_seed_read(rv, rv, ridx, rid, 0, buf.ctypes.data)
                    rows.append(buf[:k].copy())
                    cnts.append(k)
                cs["cands_maxc%d" % maxc] = np.concatenate(rows) if rows else np.zeros((0, 12), np.int32)
                cs["counts_maxc%d" % maxc] = np.array(cnts, dtype=np.int32)
            np.savez_compressed(os.path.join(OUT, name + "_cands.npz"), **cs)
        if ont:
            m4, _ = run_ref(fa, d, ["-j", "1", "-x", "1", "-g", "1"], "%s.g1.m4" % name)
            m["m4_g1_sorted_sha256"] = sha(("\n".join(m4) + "\n").encode())
            m["m4_g1_lines"] = len(m4)
            open(os.path.join(OUT, "%s.g1.m4.sorted" % name), "w").write("\n".join(m4) + "\n")
        if not ont:
            for g in (0, 1):
                if name == "config1" and g == 0:
                    continue
                m4, _ = run_ref(fa, d, ["-j", "1", "-g", str(g)], "%s.g%d.m4" % (name, g))
                m["m4_g%d_sorted_sha256" % g] = sha(("\n".join(m4) + "\n").encode())
                m["m4_g%d_lines" % g] = len(m4)
                if name != "config1":
                    open(os.path.join(OUT, "%s.g%d.m4.sorted" % (name, g)), "w").write("\n".join(m4) + "\n")
                elif g == 1:
                    # aligned query bases (col 7 - col 6) for the metric definition
                    m["m4_aligned_bases"] = int(sum(int(x.split("\t")[6]) - int(x.split("\t")[5]) for x in m4))
        R.refh_free_index(ridx)
        R.refh_free_volume(rv)
        meta["sets"][name] = m
        print(name, {k: v for k, v in m.items() if k != "index"}, file=sys.stderr)

    # ---- function-level KATs (inputs + reference outputs)
    rng = np.random.default_rng(2024)
    kat = {}
    # find_location: the SURVEY §8c vector + random
    fl_in, fl_out = [], []
    cases = [(np.array([100, 110, 120, 131, 140, 150, 160, 170, 900], np.int32), np.arange(1, 10, dtype=np.int32), 10000)]
    for it in range(200):
        k = int(rng.integers(1, 81))
        if it % 2:
            seedn = np.sort(rng.integers(1, 400, size=k)).astype(np.int32)
            loc = (seedn * 10 + rng.integers(-30, 31, size=k) + 200).astype(np.int32)
        else:
            seedn = rng.integers(1, 200, size=k).astype(np.int32)
            loc = rng.integers(0, 4000, size=k).astype(np.int32)
        cases.append((loc, seedn, int(rng.integers(500, 20000))))
    for loc, seedn, rl in cases:
        k = len(loc)
        sc = np.zeros(k, np.int32)
        lo = np.zeros(4, np.int32)
        rep = C.c_int(-1)
        l2, s2 = loc.copy(), seedn.copy()
        r = R.refh_find_location(l2.ctypes.data, s2.ctypes.data, sc.ctypes.data, lo.ctypes.data, k, C.byref(rep), 10.0, rl)
        row_in = np.full(2 * 80 + 2, -1, np.int32)
        row_in[0], row_in[1] = k, rl
        row_in[2: 2 + k] = loc
        row_in[82: 82 + k] = seedn
        row_out = np.full(80 + 6, -1, np.int32)
        row_out[0] = r
        row_out[1] = rep.value if r else -1
        row_out[2:6] = lo if r else 0
        row_out[6: 6 + k] = sc
        fl_in.append(row_in)
        fl_out.append(row_out)
    kat["find_location_in"] = np.stack(fl_in)
    kat["find_location_out"] = np.stack(fl_out)
    # insert_loc
    il_in, il_out = [], []
    for it in range(150):
        mode = it % 3
        base = int(rng.integers(0, 1500))
        lz = np.zeros(40, np.int16)
        sn = np.zeros(40, np.int16)
        for i in range(40):
            if mode == 0:
                lz[i], sn[i] = (base + 10 * i) % 2000, 1 + i
            elif mode == 1:
                lz[i], sn[i] = int(np.clip(base + 10 * i + rng.integers(-8, 9), 0, 1999)), 1 + i + int(rng.integers(0, 2))
            else:
                lz[i], sn[i] = int(rng.integers(0, 2000)), int(rng.integers(1, 1500))
        loc = int(rng.integers(0, 2000)) if mode == 2 else int(np.clip(base + 400 + rng.integers(-3, 4), 0, 1999))
        seedn = int(rng.integers(1, 1500)) if mode == 2 else 41
        sc = np.array([41], np.int16)
        il_in.append(np.concatenate([[41, loc, seedn], lz, sn]).astype(np.int32))
        R.refh_insert_loc(sc.ctypes.data, lz.ctypes.data, sn.ctypes.data, loc, seedn, 10.0)
        il_out.append(np.concatenate([[int(sc[0])], lz, sn]).astype(np.int32))
    kat["insert_loc_in"] = np.stack(il_in)
    kat["insert_loc_out"] = np.stack(il_out)
    # Align
    al_q, al_t, al_par, al_res, al_qa, al_ta = [], [], [], [], [], []
    survey_q = np.array([0, 1, 2, 3] * 5, np.int8)
    survey_t = np.array(["ACGT".index(c) for c in "ACGTACTACGTACGGTACGT"], np.int8)
    acases = [(survey_q, survey_t, 6, 1)]
    for it in range(60):
        nq = int(rng.integers(20, 600))
        q = rng.integers(0, 4, size=nq).astype(np.int8)
        t = mutate(rng, q, [0.0, 0.05, 0.15, 0.3, 0.6][it % 5])
        if len(t) < 5:
            continue
        band = int(0.3 * max(len(q), len(t))) if it % 7 else int(rng.integers(2, 40))
        acases.append((q, t, band, it % 2))
    for q, t, band, right in acases:
        res = np.zeros(6, np.int32)
        qa = np.zeros(4096, np.int8)
        ta = np.zeros(4096, np.int8)
        r = R.refh_align(q.ctypes.data, len(q), t.ctypes.data, len(t), band, 1, right, res.ctypes.data, qa.ctypes.data, ta.ctypes.data)
        al_q.append(q); al_t.append(t)
        al_par.append([len(q), len(t), band, right])
        al_res.append([r] + list(res))
        al_qa.append(qa[: res[0]].copy()); al_ta.append(ta[: res[0]].copy())
    kat["align_q"] = np.concatenate(al_q); kat["align_t"] = np.concatenate(al_t)
    kat["align_par"] = np.array(al_par, np.int32); kat["align_res"] = np.array(al_res, np.int32)
    kat["align_qaln"] = np.concatenate(al_qa); kat["align_taln"] = np.concatenate(al_ta)
    # DiffAligner::go
    dq, dt, dpar, dres = [], [], [], []
    for it in range(40):
        n = int(rng.integers(800, 5000))
        g = rng.integers(0, 4, size=n + 2000).astype(np.int8)
        a0, b0 = int(rng.integers(0, 1000)), int(rng.integers(0, 1000))
        q = mutate(rng, g[a0: a0 + n], 0.15)
        t = mutate(rng, g[b0: b0 + n], 0.15)
        mid = max(a0, b0) + n // 3
        qs = int((mid - a0) * 1.05) if it % 5 else int(rng.integers(0, len(q)))
        ts = int((mid - b0) * 1.05) if it % 5 else int(rng.integers(0, len(t)))
        qs = min(max(qs, 0), len(q) - 1)
        ts = min(max(ts, 0), len(t) - 1)
        if it % 11 == 0:
            qs = 0
        res = np.zeros(7, np.int32)
        ident = C.c_double()
        R.refh_dw_go(q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), 500, res.ctypes.data, C.byref(ident))
        dq.append(q); dt.append(t)
        dpar.append([len(q), len(t), qs, ts, 500])
        dres.append(list(res))
    kat["dw_q"] = np.concatenate(dq); kat["dw_t"] = np.concatenate(dt)
    kat["dw_par"] = np.array(dpar, np.int32); kat["dw_res"] = np.array(dres, np.int32)
    # XdropAligner::go (nanopore mode)
    xq, xt, xpar, xres = [], [], [], []
    for it in range(40):
        n = int(rng.integers(600, 5000))
        g = rng.integers(0, 4, size=n + 2000).astype(np.int8)
        a0, b0 = int(rng.integers(0, 1000)), int(rng.integers(0, 1000))
        q = mutate(rng, g[a0: a0 + n], 0.12)
        t = mutate(rng, g[b0: b0 + n], 0.12)
        mid = max(a0, b0) + n // 3
        qs = int((mid - a0) * 1.04) if it % 5 else int(rng.integers(0, len(q)))
        ts = int((mid - b0) * 1.04) if it % 5 else int(rng.integers(0, len(t)))
        qs = min(max(qs, 0), len(q) - 1)
        ts = min(max(ts, 0), len(t) - 1)
        if it % 11 == 0:
            qs = 0
        res = np.zeros(7, np.int32)
        ident = C.c_double()
        R.refh_xdrop_go(q.ctypes.data, qs, len(q), t.ctypes.data, ts, len(t), 500, res.ctypes.data, C.byref(ident))
        xq.append(q); xt.append(t)
        xpar.append([len(q), len(t), qs, ts, 500])
        xres.append(list(res))
    kat["xd_q"] = np.concatenate(xq); kat["xd_t"] = np.concatenate(xt)
    kat["xd_par"] = np.array(xpar, np.int32); kat["xd_res"] = np.array(xres, np.int32)
    np.savez_compressed(os.path.join(OUT, "kats.npz"), **kat)
    json.dump(meta, open(os.path.join(OUT, "golden.json"), "w"), indent=1, sort_keys=True)
    print("wrote", OUT, file=sys.stderr)


if __name__ == "__main__":
    main()
