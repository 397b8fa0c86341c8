#!/usr/bin/env python3
"""Reference pins at BASELINE.json's FULL sizes (configs 2, 3 and 5): runs the UNMODIFIED reference (oracle/_ref/mecat2pw, built
by `make ref` from /root/reference/src) on the synthetic read sets of SURVEY.md §8d and records hashes of its output in
tests/golden/big.json.  Everything written is data (counts + SHA-256 of `LC_ALL=C sort`-ed output lines and of the volume
files); the read sets are re-generated from their seeds by mecat_amd/bin/synth_reads wherever the tests run.

    python tests/golden/make_golden_big.py config2            # ~4 min -j 0 + ~25 min -j 1 on 7 threads
    python tests/golden/make_golden_big.py config3            # 3 volumes, 6 grid cells, -j 0: about an hour
    python tests/golden/make_golden_big.py config3_ecoli      # same reads x length on a 4.6 Mb genome (1300x): row 2 only
    python tests/golden/make_golden_big.py config5            # 19 volumes, -x 1: -j 0 rows 17 and 18, -j 1 row 18 (resume protocol)
    python tests/golden/make_golden_big.py config3_j1         # config 3, `-j 1 -g 1`, grid row 1 = cells (1,1) and (1,2): dw extension
                                                              # across two real volumes (adds config3.m4_rows; rows 0 and 2 planted)
    python tests/golden/make_golden_big.py config5_row0       # config 5, `-j 0 -x 1`, grid row 0 = 19 cells (adds config5.rows["0"])
    python tests/golden/make_golden_big.py config5_row17_j1   # config 5, `-j 1 -x 1 -g 1`, grid row 17 = cells (17,17) and (17,18) (adds config5.m4_row17)

Rows that are not pinned are skipped with the reference's own resume protocol: an existing wrk/r_<i> means "volume i has been
finished" (mecat2pw/pw.cpp:65-81), so empty r_<i> files are planted for them before the run.
"""
import hashlib
import json
import os
import struct
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "big.json")
REF = os.path.join(ROOT, "oracle", "_ref", "mecat2pw")
SYNTH = os.path.join(ROOT, "mecat_amd", "bin", "synth_reads")
WORK = os.environ.get("MECAT_BIG_DIR", "/tmp/mecat_big")
THREADS = os.environ.get("MECAT_BIG_THREADS", "7")

# name: (nreads, L, err, genome, seed, ont)  — SURVEY.md §8d
SETS = {
    "config2": (100_000, 15000, 0.15, 50_000_000, 2, 0),
    "config3": (500_000, 12000, 0.15, 200_000_000, 3, 0),
    "config3_ecoli": (500_000, 12000, 0.15, 4_600_000, 3, 0),
    "config5": (2_000_000, 20000, 0.12, 1_300_000_000, 5, 1),
}


def sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def sorted_sha(path, extra_filter=None):
    """(lines, sha256) of `LC_ALL=C sort path`; extra_filter = awk program selecting lines first"""
    env = dict(os.environ, LC_ALL="C")
    if extra_filter:
        cmd = "awk -F'\\t' '%s' %s | sort -S 4G | tee >(wc -l >&2) | sha256sum" % (extra_filter, path)
    else:
        cmd = "sort -S 4G %s | tee >(wc -l >&2) | sha256sum" % path
    p = subprocess.run(["bash", "-c", cmd], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True)
    return int(p.stderr.strip().splitlines()[-1]), p.stdout.split()[0]


def vol_header(path):
    with open(path, "rb") as f:
        num_reads, num_bases, start_id = struct.unpack("<iii", f.read(12))
    return num_reads, num_bases, start_id


def gen(name):
    n, L, e, G, seed, ont = SETS[name]
    os.makedirs(WORK, exist_ok=True)
    fa = os.path.join(WORK, name + ".fa")
    if not os.path.exists(fa):
        subprocess.run([SYNTH, fa + ".tmp", str(n), str(L), str(e), str(G), str(seed), str(ont)], check=True)
        os.rename(fa + ".tmp", fa)
    return fa


def run_ref(fa, out, wrk, args, skip_rows=()):
    if os.path.exists(out + ".done"):                # an earlier, finished run of the same command: reuse its files
        return float(open(out + ".done").read() or 0)
    os.makedirs(wrk, exist_ok=True)
    for i in skip_rows:
        open(os.path.join(wrk, "r_%d" % i), "w").close()
    t0 = time.time()
    with open(out + ".log", "w") as lg:
        subprocess.run(["nice", "-n", "10", REF, "-d", fa, "-o", out, "-w", wrk, "-t", THREADS] + args, check=True, stdout=lg, stderr=lg)
    open(out + ".done", "w").write("%.1f" % (time.time() - t0))
    return time.time() - t0


def volumes(wrk):
    names = [ln.strip() for ln in open(os.path.join(wrk, "fileindex.txt")) if ln.strip()]
    vols = []
    for p in names:
        nr, nb, sid = vol_header(p)
        vols.append({"num_reads": nr, "num_bases": nb, "start_read_id": sid, "sha256": sha_file(p)})
    return vols


def cell_filter(vols, j):
    lo = vols[j]["start_read_id"]
    hi = lo + vols[j]["num_reads"]
    return "$1 >= %d && $1 < %d" % (lo, hi)          # .can field 1 = query read id


def aligned_bases(path):
    return int(subprocess.run(["awk", "-F\t", "{s += $7 - $6} END {printf \"%.0f\", s}", path], stdout=subprocess.PIPE, text=True,
                              check=True).stdout)


def extra(big, name):
    """pins added to an existing entry of big.json (the entry's volumes must equal the ones this run splits)"""
    base = "config3" if name == "config3_j1" else "config5"
    n, L, e, G, seed, ont = SETS[base]
    fa = gen(base)
    m = big[base]
    assert os.path.getsize(fa) == m["fasta_bytes"]
    d = os.path.join(WORK, base)
    os.makedirs(d, exist_ok=True)
    nv = len(m["volumes"])
    if name == "config3_j1":
        out = os.path.join(d, "c3.m4")
        wrk = os.path.join(d, "w1")
        secs = run_ref(fa, out, wrk, ["-j", "1", "-g", "1"], skip_rows=(0, 2))
        vols = volumes(wrk)
        assert [v["sha256"] for v in vols] == [v["sha256"] for v in m["volumes"]]
        r = os.path.join(wrk, "r_1")
        row = {"seconds": secs, "threads": int(THREADS)}
        row["lines"], row["sorted_sha256"] = sorted_sha(r)
        row["aligned_bases"] = aligned_bases(r)
        row["cells"] = {}
        for j in (1, 2):                              # .m4 field 2 = query read id (field 1 is the subject)
            lo = vols[j]["start_read_id"]
            c = {}
            c["lines"], c["sorted_sha256"] = sorted_sha(r, "$2 >= %d && $2 < %d" % (lo, lo + vols[j]["num_reads"]))
            row["cells"]["1,%d" % j] = c
        m.setdefault("m4_rows", {})["1"] = row
    elif name == "config5_row17_j1":
        # -j 1 -x 1 -g 1 (X-drop extension) of grid row 17 = the diagonal cell (17, 17) and the off-diagonal cell (17, 18) at real volume size
        out = os.path.join(d, "c5r17.m4")
        wrk = os.path.join(d, "w3")
        secs = run_ref(fa, out, wrk, ["-j", "1", "-x", "1", "-g", "1"], skip_rows=[i for i in range(nv) if i != 17])
        vols = volumes(wrk)
        assert [(v["num_reads"], v["num_bases"], v["start_read_id"]) for v in vols] == \
               [(v["num_reads"], v["num_bases"], v["start_read_id"]) for v in m["volumes"]]
        r = os.path.join(wrk, "r_17")
        row = {"seconds": secs, "threads": int(THREADS)}
        row["lines"], row["sorted_sha256"] = sorted_sha(r)
        row["aligned_bases"] = aligned_bases(r)
        row["cells"] = {}
        for j in (17, 18):
            lo = vols[j]["start_read_id"]
            c = {}
            c["lines"], c["sorted_sha256"] = sorted_sha(r, "$2 >= %d && $2 < %d" % (lo, lo + vols[j]["num_reads"]))
            row["cells"]["17,%d" % j] = c
        m["m4_row17"] = row
    else:
        out = os.path.join(d, "c5r0.can")
        wrk = os.path.join(d, "w2")
        secs = run_ref(fa, out, wrk, ["-j", "0", "-x", "1"], skip_rows=range(1, nv))
        vols = volumes(wrk)
        assert [(v["num_reads"], v["num_bases"], v["start_read_id"]) for v in vols] == \
               [(v["num_reads"], v["num_bases"], v["start_read_id"]) for v in m["volumes"]]
        r = os.path.join(wrk, "r_0")
        row = {"seconds": secs, "threads": int(THREADS)}
        row["lines"], row["sorted_sha256"] = sorted_sha(r)
        row["cells"] = {}
        for j in range(nv):
            c = {}
            c["lines"], c["sorted_sha256"] = sorted_sha(r, cell_filter(vols, j))
            row["cells"]["0,%d" % j] = c
        assert [v["sha256"] for v in vols] == [v["sha256"] for v in m["volumes"]]
        m["rows"]["0"] = row
    json.dump(big, open(OUT, "w"), indent=1, sort_keys=True)
    print(name, json.dumps(row)[:600], file=sys.stderr)


def main():
    which = sys.argv[1:] or ["config2"]
    big = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in which:
        if name in ("config3_j1", "config5_row0", "config5_row17_j1"):
            extra(big, name)
            big = json.load(open(OUT))
            continue
        n, L, e, G, seed, ont = SETS[name]
        fa = gen(name)
        m = {"gen": dict(nreads=n, L=L, err=e, genome=G, seed=seed, ont=ont), "fasta_bytes": os.path.getsize(fa),
             "reference_threads": int(THREADS)}
        d = os.path.join(WORK, name)
        os.makedirs(d, exist_ok=True)
        if name == "config2":
            out = os.path.join(d, "c2.can")
            m["j0_seconds"] = run_ref(fa, out, os.path.join(d, "w0"), ["-j", "0"])
            m["can_lines"], m["can_sorted_sha256"] = sorted_sha(out)
            m["volumes"] = volumes(os.path.join(d, "w0"))
            out = os.path.join(d, "c2.m4")
            m["j1_seconds"] = run_ref(fa, out, os.path.join(d, "w1"), ["-j", "1", "-g", "1"])
            m["m4_g1_lines"], m["m4_g1_sorted_sha256"] = sorted_sha(out)
            m["m4_aligned_bases"] = int(subprocess.run(["awk", "-F\t", "{s += $7 - $6} END {printf \"%.0f\", s}", out], stdout=subprocess.PIPE,
                                                       text=True, check=True).stdout)
        elif name in ("config3", "config3_ecoli"):
            out = os.path.join(d, "c3.can")
            wrk = os.path.join(d, "w0")
            skip = (0, 1) if name == "config3_ecoli" else ()
            m["j0_seconds"] = run_ref(fa, out, wrk, ["-j", "0"], skip_rows=skip)
            vols = volumes(wrk)
            m["volumes"] = vols
            m["rows"] = {}
            for i in range(len(vols)):
                if i in skip:
                    continue
                r = os.path.join(wrk, "r_%d" % i)
                row = {}
                row["lines"], row["sorted_sha256"] = sorted_sha(r)
                row["cells"] = {}
                for j in range(i, len(vols)):
                    c = {}
                    c["lines"], c["sorted_sha256"] = sorted_sha(r, cell_filter(vols, j))
                    row["cells"]["%d,%d" % (i, j)] = c
                m["rows"][str(i)] = row
            if not skip:
                m["can_lines"], m["can_sorted_sha256"] = sorted_sha(out)
        elif name == "config5":
            out = os.path.join(d, "c5.can")
            wrk = os.path.join(d, "w0")
            pinned = (17, 18)
            # 19 volumes expected (40 Gbase / 2.14 Gbase); the rows before the pinned ones are planted as finished
            m["j0_seconds"] = run_ref(fa, out, wrk, ["-j", "0", "-x", "1"], skip_rows=range(0, pinned[0]))
            if all(os.path.exists(ln.strip()) for ln in open(os.path.join(wrk, "fileindex.txt")) if ln.strip()):
                vols = volumes(wrk)
            else:                                    # an earlier run's volume files were removed: keep what it recorded
                vols = big[name]["volumes"]
            assert len(vols) == 19, len(vols)
            m["volumes"] = vols
            m["rows"] = {}
            for i in pinned:
                r = os.path.join(wrk, "r_%d" % i)
                row = {}
                row["lines"], row["sorted_sha256"] = sorted_sha(r)
                m["rows"][str(i)] = row
            # -j 1 (X-drop extension) of the last row only: cell (18, 18)
            out1 = os.path.join(d, "c5.m4")
            wrk1 = os.path.join(d, "w1")
            m["j1_seconds"] = run_ref(fa, out1, wrk1, ["-j", "1", "-x", "1", "-g", "1"], skip_rows=range(0, 18))
            m["m4_row18"] = {}
            m["m4_row18"]["lines"], m["m4_row18"]["sorted_sha256"] = sorted_sha(os.path.join(wrk1, "r_18"))
        big[name] = m
        json.dump(big, open(OUT, "w"), indent=1, sort_keys=True)
        print(name, json.dumps({k: v for k, v in m.items() if k != "volumes"})[:600], file=sys.stderr)


if __name__ == "__main__":
    main()
