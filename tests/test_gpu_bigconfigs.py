"""Reference pins at BASELINE.json's full sizes, through the drop-in binary (mecat_amd/bin/mecat2pw, i.e. FASTA split -> volumes ->
C ABI -> text output), against hashes of the UNMODIFIED reference's output on the same synthetic read sets
(tests/golden/big.json, written by tests/golden/make_golden_big.py from oracle/_ref/mecat2pw in the build container):

  config 2   100 000 x 15 kb @ 15 %, one volume:    `-j 0` and `-j 1 -g 1`, sorted-output SHA-256, line counts, aligned bases
  config 3   500 000 x 12 kb @ 15 %, three volumes: `-j 0`, every grid row r_<i> and every grid cell (i, j) hashed separately
             (real 2.14 Gbase volume limit, int32 coordinates up to the limit, no test knob); `-j 1 -g 1` grid row 1 = cells (1, 1)
             and (1, 2): dw extension across two real volumes
  config 5   2 000 000 x 20 kb ONT-style, 19 volumes, `-x 1`: `-j 0` grid rows 0, 17 and 18 and `-j 1 -g 1` (X-drop extension) rows 17
             (cells (17, 17) and (17, 18): an off-diagonal X-drop cell of two real volumes, 365 463 overlaps) and 18
             against the reference (the rows before them are planted as finished through the reference's own resume protocol,
             as the golden run did);
             MECAT_TEST_CONFIG5_FULL=1 runs all 190 cells and checks the size-independent properties on the whole output.

Line order is not part of the contract: both sides are compared as `LC_ALL=C sort`-ed multisets.
"""
import hashlib
import json
import os
import shutil
import struct
import subprocess
import tempfile
import time

import pytest

import helpers as H

pytestmark = pytest.mark.gpu

BIN = os.path.join(H.ROOT, "mecat_amd", "bin", "mecat2pw")
SYNTH = os.path.join(H.ROOT, "mecat_amd", "bin", "synth_reads")
BIG = os.path.join(H.GOLDEN, "big.json")


def _golden(name):
    if not os.path.exists(BIG):
        pytest.skip("tests/golden/big.json missing")
    g = json.load(open(BIG))
    if name not in g:
        pytest.skip("no reference pin for %s in tests/golden/big.json" % name)
    return g[name]


@pytest.fixture()
def workdir():
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="mecat_big_", dir=base)
    yield d
    shutil.rmtree(d, ignore_errors=True)


def _gen(d, g):
    fa = os.path.join(d, "reads.fa")
    gen = g["gen"]
    subprocess.run([SYNTH, fa, str(gen["nreads"]), str(gen["L"]), str(gen["err"]), str(gen["genome"]), str(gen["seed"]), str(gen["ont"])],
                   check=True, stderr=subprocess.DEVNULL)
    assert os.path.getsize(fa) == g["fasta_bytes"]
    return fa


def _sorted_sha(path, awk_filter=None):
    env = dict(os.environ, LC_ALL="C")
    src = "awk -F'\\t' '%s' %s" % (awk_filter, path) if awk_filter else "cat %s" % path
    cmd = "%s | sort -S 8G --parallel=16 | tee >(wc -l >&2) | sha256sum" % src
    p = subprocess.run(["bash", "-c", cmd], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True)
    return int(p.stderr.strip().splitlines()[-1]), p.stdout.split()[0]


def _run(args, fa, out, wrk, threads=32, env=None):
    t0 = time.time()
    r = subprocess.run([BIN, "-d", fa, "-o", out, "-w", wrk, "-t", str(threads)] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    return time.time() - t0


def _volumes(wrk):
    vols = []
    for ln in open(os.path.join(wrk, "fileindex.txt")):
        p = ln.strip()
        if not p:
            continue
        with open(p, "rb") as f:
            nr, nb, sid = struct.unpack("<iii", f.read(12))
        vols.append({"path": p, "num_reads": nr, "num_bases": nb, "start_read_id": sid})
    return vols


def _sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for b in iter(lambda: f.read(1 << 24), b""):
            h.update(b)
    return h.hexdigest()


def test_config2_cli_equals_reference(workdir):
    g = _golden("config2")
    fa = _gen(workdir, g)
    out = os.path.join(workdir, "o.can")
    _run(["-j", "0"], fa, out, os.path.join(workdir, "w0"))
    vols = _volumes(os.path.join(workdir, "w0"))
    assert len(vols) == 1 and _sha_file(vols[0]["path"]) == g["volumes"][0]["sha256"]
    assert _sorted_sha(out) == (g["can_lines"], g["can_sorted_sha256"])
    out = os.path.join(workdir, "o.m4")
    _run(["-j", "1", "-g", "1"], fa, out, os.path.join(workdir, "w1"))
    assert _sorted_sha(out) == (g["m4_g1_lines"], g["m4_g1_sorted_sha256"])
    ab = int(subprocess.run(["awk", "-F\t", "{s += $7 - $6} END {printf \"%.0f\", s}", out], stdout=subprocess.PIPE, text=True, check=True).stdout)
    assert ab == g["m4_aligned_bases"]


def test_config2_partition_files_two_ranks_equal_one_process(workdir):
    """SURVEY.md §8f row N4 at config-2 size in a multi-process run: two ranks (both on GPU 0, host-file transport)
    write the partition records of their own lines as streams, rank 0 merges them by (row, query read) — no text is parsed — and every
    partition file has the bytes of the one-process run's (2.2 M candidate lines, 4.4 M records of 52 bytes, 20 batches of 5 000 reads)."""
    import uuid
    g = _golden("config2")
    fa = _gen(workdir, g)
    os.mkdir(os.path.join(workdir, "a"))
    os.mkdir(os.path.join(workdir, "b"))
    one = os.path.join(workdir, "a", "o.can")
    env = {"MECAT_HIP_PARTITION": "5000"}
    _run(["-j", "0"], fa, one, os.path.join(workdir, "w_one"), env=env)
    assert _sorted_sha(one) == (g["can_lines"], g["can_sorted_sha256"])
    two = os.path.join(workdir, "b", "o.can")
    run = uuid.uuid4().hex[:10]
    procs = []
    for rank in (1, 0):
        e = dict(os.environ, MECAT_HIP_WORLD="2", MECAT_HIP_RANK=str(rank), MECAT_HIP_DEVICE="0", MECAT_HIP_SHARD="cells", MECAT_HIP_COMM="file",
                 MECAT_HIP_RUN_ID=run, MECAT_HIP_COMM_TIMEOUT_S="300", MECAT_HIP_WAIT_S="600", **env)
        procs.append(subprocess.Popen([BIN, "-j", "0", "-d", fa, "-o", two, "-w", os.path.join(workdir, "w_two"), "-t", "16"], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=e))
    errs = ""
    for p in procs:
        out, err = p.communicate(timeout=900)
        errs += err
        assert p.returncode == 0, err[-3000:]
    assert "partition_files(text)" not in errs
    assert _sorted_sha(two) == (g["can_lines"], g["can_sorted_sha256"])
    names = sorted(f for f in os.listdir(os.path.join(workdir, "a")) if f.startswith("o.can.part"))
    assert names == sorted(f for f in os.listdir(os.path.join(workdir, "b")) if f.startswith("o.can.part")) and len(names) == 21
    total = 0
    for f in names:
        a, b = open(os.path.join(workdir, "a", f), "rb").read(), open(os.path.join(workdir, "b", f), "rb").read()
        if f.endswith(".partition_files"):
            assert a.replace(one.encode(), b"<out>") == b.replace(two.encode(), b"<out>")
        else:
            assert a == b, f
            total += len(a)
    assert 0.98 * 104 * g["can_lines"] < total <= 104 * g["can_lines"]      # (two records per line whose reads both reach the 5 000-base minimum)


@pytest.mark.parametrize("name", ["config3", "config3_ecoli"])
def test_config3_three_volume_grid_equals_reference(workdir, name):
    g = _golden(name)
    fa = _gen(workdir, g)
    out = os.path.join(workdir, "o.can")
    wrk = os.path.join(workdir, "w0")
    os.makedirs(wrk)
    pinned = sorted(int(i) for i in g["rows"])
    for i in range(len(g["volumes"])):
        if i not in pinned:
            open(os.path.join(wrk, "r_%d" % i), "w").close()          # "volume i has been finished" (pw.cpp:65-81), as in the golden run
    _run(["-j", "0"], fa, out, wrk)
    vols = _volumes(wrk)
    assert [(v["num_reads"], v["num_bases"], v["start_read_id"]) for v in vols] == [(v["num_reads"], v["num_bases"], v["start_read_id"])
                                                                                     for v in g["volumes"]]
    assert len(vols) == 3
    for v, gv in zip(vols, g["volumes"]):
        assert _sha_file(v["path"]) == gv["sha256"]
    for i in pinned:
        row = g["rows"][str(i)]
        r = os.path.join(wrk, "r_%d" % i)
        assert _sorted_sha(r) == (row["lines"], row["sorted_sha256"]), "row %d" % i
        for cell, c in row["cells"].items():
            j = int(cell.split(",")[1])
            lo, hi = vols[j]["start_read_id"], vols[j]["start_read_id"] + vols[j]["num_reads"]
            assert _sorted_sha(r, "$1 >= %d && $1 < %d" % (lo, hi)) == (c["lines"], c["sorted_sha256"]), "cell " + cell
    if "can_sorted_sha256" in g:
        assert _sorted_sha(out) == (g["can_lines"], g["can_sorted_sha256"])


def test_config3_extension_across_two_real_volumes(workdir):
    """`-j 1 -g 1` on grid row 1 of config 3 = cells (1, 1) and (1, 2): dw extension of candidates whose query reads live in another
    2.14 Gbase volume than their subject reads (local ids, int32 volume coordinates up to the limit, no test knob), against the
    unmodified reference's r_1 (rows 0 and 2 planted as finished on both sides: pw.cpp:65-81)."""
    g = _golden("config3")
    if "m4_rows" not in g:
        pytest.skip("no -j 1 pin for config 3 in tests/golden/big.json")
    fa = _gen(workdir, g)
    wrk = os.path.join(workdir, "w1")
    os.makedirs(wrk)
    pinned = sorted(int(i) for i in g["m4_rows"])
    for i in range(len(g["volumes"])):
        if i not in pinned:
            open(os.path.join(wrk, "r_%d" % i), "w").close()
    _run(["-j", "1", "-g", "1"], fa, os.path.join(workdir, "o.m4"), wrk)
    os.unlink(fa)
    vols = _volumes(wrk)
    for i in pinned:
        row = g["m4_rows"][str(i)]
        r = os.path.join(wrk, "r_%d" % i)
        assert _sorted_sha(r) == (row["lines"], row["sorted_sha256"]), "row %d" % i
        ab = int(subprocess.run(["awk", "-F\t", "{s += $7 - $6} END {printf \"%.0f\", s}", r], stdout=subprocess.PIPE, text=True, check=True).stdout)
        assert ab == row["aligned_bases"]
        for cell, c in row["cells"].items():
            j = int(cell.split(",")[1])          # .m4 field 2 = the query read (field 1 is the subject)
            lo, hi = vols[j]["start_read_id"], vols[j]["start_read_id"] + vols[j]["num_reads"]
            assert _sorted_sha(r, "$2 >= %d && $2 < %d" % (lo, hi)) == (c["lines"], c["sorted_sha256"]), "cell " + cell


def test_config5_nanopore_19_volume_grid(workdir):
    g = _golden("config5")
    full = os.environ.get("MECAT_TEST_CONFIG5_FULL") == "1"
    fa = _gen(workdir, g)
    out = os.path.join(workdir, "o.can")
    wrk = os.path.join(workdir, "w0")
    os.makedirs(wrk)
    pinned = sorted(int(i) for i in g["rows"])
    if not full:
        # rows that are not pinned are skipped through the resume protocol (an existing r_<i> = "volume i is finished")
        for i in range(len(g["volumes"])):
            if i not in pinned:
                open(os.path.join(wrk, "r_%d" % i), "w").close()
    secs = _run(["-j", "0", "-x", "1"], fa, out, wrk, threads=64)
    if "m4_row18" in g:
        # X-drop extension at this scale, `-j 1 -x 1 -g 1`: grid row 18 = cell (18, 18), and (when pinned) grid row 17 = the diagonal
        # cell (17, 17) and the off-diagonal cell (17, 18) of two real volumes; the other rows planted as finished
        rows1 = [18] + ([17] if "m4_row17" in g else [])
        wrk1 = os.path.join(workdir, "w1")
        os.makedirs(wrk1)
        for i in range(19):
            if i not in rows1:
                open(os.path.join(wrk1, "r_%d" % i), "w").close()
        _run(["-j", "1", "-x", "1", "-g", "1"], fa, os.path.join(workdir, "o.m4"), wrk1, threads=64)
        assert _sorted_sha(os.path.join(wrk1, "r_18")) == (g["m4_row18"]["lines"], g["m4_row18"]["sorted_sha256"])
        if 17 in rows1:
            r17 = os.path.join(wrk1, "r_17")
            m = g["m4_row17"]
            assert _sorted_sha(r17) == (m["lines"], m["sorted_sha256"])
            ab = int(subprocess.run(["awk", "-F\t", "{s += $7 - $6} END {printf \"%.0f\", s}", r17], stdout=subprocess.PIPE, text=True, check=True).stdout)
            assert ab == m["aligned_bases"]
            for cell, c in m["cells"].items():              # .m4 field 2 = the query read (field 1 is the subject)
                j = int(cell.split(",")[1])
                lo = g["volumes"][j]["start_read_id"]
                assert _sorted_sha(r17, "$2 >= %d && $2 < %d" % (lo, lo + g["volumes"][j]["num_reads"])) == (c["lines"], c["sorted_sha256"]), cell
        shutil.rmtree(wrk1, ignore_errors=True)
    os.unlink(fa)
    vols = _volumes(wrk)
    assert len(vols) == 19
    for v, gv in zip(vols, g["volumes"]):
        assert (v["num_reads"], v["num_bases"], v["start_read_id"]) == (gv["num_reads"], gv["num_bases"], gv["start_read_id"])
    for i in pinned:
        assert _sha_file(vols[i]["path"]) == g["volumes"][i]["sha256"]
        row = g["rows"][str(i)]
        assert _sorted_sha(os.path.join(wrk, "r_%d" % i)) == (row["lines"], row["sorted_sha256"]), "row %d" % i
        for cell, c in row.get("cells", {}).items():      # .can field 1 = the query read (field 2 is the subject, of volume i)
            j = int(cell.split(",")[1])
            lo, hi = vols[j]["start_read_id"], vols[j]["start_read_id"] + vols[j]["num_reads"]
            assert _sorted_sha(os.path.join(wrk, "r_%d" % i), "$1 >= %d && $1 < %d" % (lo, hi)) == (c["lines"], c["sorted_sha256"]), "cell " + cell
    if full:
        # size-independent properties over all 190 cells: subject < query, coordinates inside the reads, sdir == 0 (mecat2cns
        # asserts it, mecat_correction.cpp:423), every row file holds subjects of its own volume only, at most MAXC lines per
        # (query read, reference volume)
        starts = [v["start_read_id"] for v in vols] + [vols[-1]["start_read_id"] + vols[-1]["num_reads"]]
        bad = subprocess.run(["awk", "-F\t", "$2 >= $1 || $4 != 0 || $5 < 0 || $5 >= $8 || $6 < 0 || $6 >= $9 || ($3 != 0 && $3 != 1) {b++} END {printf \"%d\", b}", out],
                             stdout=subprocess.PIPE, text=True, check=True).stdout
        assert int(bad) == 0
        total = 0
        for i in range(19):
            r = os.path.join(wrk, "r_%d" % i)
            chk = subprocess.run(["awk", "-F\t", "-v", "lo=%d" % starts[i], "-v", "hi=%d" % starts[i + 1],
                                  "$2 < lo || $2 >= hi {b++} {c[$1]++} END {m = 0; for (k in c) if (c[k] > m) m = c[k]; printf \"%d %d %d\", b, m, NR}", r],
                                 stdout=subprocess.PIPE, text=True, check=True).stdout.split()
            assert int(chk[0]) == 0 and int(chk[1]) <= 100, (i, chk)
            total += int(chk[2])
        lines = int(subprocess.run(["wc", "-l", out], stdout=subprocess.PIPE, text=True, check=True).stdout.split()[0])
        assert lines == total and lines > 20 * g["gen"]["nreads"]
        print("config5 full grid: %d candidates, 190 cells, %.1f s end to end" % (lines, secs))
