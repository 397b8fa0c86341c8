"""SURVEY.md §8f row N4 — mecat2cns' candidate partition files written by mecat_amd/host/partition.cpp (standalone tool
mecat2cns_partition; the mecat2pw driver feeds the same writer, tests/test_gpu_cli.py) against the compiled reference
(oracle/_ref/libref_part.so, this container and the GPU box) and the restatement oracle/partition_oracle.py."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H

sys.path.insert(0, os.path.join(H.ROOT, "oracle"))
import partition_oracle as PO  # noqa: E402

TOOL = os.path.join(H.ROOT, "mecat_amd", "bin", "mecat2cns_partition")
REFLIB = os.path.join(H.ROOT, "oracle", "_ref", "libref_part.so")


def _tool():
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-C", H.ROOT, "host"], check=True, capture_output=True)
    return TOOL


def _random_can(rng, n, nreads, min_size):
    lines = []
    for _ in range(n):
        qid, sid = int(rng.integers(0, nreads)), int(rng.integers(0, nreads))
        qs = int(rng.integers(min_size - 300, min_size + 5000))
        ss = int(rng.integers(min_size - 300, min_size + 5000))
        lines.append("%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d" % (qid, sid, int(rng.integers(0, 2)), int(rng.integers(0, 2)),
                                                             int(rng.integers(0, qs)), int(rng.integers(0, ss)), int(rng.integers(1, 90)), qs, ss))
    return "\n".join(lines) + ("\n" if lines else "")


def _read_parts(can):
    """-> ({k: int32 array (n, 13)}, index lines with the directory stripped)"""
    d, base = os.path.dirname(can), os.path.basename(can)
    files = {}
    for f in os.listdir(d):
        if f.startswith(base + ".part") and not f.endswith("partition_files"):
            a = np.fromfile(os.path.join(d, f), dtype=np.int32)
            assert a.size % 13 == 0, f
            files[int(f[len(base) + 5:])] = a.reshape(-1, 13)
    idx = []
    for ln in open(can + ".partition_files").read().splitlines():
        name, lo, hi = ln.split("\t")
        assert os.path.dirname(name) == d
        idx.append((os.path.basename(name), int(lo), int(hi)))
    return files, idx


def _check_against_oracle(can, text, batch, min_size):
    files, idx = _read_parts(can)
    ofiles, oidx = PO.partition(PO.parse_can(text), batch, min_size)
    assert sorted(files) == sorted(ofiles)
    for k in ofiles:
        want = np.array(ofiles[k], dtype=np.int32).reshape(-1, 13)
        assert files[k].shape == want.shape, k
        assert np.array_equal(files[k][:, PO.DEFINED], want[:, PO.DEFINED]), k
    assert idx == [(os.path.basename(can) + ".part%d" % k, lo, hi) for k, lo, hi in oidx]
    return files, idx


@pytest.mark.parametrize("n,nreads,batch,threads", [(0, 10, 5, 1), (1, 10, 5, 2), (5000, 2300, 500, 4), (20000, 977, 100, 7)])
def test_tool_equals_oracle(tmp_path, n, nreads, batch, threads):
    rng = np.random.default_rng(n + batch)
    text = _random_can(rng, n, nreads, 2000)
    can = str(tmp_path / "x.can")
    open(can, "w").write(text)
    r = subprocess.run([_tool(), can, str(batch), "2000", str(threads)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    files, _ = _check_against_oracle(can, text, batch, 2000)
    if n:
        assert not np.any(np.concatenate(list(files.values()))[:, [4, 5, 10, 11]])      # the undefined ints are written as 0


def test_same_batch_line_writes_query_side_first(tmp_path):
    text = "3\t4\t0\t1\t10\t20\t7\t9000\t9001\n"
    can = str(tmp_path / "y.can")
    open(can, "w").write(text)
    assert subprocess.run([_tool(), can, "100", "5000"], capture_output=True).returncode == 0
    files, idx = _check_against_oracle(can, text, 100, 5000)
    # seen from the query's side the template is read 3 (forward already); as it stands the template 4 is reversed -> both strands flip
    assert files[0].tolist() == [[1, 4, 20, 9001, 0, 0, 0, 3, 10, 9000, 0, 0, 7], [1, 3, 10, 9000, 0, 0, 0, 4, 20, 9001, 0, 0, 7]]
    assert idx == [("y.can.part0", 3, 4)]


def test_malformed_line_is_an_error(tmp_path):
    can = str(tmp_path / "z.can")
    open(can, "w").write("1\t2\t0\t0\t5\t6\t7\t9000\n")
    r = subprocess.run([_tool(), can, "100", "5000"], capture_output=True, text=True)
    assert r.returncode != 0 and "malformed" in r.stderr


@pytest.mark.skipif(not os.path.exists(REFLIB), reason="oracle/_ref/libref_part.so not built (needs /root/reference)")
@pytest.mark.parametrize("num_files", [3, 64])
def test_tool_and_oracle_equal_reference(tmp_path, num_files):
    """the compiled, unmodified partition_candidates: same files, same record order, same index (num_files = 3 makes it take
    several passes over the text; the bytes do not depend on that)"""
    rng = np.random.default_rng(11)
    text = _random_can(rng, 30000, 1500, 2000)
    for sub in ("ref", "ours"):
        os.mkdir(tmp_path / sub)
        open(tmp_path / sub / "c.can", "w").write(text)
    code = ("import ctypes as C; L = C.CDLL(%r); L.refp_partition_candidates.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_int]; "
            "L.refp_partition_candidates(%r, 200, 2000, %d)" % (REFLIB, str(tmp_path / "ref" / "c.can").encode(), num_files))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([_tool(), str(tmp_path / "ours" / "c.can"), "200", "2000", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rf, ri = _read_parts(str(tmp_path / "ref" / "c.can"))
    of, oi = _check_against_oracle(str(tmp_path / "ours" / "c.can"), text, 200, 2000)
    assert ri == oi and sorted(rf) == sorted(of)
    for k in rf:
        assert rf[k].shape == of[k].shape
        assert np.array_equal(rf[k][:, PO.DEFINED], of[k][:, PO.DEFINED]), k


# ---- the m4 flavour (partition_m4records)
def _random_m4(rng, n, nreads, min_size):
    lines = []
    for _ in range(n):
        qid, sid = int(rng.integers(0, nreads)), int(rng.integers(0, nreads))
        qs = int(rng.integers(min_size - 300, min_size + 5000))
        ss = int(rng.integers(min_size - 300, min_size + 5000))
        qo = int(rng.integers(0, qs // 2)); qe = int(rng.integers(qo + 1, qs + 1))
        so = int(rng.integers(0, ss // 2)); se = int(rng.integers(so + 1, ss + 1))
        if rng.random() < 0.3:
            qo, qe = 0, qs - int(rng.integers(0, qs // 8))
        lines.append("%d\t%d\t%g\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d" % (
            qid, sid, 70 + 30 * rng.random(), int(rng.integers(1, 90)), 0, qo, qe, qs, int(rng.integers(0, 2)), so, se, ss,
            int(rng.integers(qo, qe)), int(rng.integers(so, se))))
    return "\n".join(lines) + ("\n" if lines else "")


def _check_m4_against_oracle(path, text, ratio, batch, min_size):
    files, idx = _read_parts(path)
    ofiles, oidx = PO.partition_m4(PO.parse_m4(text), ratio, batch, min_size)
    assert sorted(files) == sorted(ofiles)
    for k in ofiles:
        assert np.array_equal(files[k], np.array(ofiles[k], dtype=np.int32).reshape(-1, 13)), k
    assert idx == [(os.path.basename(path) + ".part%d" % k, lo, hi) for k, lo, hi in oidx]
    return files


@pytest.mark.parametrize("n,threads", [(0, 1), (4000, 3)])
def test_m4_tool_equals_oracle(tmp_path, n, threads):
    rng = np.random.default_rng(5 + n)
    text = _random_m4(rng, n, 900, 2000)
    path = str(tmp_path / "o.m4")
    open(path, "w").write(text)
    r = subprocess.run([_tool(), "-m", "0.88", path, "128", "2000", str(threads)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    files = _check_m4_against_oracle(path, text, 0.88, 128, 2000)
    if n:
        assert 200 < sum(len(v) for v in files.values()) < 2 * n          # the coverage filter drops some, keeps some


def test_m4_without_gapped_start_points_is_the_reference_error(tmp_path):
    path = str(tmp_path / "g0.m4")
    open(path, "w").write("1\t2\t88.5\t30\t0\t0\t9000\t9000\t0\t0\t9000\t9000\n")
    r = subprocess.run([_tool(), "-m", "0.88", path, "100", "2000"], capture_output=True, text=True)
    assert r.returncode != 0 and "-g 1" in r.stderr


@pytest.mark.skipif(not os.path.exists(REFLIB), reason="oracle/_ref/libref_part.so not built (needs /root/reference)")
def test_m4_tool_and_oracle_equal_reference(tmp_path):
    rng = np.random.default_rng(12)
    text = _random_m4(rng, 20000, 1500, 2000)
    for sub in ("ref", "ours"):
        os.mkdir(tmp_path / sub)
        open(tmp_path / sub / "o.m4", "w").write(text)
    ratio = 0.9 - 0.02                                            # what mecat2cns passes for PacBio (reads_correction_m4.cpp:79)
    code = ("import ctypes as C; L = C.CDLL(%r); L.refp_partition_m4records.argtypes = [C.c_char_p, C.c_double, C.c_long, C.c_int, C.c_int]; "
            "L.refp_partition_m4records(%r, %r, 200, 2000, 4)" % (REFLIB, str(tmp_path / "ref" / "o.m4").encode(), ratio))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([_tool(), "-m", repr(ratio), str(tmp_path / "ours" / "o.m4"), "200", "2000", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rf, ri = _read_parts(str(tmp_path / "ref" / "o.m4"))
    of = _check_m4_against_oracle(str(tmp_path / "ours" / "o.m4"), text, ratio, 200, 2000)
    _, oi = _read_parts(str(tmp_path / "ours" / "o.m4"))
    assert ri == oi and sorted(rf) == sorted(of)
    for k in rf:
        assert np.array_equal(rf[k], of[k]), k                    # all 13 ints are defined in this flavour
